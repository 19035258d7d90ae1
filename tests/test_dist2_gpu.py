"""World size 2 on ONE MI355X: two processes share cuda:0 and talk over gloo (RCCL refuses two ranks on one device),
so the REAL engine runs the real multi-rank step: per-rank batches, the packed embedding all-gather with the
slice-only backward, the asynchronous per-block gradient all-reduce started from inside backward, the 1/W convention
(HipDataParallel.backward) and AdamW.  Checks (SURVEY.md 8e "must match"):
  * both ranks hold identical parameters after every step (same collectives, same order);
  * every rank reports the same global loss, and it equals the single-process loss on the concatenated batch;
  * the parameters track a single-process run on the concatenated batch (AdamW is scale-invariant up to eps, so the
    1/W gradient convention leaves the update direction unchanged)."""
import argparse
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
STEPS, BR, T, L = 2, 2, 2, 10


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    from OATrans import model as module_arch
    torch.manual_seed(11)
    m = module_arch.FrozenInTime(
        video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=T, pretrained=True,
                          time_init="rand", arch_kwargs=dict(depth=2)),
        object_params=dict(model="", input_objects=False),
        text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text", config=dict(n_layers=1)),
        projection="minimal", load_checkpoint="").cuda()
    m.set_device(torch.device("cuda"))
    m.text_model.eval()          # dropout masks are drawn per local row: 2 x 2 rows and 1 x 4 rows would see different masks
    for sub in (m.video_model, m.text_model):
        sub.flatten_parameters()
    return m


def _batch(rank_lo, rank_hi):
    g = torch.Generator().manual_seed(21)
    video = torch.randn(2 * BR, T, 3, 224, 224, generator=g)
    ids = torch.randint(1000, 30000, (2 * BR, L), generator=g)
    sl = slice(rank_lo * BR, rank_hi * BR)
    return {"video": video[sl].cuda(), "text": {"input_ids": ids[sl].cuda(),
                                                "attention_mask": torch.ones(ids[sl].shape, dtype=torch.int64).cuda()}}


def _train(world, rank):
    from OATrans import model as module_arch
    from OATrans.optim import AdamW
    from OATrans.parallel import HipDataParallel
    from OATrans.trainer.step import hot_step
    m = _model()
    dp = HipDataParallel(m)
    opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    args = argparse.Namespace(world_size=world, rank=rank, local_rank=0)
    data = _batch(rank, rank + 1) if world > 1 else _batch(0, 2)
    losses = [hot_step(dp, module_arch.NormSoftmaxLoss(), opt, data, args).item() for _ in range(STEPS)]
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().float().flatten() for p in m.parameters()]).cpu().numpy()
    return losses, flat


def _worker(rank, world, port, q):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        losses, flat = _train(world, rank)
        q.put((rank, losses, flat))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_each_other_and_the_single_process_run():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, l0, p0), (_, l1, p1) = res
    assert l0 == l1, (l0, l1)                                   # every rank computes the same global loss
    assert np.array_equal(p0, p1)                               # and holds the same parameters, bit for bit
    ls, ps = _train(1, 0)                                       # single process, concatenated batch
    assert abs(l0[0] - ls[0]) < 2e-3 * abs(ls[0]), (l0, ls)    # identical weights at step 0: same loss up to bf16 tiling noise
    assert abs(l0[1] - ls[1]) < 2e-2 * abs(ls[1]), (l0, ls)
    d = np.abs(p0 - ps)
    # STEPS AdamW steps of at most lr each; the directions agree except where a tiny gradient flips sign under the noise
    assert d.max() <= 2 * STEPS * 1e-4 + 1e-6 and d.mean() < 2e-5, (d.max(), d.mean())
