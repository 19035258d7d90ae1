"""The N>1 code path on ONE MI355X: a 1-rank RCCL process group with GradSync(force=True) issues the same
collectives (packed embedding all-gather, per-block asynchronous gradient all-reduce started from inside
backward, final sync) that a multi-GPU job issues.  With one rank every collective is the identity, so the
losses must equal a run without a process group - up to the run-to-run noise of the fp32 atomics that
accumulate the CLS-row attention gradients (first step identical, later steps within 1e-3).
(World-size-2 semantics: tests/test_parallel_cpu.py.)"""
import argparse
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _model():
    from OATrans import model as module_arch
    torch.manual_seed(3)
    m = module_arch.FrozenInTime(
        video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=2, pretrained=True,
                          time_init="rand", arch_kwargs=dict(depth=3)),
        object_params=dict(model="", input_objects=False),
        text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text", config=dict(n_layers=2)),
        projection="minimal", load_checkpoint="")
    m = m.cuda()
    m.set_device(torch.device("cuda"))
    for sub in (m.video_model, m.text_model):
        sub.flatten_parameters()
    return m


def _run(force):
    from OATrans import model as module_arch
    from OATrans.optim import AdamW
    from OATrans.parallel import GradSync, HipDataParallel
    from OATrans.trainer.step import hot_step
    m = _model()
    dp = HipDataParallel(m)
    announced = []
    if force:
        dp.sync = GradSync(m, overlap=True, force=True)
        orig = dp.sync.on_ready
        dp.sync.on_ready = lambda mod, lo, hi: (announced.append((type(mod).__name__, lo, hi)), orig(mod, lo, hi))[1]
        for sub in (m.video_model, m.text_model):
            sub.grad_ready_hook = dp.sync.on_ready
    opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    g = torch.Generator().manual_seed(5)
    data = {"video": torch.randn(4, 2, 3, 224, 224, generator=g).cuda(),
            "text": {"input_ids": torch.randint(1000, 30000, (4, 12), generator=g).cuda(),
                     "attention_mask": torch.ones(4, 12, dtype=torch.int64).cuda()}}
    args = argparse.Namespace(world_size=1, rank=0, local_rank=0)
    losses = [hot_step(dp, module_arch.NormSoftmaxLoss(), opt, data, args).item() for _ in range(3)]
    torch.cuda.synchronize()
    return losses, announced, m


def test_one_rank_rccl_group_runs_the_overlapped_gradient_sync():
    base, _, _ = _run(force=False)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        got, announced, m = _run(force=True)
    finally:
        dist.destroy_process_group()
    assert got[0] == base[0] and all(abs(a - b) < 1e-3 * abs(b) for a, b in zip(got, base)), (got, base)
    # per step: text tower once + one range per ViT block + the embedding tables
    per_step = [a for a in announced[:len(announced) // 3]]
    assert len(per_step) == 1 + 3 + 1, per_step
    vid = [a for a in per_step if a[0] == "SpaceTimeTransformer"]
    covered = sum(hi - lo for _, lo, hi in vid)
    assert covered == m.video_model.flat_grad().numel()          # the announced ranges tile the whole buffer
    assert sorted((lo, hi) for _, lo, hi in vid)[0][0] == 0
