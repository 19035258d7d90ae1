"""N>1 path on CPU with gloo, world_size 2 (the GPU box runs the same code over RCCL):
AllGather_multi semantics, the packed gather, flat-range gradient mean all-reduce, and the
(1/W) gradient convention of trainer_dist.py:29-45 + DDP (SURVEY.md 8e 'must match')."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oatrans_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import argparse
    from OATrans.parallel import AllGather_multi, GradSync, allgather_pair
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    args = argparse.Namespace(world_size=world, rank=rank, local_rank=rank)
    torch.manual_seed(100)                                 # same "model" on every rank
    W1 = torch.nn.Parameter(torch.randn(16, 8))
    W2 = torch.nn.Parameter(torch.randn(16, 8))
    torch.manual_seed(200 + rank)                          # different data per rank
    xa, xb = torch.randn(3, 8), torch.randn(3, 8)
    t, v = xa @ W1.t(), xb @ W2.t()
    # 1. AllGather_multi: rank-order concat forward
    g = AllGather_multi.apply(t, world, args)
    assert g.shape == (world * 3, 16) and torch.equal(g[3 * rank:3 * rank + 3], t)
    # 2. packed gather == two gathers; loss on the global batch
    v_all, t_all = allgather_pair(v, t, args)
    assert torch.equal(t_all, g)
    loss = orc.norm_softmax_loss(orc.sim_matrix(t_all, v_all))
    loss.backward()
    # 3. flat-range mean all-reduce over a shared gradient buffer
    flat = torch.cat([W1.grad.flatten(), W2.grad.flatten()])
    W1.grad, W2.grad = flat[:128].view(16, 8), flat[128:].view(16, 8)
    model = torch.nn.ParameterList([W1, W2])
    sync = GradSync(model)
    assert len(sync.ranges()) == 1 and sync.ranges()[0].numel() == 256
    sync.all_reduce(average=True)
    # 4. HipDataParallel.backward: 1/W on the loss + SUM all-reduce must give the same mean, bit for bit (W = 2)
    from OATrans.parallel import HipDataParallel
    lin = torch.nn.Linear(8, 16, bias=False)
    with torch.no_grad():
        lin.weight.copy_(W1)
    dp = HipDataParallel(lin)
    v_all2, t_all2 = allgather_pair(xb @ W2.detach().t(), dp(xa), args)
    dp.backward(orc.norm_softmax_loss(orc.sim_matrix(t_all2, v_all2)))
    dp.sync_gradients()
    assert torch.equal(lin.weight.grad, W1.grad), (lin.weight.grad - W1.grad).abs().max()
    # numpy, not torch tensors: torch shares tensor storage by file descriptor, which fails if this process exits
    # before the parent has unpickled the message
    q.put((rank, loss.item(), W1.grad.numpy().copy(), W2.grad.numpy().copy(), xa.numpy().copy(), xb.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_and_gradient_convention():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    res = [(r[0], r[1]) + tuple(torch.from_numpy(a) for a in r[2:]) for r in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process truth on the concatenated batch
    torch.manual_seed(100)
    W1 = torch.randn(16, 8, requires_grad=True)
    W2 = torch.randn(16, 8, requires_grad=True)
    xa = torch.cat([r[4] for r in res])
    xb = torch.cat([r[5] for r in res])
    loss = orc.norm_softmax_loss(orc.sim_matrix(xa @ W1.t(), xb @ W2.t()))
    loss.backward()
    for r in res:
        assert abs(r[1] - loss.item()) < 1e-5                       # every rank sees the same global loss
        # slice-only backward + gradient MEAN == (1/W) * dL_global/dtheta
        assert torch.allclose(r[2], W1.grad / world, atol=1e-5)
        assert torch.allclose(r[3], W2.grad / world, atol=1e-5)
    assert torch.equal(res[0][2], res[1][2])


def test_single_rank_gather_is_identity():
    import argparse
    from OATrans.parallel import AllGather_multi
    args = argparse.Namespace(world_size=1, rank=0, local_rank=0)
    x = torch.randn(4, 5, requires_grad=True)
    y = AllGather_multi.apply(x, 1, args)
    y.sum().backward()
    assert torch.equal(y, x) and torch.equal(x.grad, torch.ones_like(x))


class _FlatToy(torch.nn.Module):
    """CPU stand-in for an engine module: parameters 'blocks.0.w', 'blocks.1.w', 'norm.w' over one flat
    gradient buffer (the real EngineModule plumbing, no kernels)."""

    def __new__(cls):
        from OATrans.engine.module import EngineModule

        class Toy(EngineModule):
            def __init__(self):
                super().__init__()
                self.blocks = torch.nn.ModuleList([torch.nn.Linear(4, 4, bias=False) for _ in range(2)])
                self.norm = torch.nn.LayerNorm(4)
        return Toy()


def _overlap_worker(rank, world, port, q):
    from OATrans.parallel import GradSync
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(1)
    toy = _FlatToy()
    loose = torch.nn.Parameter(torch.zeros(5))
    model = torch.nn.ModuleDict({"toy": toy})
    model.register_parameter("loose", loose)
    sync = GradSync(model, overlap=True)
    assert toy.grad_ready_hook is not None
    views = toy._grad_views()
    loose.grad = torch.zeros(5)
    for step in range(2):                                   # second step: state was reset by all_reduce()
        for i, (n, v) in enumerate(views.items()):
            v.fill_(float((rank + 1) * (i + 1) + step))
        loose.grad.fill_(float(10 * (rank + 1)))
        toy._announce(("blocks.1.", "norm."))               # top block + final norm: contiguous tail
        toy._announce(("blocks.0.", "norm."))               # NOT contiguous (blocks.1 lies between): ignored
        assert len(sync._pending) == 1
        toy._announce(("blocks.0.",))
        assert len(sync._pending) == 2
        sync.all_reduce(average=True)
        assert not sync._pending and not sync._covered and sync.started_last_step == 2      # what bench.py reports as async_all_reduces_per_step
        for i, (n, v) in enumerate(views.items()):
            want = sum((r + 1) * (i + 1) + step for r in range(world)) / world
            assert torch.allclose(v, torch.full_like(v, want)), (n, v, want)     # reduced exactly once
        assert torch.allclose(loose.grad, torch.full((5,), sum(10.0 * (r + 1) for r in range(world)) / world))
    q.put(rank)
    dist.barrier()
    dist.destroy_process_group()


def test_forced_single_rank_sync_issues_the_collectives_and_leaves_the_gradients_alone():
    """GradSync(force=True) on a ONE-rank group - what bench.py's `w1_forced` leg and tests/test_dist_gpu.py run on the GPU: the announced
    ranges go out as asynchronous all-reduces (identity at one rank), the rest synchronously, and no 1/W pass touches the gradients
    (at W > 1 the mean is applied to the loss, HipDataParallel.backward, so the path being measured has none either)."""
    from OATrans.parallel import GradSync
    port = _free_port()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        torch.manual_seed(1)
        toy = _FlatToy()
        model = torch.nn.ModuleDict({"toy": toy})
        plain = GradSync(model, overlap=True)                      # one rank, not forced: inert
        assert toy.grad_ready_hook is None and plain.bwd_nt_grid == 0
        sync = GradSync(model, overlap=True, force=True)
        assert toy.grad_ready_hook is not None
        views = toy._grad_views()
        for i, v in enumerate(views.values()):
            v.fill_(float(i + 1))
        before = toy.flat_grad().clone()
        toy._announce(("blocks.1.", "norm."))
        toy._announce(("blocks.0.",))
        assert len(sync._pending) == 2
        sync.all_reduce(average=True)
        assert sync.started_last_step == 2 and not sync._pending
        assert torch.equal(toy.flat_grad(), before)                # sum over one rank, no division pass
        toy.grad_ready_hook = None
    finally:
        dist.destroy_process_group()


def test_two_rank_overlapped_gradient_sync():
    """Ranges announced during backward are all-reduced asynchronously; the final sync covers the rest and
    every element is averaged exactly once (parallel.GradSync.on_ready / _uncovered)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [0, 1]


def _four_rank_worker(rank, world, port, q, grad_dtype):
    """W = 4, a bucket size (7 elements) that divides no range, ranges announced in two pieces + loose parameters."""
    from OATrans.parallel import GradSync
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(1)
    toy = _FlatToy()
    loose = torch.nn.Parameter(torch.zeros(23))
    model = torch.nn.ModuleDict({"toy": toy})
    model.register_parameter("loose", loose)
    sync = GradSync(model, overlap=True, bucket_elems=7, grad_dtype=grad_dtype)
    views = toy._grad_views()
    loose.grad = torch.zeros(23)
    g = torch.Generator().manual_seed(5)
    per_rank = [{n: torch.randn(v.shape, generator=g) for n, v in views.items()} for _ in range(world)]
    loose_rank = [torch.randn(23, generator=g) for _ in range(world)]
    for step in range(2):
        for n, v in views.items():
            v.copy_(per_rank[rank][n])
        loose.grad.copy_(loose_rank[rank])
        toy._announce(("blocks.1.", "norm."))               # overlapped piece; blocks.0 + loose are left to all_reduce()
        sync.all_reduce(average=True)
        tol = 0.0 if grad_dtype == torch.float32 else (world + 1) * 2.0 ** -9
        for n, v in views.items():
            want = sum(per_rank[r][n] for r in range(world)) / world
            bound = tol * sum(per_rank[r][n].abs() for r in range(world)) / world + 1e-6
            assert ((v - want).abs() <= bound).all(), (n, (v - want).abs().max())
        want = sum(loose_rank) / world
        bound = tol * sum(l.abs() for l in loose_rank) / world + 1e-6
        assert ((loose.grad - want).abs() <= bound).all()
    q.put(rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("grad_dtype", [torch.float32, torch.bfloat16])
def test_four_rank_gradient_sync_with_ragged_buckets(grad_dtype):
    """4 ranks over gloo: every gradient element is averaged exactly once although the 7-element buckets split every
    range raggedly; with the bf16 exchange the result stays inside the stated (W + 1) * 2^-9 bound."""
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_four_rank_worker, args=(r, world, port, q, grad_dtype)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [0, 1, 2, 3]


def _eight_rank_worker(rank, world, port, q):
    """W = 8 (config 4's rank count): the engine's announcement order - top block (+ final norm) first, then the block
    below it, each announced range all-reduced asynchronously in 5-element buckets whose last one is ragged - then the
    final sync for the loose parameters; three steps so that the per-step reset of the pending / covered state is seen."""
    from OATrans.parallel import GradSync
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(1)
    toy = _FlatToy()
    loose = torch.nn.Parameter(torch.zeros(13))
    model = torch.nn.ModuleDict({"toy": toy})
    model.register_parameter("loose", loose)
    sync = GradSync(model, overlap=True, bucket_elems=5)
    views = toy._grad_views()
    loose.grad = torch.zeros(13)
    g = torch.Generator().manual_seed(11)
    for step in range(3):
        per_rank = [{n: torch.randn(v.shape, generator=g) for n, v in views.items()} for _ in range(world)]
        loose_rank = [torch.randn(13, generator=g) for _ in range(world)]
        for n, v in views.items():
            v.copy_(per_rank[rank][n])
        loose.grad.copy_(loose_rank[rank])
        toy._announce(("blocks.1.", "norm."))               # backward runs top to bottom: the same order on every rank
        n_first = len(sync._pending)
        toy._announce(("blocks.0.",))
        assert n_first >= 1 and len(sync._pending) > n_first
        sync.all_reduce(average=True)
        assert not sync._pending and not sync._covered
        for n, v in views.items():
            want = sum(per_rank[r][n] for r in range(world)) / world
            assert torch.allclose(v, want, atol=1e-6, rtol=1e-6), (n, (v - want).abs().max())      # averaged exactly once
        assert torch.allclose(loose.grad, sum(loose_rank) / world, atol=1e-6, rtol=1e-6)
    q.put(rank)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_gradient_sync_in_announcement_order():
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eight_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == list(range(8))


def _broadcast_worker(rank, world, port, q):
    from OATrans.parallel import HipDataParallel
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(1000 + rank)                         # every rank initialises DIFFERENTLY (no seed in the entry points)
    toy = _FlatToy()
    toy.flatten_parameters()
    model = torch.nn.ModuleDict({"toy": toy, "head": torch.nn.Linear(4, 3)})
    model.register_buffer("stat", torch.randn(5))
    before = torch.cat([p.detach().flatten() for p in model.parameters()] + [model.stat.flatten()]).clone()
    dp = HipDataParallel(model)                            # DDP broadcasts rank 0's parameters at construction; so does this
    after = torch.cat([p.detach().flatten() for p in model.parameters()] + [model.stat.flatten()])
    flat_ok = all(p.data_ptr() >= toy._flat_param.data_ptr() and
                  p.data_ptr() < toy._flat_param.data_ptr() + 4 * toy._flat_param.numel() for _, p in toy._engine_params())
    q.put((rank, before.numpy().copy(), after.numpy().copy(), flat_ok))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_wrapper_broadcasts_rank0_parameters():
    """base_trainer.py:20-23 of the reference wraps the model in DistributedDataParallel, which broadcasts rank 0's
    parameters and buffers at construction.  HipDataParallel must do the same: the entry points set no seed, so the
    randomly initialised projections / temporal embeddings would otherwise differ per rank for the whole run."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_broadcast_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, a0, ok0), (_, b1, a1, ok1) = res
    assert ok0 and ok1                                      # parameters still live in the flat engine buffer
    assert not (b0 == b1).all()                             # the ranks really started apart
    assert (a0 == b0).all() and (a1 == b0).all()            # ... and both now hold rank 0's values
