"""Dev: host time of one training step with / without launch tapes, and of the bare tape replays."""
import os, sys, time, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from OATrans.ops import hip
from OATrans.trainer.step import hot_step as _hs, global_local_step, region_mem_step

VAR = os.environ.get("VARIANT", "frozen")
hot_step = {"frozen": _hs, "global_local": global_local_step, "region_mem": region_mem_step}[VAR]
args = argparse.Namespace(variant=VAR, frames=8, res=224, batch=32, lr=2e-5, dtype="bf16")
dev = torch.device("cuda:0")
dp, opt, loss_fn = bench.build(args, dev)
data = bench.synthetic_batch(args, 0, dev)
sa = argparse.Namespace(world_size=1, rank=0, local_rank=0)
for taped in (True, False, True):
    for sub in (dp.module.video_model, dp.module.text_model):
        sub._engine.use_tape = taped
    for _ in range(3):
        hot_step(dp, loss_fn, opt, data, sa)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        hot_step(dp, loss_fn, opt, data, sa)
    host = (time.perf_counter() - t0) / 5
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 5
    print(f"tape={taped}: host {host*1e3:.2f} ms/step, wall {tot*1e3:.2f} ms/step")
eng = dp.module.video_model._engine
pl = next(iter(eng.plans.values()))
for name in ("tape_fwd", "tape_bwd"):
    key, tid, out, nseg = getattr(pl, name)
    n = hip.lib().oat_tape_ops(tid)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hip.tape_replay(tid)
    h = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name}: {n} ops, {nseg} segments, replay host {h*1e3:.2f} ms = {h/n*1e6:.1f} us/op")
import cProfile, pstats
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(3):
    hot_step(dp, loss_fn, opt, data, sa)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
