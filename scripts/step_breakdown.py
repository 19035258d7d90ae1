"""Dev: where one training step's wall time goes (HIP events on the main stream between the phases of hot_step)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd"))
import torch
import bench
from OATrans.model.layers import sim_matrix
from OATrans.parallel import allgather_pair
args = argparse.Namespace(variant="frozen", frames=8, batch=32, res=224, lr=2e-5)
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dp, opt, loss_fn = bench.build(args, dev)
data = bench.synthetic_batch(args, 0, dev)
sa = argparse.Namespace(world_size=1, rank=0, local_rank=0)
names = ["forward (both towers)", "gather + sim + loss", "backward", "optimizer"]
tot = [0.0] * 4
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(13):
    core = dp.module
    core.begin_step(); opt.zero_grad()
    e0 = ev()
    t, v = dp(data, aug=True)
    e1 = ev()
    va, ta = allgather_pair(v, t, sa)
    loss = loss_fn(sim_matrix(ta, va))
    e2 = ev()
    dp.backward(loss); dp.sync_gradients()
    e3 = ev()
    opt.step()
    e4 = ev()
    torch.cuda.synchronize()
    if it >= 3:
        for i, (a, b) in enumerate(((e0, e1), (e1, e2), (e2, e3), (e3, e4))):
            tot[i] += a.elapsed_time(b) / 10
for n, t in zip(names, tot):
    print(f"{n:28s} {t:7.2f} ms")
print(f"{'sum':28s} {sum(tot):7.2f} ms")
