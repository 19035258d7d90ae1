"""Dev: where a config-3 step spends its time on the main stream (HIP events, nothing under a profiler):
forward (incl. the join with the text stream) | all-gather + losses | backward | gradient sync + AdamW."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from OATrans.model.layers import sim_matrix
from OATrans.model.oa_layers import mean_rows, bce_sum
from OATrans.parallel import allgather_packed, allgather_pair
from OATrans.trainer.step import _nce
variant = os.environ.get("VARIANT", "global_local")
args = argparse.Namespace(variant=variant, frames=8, res=224, batch=32, lr=2e-5, dtype="bf16")
dev = torch.device("cuda:0")
dp, opt, loss_fn = bench.build(args, dev)
data = bench.synthetic_batch(args, 0, dev)
sa = argparse.Namespace(world_size=1, rank=0, local_rank=0)
ev = lambda: torch.cuda.Event(enable_timing=True)
acc = [0.0] * 4
N = 12
SYNC = os.environ.get("SYNC_EACH", "0") == "1"     # 0: the host runs ahead as in bench.py, events are read after the last step
allev = []
for it in range(N + 4):
    e = [ev() for _ in range(5)]
    allev.append(e)
    core = dp.module
    core.begin_step(); opt.zero_grad()
    e[0].record()
    if variant == "global_local":
        text, pad_text, video, pad_video, extra = dp(data)
        e[1].record()
        region_feat, tags_feat = extra[4], extra[5]
        video, pad_text, pad_video, text, region_feat, tags_feat = allgather_packed([video, pad_text, pad_video, text, region_feat, tags_feat], sa)
        loss = _nce(loss_fn, text, video) + _nce(loss_fn, pad_text, video)
        loss = loss + _nce(loss_fn, mean_rows(region_feat), mean_rows(tags_feat))
    else:
        t, v = dp(data, aug=True)
        e[1].record()
        va, ta = allgather_pair(v, t, sa)
        loss = loss_fn(sim_matrix(ta, va))
    e[2].record()
    dp.backward(loss)
    e[3].record()
    dp.sync_gradients(); opt.step()
    e[4].record()
    if SYNC:
        torch.cuda.synchronize()
torch.cuda.synchronize()
for e in allev[4:]:
    for k in range(4): acc[k] += e[k].elapsed_time(e[k + 1])
print("step period %.2f ms" % (allev[4][0].elapsed_time(allev[-1][0]) / (N - 1)))
print(variant, "ms per step: forward %.2f | gather + loss %.2f | backward %.2f | sync + AdamW %.2f | sum %.2f" % tuple([a / N for a in acc] + [sum(acc) / N]))
