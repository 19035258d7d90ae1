"""dev (timing only, wrong gradients in the skipped mode): what the SECOND DistilBERT pass of oa_model_global_local (caption + tags,
`pad_text`: oa_model_global_local.py:161-164 of the reference) costs the config-3 step.  Interleaved: the real step against the same
step with the pad_text pass served from a cache (no forward kernels, no backward) - an upper bound on what any treatment of that pass
could return.  Prints ms per step."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from OATrans.trainer.step import global_local_step
args = argparse.Namespace(variant="global_local", frames=8, res=224, batch=32, lr=2e-5, dtype="bf16")
dev = torch.device("cuda:0")
dp, opt, loss_fn = bench.build(args, dev)
data = bench.synthetic_batch(args, 0, dev)
sa = argparse.Namespace(world_size=1, rank=0, local_rank=0)
m = dp.module
state = dict(skip=False, cache=None)
orig_compute, orig_launch = m.compute_text, m.text_model.launch


def launch(input_ids=None, attention_mask=None, **kw):
    if state["skip"] and input_ids is data["pad_text"]["input_ids"]:
        return None
    return orig_launch(input_ids=input_ids, attention_mask=attention_mask, **kw)


def compute_text(text_data, launched=None):
    if state["skip"] and text_data is data["pad_text"]:
        return state["cache"]
    out = orig_compute(text_data, launched=launched)
    if text_data is data["pad_text"]:
        state["cache"] = (out[0].detach(), out[1].detach())
    return out


m.text_model.launch, m.compute_text = launch, compute_text


def run(n=12):
    for _ in range(3):
        global_local_step(dp, loss_fn, opt, data, sa)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        global_local_step(dp, loss_fn, opt, data, sa)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = []
for rnd in range(3):
    state["skip"] = False
    a = run()
    state["skip"] = True
    b = run()
    res.append((a, b))
print("global_local ms/step, both text passes:", [round(a, 2) for a, _ in res], "| second pass served from a cache:", [round(b, 2) for _, b in res], flush=True)
