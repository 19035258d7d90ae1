"""Dev: the DistilBERT tower alone on the GPU (B = 32, L = 32: the headline's text side) - forward and backward wall
time per call with nothing beside it, and (under rocprofv3 --kernel-trace --stats) its per-kernel times undistorted by
the video tower's kernels.  TRAIN=1 runs with dropout (the trainers' mode)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.model.text_transformer import DistilBertHIP
B = int(os.environ.get("B", 32)); L = int(os.environ.get("L", 32)); steps = int(os.environ.get("STEPS", 20))
torch.manual_seed(0)
m = DistilBertHIP().cuda()
m.flatten_parameters()
if os.environ.get("TRAIN", "1") == "1":
    m.train()
else:
    m.eval()
ids = torch.randint(1000, 30000, (B, L), device="cuda"); mask = torch.ones(B, L, dtype=torch.int64, device="cuda")
g = torch.randn(B, L, 768, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for it in range(steps + 3):
    m.begin_step()
    ev[0].record()
    h = m(input_ids=ids, attention_mask=mask).last_hidden_state
    ev[1].record()
    h.backward(g)
    ev[2].record()
    torch.cuda.synchronize()
    if it >= 3:
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print(f"text tower alone B={B} L={L}: forward {tf/steps*1e3:.0f} us, backward {tb/steps*1e3:.0f} us per call")
