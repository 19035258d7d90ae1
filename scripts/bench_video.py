"""Dev microbench: video encoder fwd+bwd only (random weights), prints ms/step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.model.video_transformer import SpaceTimeTransformer
B = int(os.environ.get("B", 32)); T = int(os.environ.get("T", 8)); steps = int(os.environ.get("STEPS", 5))
torch.manual_seed(0)
m = SpaceTimeTransformer(num_frames=T, time_init="rand"); m.head = torch.nn.Identity(); m = m.cuda()
m.need_patch_tokens = False
video = torch.randn(B, T, 3, 224, 224, device="cuda", dtype=torch.bfloat16)
g = torch.randn(B, 768, device="cuda")
def step():
    cls, _ = m(video)
    (cls * g).sum().backward()
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps): step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
gf = 3 * 369.24 * B
print(f"B={B} T={T} {dt*1e3:.2f} ms/step  {B/dt:.1f} clips/s  {gf/dt/1e3:.1f} TFLOP/s algorithmic (video only)")
print(torch.cuda.max_memory_allocated() / 2**30, "GiB peak")
