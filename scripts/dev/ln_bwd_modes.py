"""Dev: the three folded LayerNorm-backward forms of a block (norm2: reads G, writes gb + dx2; norm1: writes gc only; norm3: reads
G + both increments, writes G + ga) and the old read-modify-write form, alone on the GPU, at M = 50208."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M, D = 50208, 768
Mp = (M + 255) // 256 * 256
z16 = lambda: torch.randn(Mp, D, device="cuda").bfloat16()
dxh, xh, gb, dx2, gc, ga = z16(), z16(), z16(), z16(), z16(), z16()
G = torch.randn(Mp, D, device="cuda"); rstd = torch.rand(Mp, device="cuda") + 0.5
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
forms = {"old rmw  (539 MB)": lambda: hip.layernorm_bwd_xhat(dxh, xh, rstd, M, D, dx=G, dx16=gb, dres=G),
         "norm2    (462 MB)": lambda: hip.layernorm_bwd_xhat(dxh, xh, rstd, M, D, dx16=gb, dres=G, dxp16=dx2),
         "norm1    (231 MB)": lambda: hip.layernorm_bwd_xhat(dxh, xh, rstd, M, D, dx16=gc),
         "norm3    (693 MB)": lambda: hip.layernorm_bwd_xhat(dxh, xh, rstd, M, D, dx=G, dx16=ga, dres=G, add_a=dx2, add_b=gc)}
for r in range(2):
    for k, f in forms.items():
        t = timeit(f)
        mb = float(k.split("(")[1].split()[0])
        print(f"OAT_LN_BWDX_BLOCKS={os.environ.get('OAT_LN_BWDX_BLOCKS', 'default')} {k}: {t:6.1f} us  {mb / t:5.2f} TB/s")
