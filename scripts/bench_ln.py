"""Dev: LayerNorm forward / backward alone on the GPU at the ViT-B/16 step shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M, D = 50208, 768
x = torch.randn(M, D, device="cuda"); br = torch.randn(M, D, device="cuda").bfloat16(); s32 = torch.empty_like(x)
g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
y = torch.empty(M, D, device="cuda", dtype=torch.bfloat16); mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3
for rep in range(3):
    t = timeit(lambda: hip.add_layernorm_fwd(x, br, s32, g, b, M, D, 1e-6, y=y, mean=mean, rstd=rstd))
    print(f"add_layernorm_fwd {t:7.1f} us  {(M*D*(4+2+4+2))/t/1e6:6.2f} TB/s")
    t = timeit(lambda: hip.layernorm_fwd(x, g, b, M, D, 1e-6, y=y, mean=mean, rstd=rstd))
    print(f"layernorm_fwd     {t:7.1f} us  {(M*D*(4+2))/t/1e6:6.2f} TB/s")
dy = torch.randn(M, D, device="cuda").bfloat16(); G = torch.randn(M, D, device="cuda"); dx16 = torch.empty_like(y)
dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
for rep in range(3):
    t = timeit(lambda: hip.layernorm_bwd(dy, x, mean, rstd, g, M, D, dx=G, dx16=dx16, dres=G, dgamma=dg, dbeta=db, accumulate=False))
    print(f"layernorm_bwd     {t:7.1f} us  {(M*D*(2+4+4+4+2))/t/1e6:6.2f} TB/s")
