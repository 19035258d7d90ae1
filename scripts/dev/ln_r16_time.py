"""Dev (GPU): the LayerNorm kernels of a ViT block alone, fp32 stream vs bf16 stream, on ROTATING buffers (4 sets, 2.5 GB:
nothing survives in the 256 MB MALL between two calls, as in the step where GEMMs run in between)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M, D, R = int(os.environ.get("M", 50208)), 768, 4
z16 = lambda: [torch.randn(M, D, device="cuda").bfloat16() for _ in range(R)]
z32 = lambda: [torch.randn(M, D, device="cuda") for _ in range(R)]
x32, o32, x16, o16, a16, b16, y16, d16, g16 = z32(), z32(), z16(), z16(), z16(), z16(), z16(), z16(), z16()
mean = torch.empty(M, device="cuda"); rstd = torch.rand(M, device="cuda") + 0.5
def timeit(fn, n=24):
    for k in range(4): fn(k % R)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for k in range(n): fn(k % R)
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3
u = M * D * 2 / 1e6     # one bf16 pass in MB
cases = [
 ("fwd norm3 fp32 stream (add2)", 7, lambda k: hip.add2_layernorm_fwd(x32[k], a16[k], b16[k], o32[k], None, None, M, D, 1e-6, y=y16[k], mean=mean, rstd=rstd)),
 ("fwd norm3 bf16 stream", 5, lambda k: hip.layernorm_fwd_r16(x16[k], M, D, 1e-6, add_a=a16[k], add_b=b16[k], sum16=o16[k], y=y16[k], mean=mean, rstd=rstd)),
 ("fwd norm1 fp32 stream (add)", 4, lambda k: hip.add_layernorm_fwd(x32[k], a16[k], None, None, None, M, D, 1e-6, y=y16[k], mean=mean, rstd=rstd)),
 ("fwd norm1 bf16 stream", 3, lambda k: hip.layernorm_fwd_r16(x16[k], M, D, 1e-6, add_a=a16[k], y=y16[k], mean=mean, rstd=rstd)),
 ("bwd norm2 fp32 G", 6, lambda k: hip.layernorm_bwd_xhat(d16[k], y16[k], rstd, M, D, dx16=g16[k], dres=x32[k], dxp16=o16[k])),
 ("bwd norm2 bf16 G", 4, lambda k: hip.layernorm_bwd_xhat(d16[k], y16[k], rstd, M, D, dx16=g16[k], add_a=a16[k])),
 ("bwd norm1", 3, lambda k: hip.layernorm_bwd_xhat(d16[k], y16[k], rstd, M, D, dx16=g16[k])),
 ("bwd norm3 fp32 G", 9, lambda k: hip.layernorm_bwd_xhat(d16[k], y16[k], rstd, M, D, dx=x32[k], dx16=g16[k], dres=x32[k], add_a=a16[k], add_b=b16[k])),
 ("bwd norm3 bf16 G", 5, lambda k: hip.layernorm_bwd_xhat(d16[k], y16[k], rstd, M, D, dx16=g16[k], add_a=a16[k], add_b=b16[k])),
]
for rep in range(2):
    for name, units, fn in cases:
        t = timeit(fn)
        print(f"{name:32s} {t:7.1f} us  {units * u:6.0f} MB  {units * u / t:6.2f} TB/s", flush=True)
