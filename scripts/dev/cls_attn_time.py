"""Dev: time of the CLS-query attention kernels at the headline shape (B=32, T=8, N=196, H=12), alone on the GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
B, T, N, H = 32, 8, 196, 12; D = 768; M = B * T * N + B
qkv = torch.randn(M, 3 * D, device="cuda").bfloat16(); q32 = torch.randn(B, D, device="cuda")
out = torch.zeros(M, D, device="cuda", dtype=torch.bfloat16); lse = torch.zeros(M, H, device="cuda"); o32 = torch.zeros(B, D, device="cuda")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3
print(f"attn_cls_fwd_dual {t(lambda: hip.attn_cls_fwd_dual(qkv, out, lse, q32, o32, B, T, N, H, D, 0.125)):.1f} us   "
      f"attn_cls_fwd {t(lambda: hip.attn_cls_fwd(qkv, out, lse, B, T, N, H, D, 0.125)):.1f} us   "
      f"attn_time_fwd {t(lambda: hip.attn_time_fwd(qkv, out, lse, B, T, N, H, D, 0.125)):.1f} us")
