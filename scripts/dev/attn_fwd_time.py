"""Dev: time of the forward attention kernels at the headline shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
B, T, N, H, D = 32, 8, int(os.environ.get("N", 196)), 12, 768
M = B * T * N + B
Mp = (M + 255) // 256 * 256
qkv = (torch.randn(Mp, 3 * D, device="cuda") * 0.5).bfloat16()
out = torch.zeros(Mp, D, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(Mp, H, device="cuda")
def timeit(fn, n=20):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3
for name, fn in (("space_fwd", hip.attn_space_fwd), ("time_fwd", hip.attn_time_fwd), ("cls_fwd", hip.attn_cls_fwd)):
    ts = sorted(timeit(lambda: fn(qkv, out, lse, B, T, N, H, D, 0.125)) for _ in range(5))
    print(f"{name}: {ts[2]:.1f} us (min {ts[0]:.1f})")
print("checksum", out.float().abs().sum().item())
q32 = torch.randn(B, D, device="cuda"); o32 = torch.zeros(B, D, device="cuda")
ts = sorted(timeit(lambda: hip.attn_cls_fwd_dual(qkv, out, lse, q32, o32, B, T, N, H, D, 0.125)) for _ in range(5))
print(f"cls_fwd_dual: {ts[2]:.1f} us (min {ts[0]:.1f})")
side = torch.cuda.Stream()
def both():
    ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
    with torch.cuda.stream(side):
        hip.attn_cls_fwd_dual(qkv, out, lse, q32, o32, B, T, N, H, D, 0.125)
        d = torch.cuda.Event(); d.record()
    hip.attn_space_fwd(qkv, out, lse, B, T, N, H, D, 0.125)
    torch.cuda.current_stream().wait_event(d)
ts = sorted(timeit(both) for _ in range(5))
print(f"space_fwd beside cls_fwd_dual: {ts[2]:.1f} us")
