"""Dev: do the attention kernels pay for the [M, 3 D] row layout (a head's q / k / v slices are 128-byte pieces 4.6 KB apart)?  The same number of
problems and bytes with H = 1 and 12 x the samples: rows of 384 bytes, every line of a row used by the one head."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (B, H) in [(32, 12), (384, 1)]:
    T, N = 8, 196
    D = H * 64; M = B * T * N + B; Mp = (M + 255) // 256 * 256
    qkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda"); qkv[:M] = torch.randn(M, 3 * D, device="cuda").bfloat16()
    out = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(Mp, H, device="cuda")
    dout = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda"); dout[:M] = torch.randn(M, D, device="cuda").bfloat16()
    dqkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda"); side = torch.zeros(B, H, 3, 64, device="cuda")
    hip.attn_cls_fwd(qkv, out, lse, B, T, N, H, D, 0.125)
    for rep in range(2):
        r = [timeit(lambda: hip.attn_space_fwd(qkv, out, lse, B, T, N, H, D, 0.125)),
             timeit(lambda: hip.attn_time_fwd(qkv, out, lse, B, T, N, H, D, 0.125)),
             timeit(lambda: hip.attn_space_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, 0.125)),
             timeit(lambda: hip.attn_time_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, 0.125))]
        print(f"B {B:3d} H {H:2d} (row = {6 * D} bytes): space fwd {r[0]:6.1f}  time fwd {r[1]:6.1f}  space bwd {r[2]:6.1f}  time bwd {r[3]:6.1f} us")
    del qkv, out, lse, dout, dqkv
