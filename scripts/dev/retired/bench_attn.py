import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
B, T, N, H = int(os.environ.get("B", 32)), 8, int(os.environ.get("N", 196)), 12
D = H * 64; M = B * T * N + B; Mp = (M + 255) // 256 * 256
qkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda"); qkv[:M] = torch.randn(M, 3 * D, device="cuda").bfloat16()
out = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(Mp, H, device="cuda")
dout = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda"); dout[:M] = torch.randn(M, D, device="cuda").bfloat16()
dqkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda"); side = torch.zeros(B, H, 3, 64, device="cuda")
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
sc = 0.125
for _r in range(3): print("space fwd us", timeit(lambda: hip.attn_space_fwd(qkv, out, lse, B, T, N, H, D, sc)))
print("space fwd us", timeit(lambda: hip.attn_space_fwd(qkv, out, lse, B, T, N, H, D, sc)))
for _r in range(2): print("cls   fwd us", timeit(lambda: hip.attn_cls_fwd(qkv, out, lse, B, T, N, H, D, sc)))
print("cls   fwd us", timeit(lambda: hip.attn_cls_fwd(qkv, out, lse, B, T, N, H, D, sc)))
print("space bwd us", timeit(lambda: hip.attn_space_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, sc)))
for _r in range(3): print("time  fwd us", timeit(lambda: hip.attn_time_fwd(qkv, out, lse, B, T, N, H, D, sc)))
print("time  fwd us", timeit(lambda: hip.attn_time_fwd(qkv, out, lse, B, T, N, H, D, sc)))
print("time  bwd us", timeit(lambda: hip.attn_time_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, sc)))
hip.attn_cls_fwd(qkv, out, lse, B, T, N, H, D, sc); hip.attn_time_fwd(qkv, out, lse, B, T, N, H, D, sc)
for rep in range(2):
    for var in (1, 2, 0):
        hip.lib().oat_attn_time_set_variant(var)
        side.zero_()
        t = timeit(lambda: hip.attn_time_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, sc))
        print(f"time bwd variant {var} ({['mfma', 'two-pass', 'single-read LDS'][var]}): {t:.1f} us  checksum {dqkv.float().abs().sum().item():.6e}")
hip.lib().oat_attn_time_set_variant(0)
hip.attn_space_fwd(qkv, out, lse, B, T, N, H, D, sc)
ref_dq = None
for rep in range(3):
    for var in (0, 1, 2):
        hip.lib().oat_attn_space_set_variant(var)
        side.zero_(); dqkv.zero_()
        hip.attn_space_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, sc)
        torch.cuda.synchronize()
        if ref_dq is None: ref_dq, ref_side = dqkv.clone(), side.clone()
        err = (dqkv.float() - ref_dq.float()).abs().max().item(); serr = ((side - ref_side).abs().max() / ref_side.abs().max()).item()
        t = timeit(lambda: hip.attn_space_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, sc), n=20)
        print(f"space bwd variant {var}: {t:.1f} us  checksum {dqkv[:M - B].float().abs().sum().item():.6e}  max|dqkv - v0| {err:.3e}  side rel {serr:.3e}")
hip.lib().oat_attn_space_set_variant(0)
for gpw in (1, 2, 4, 7, 14, 25):
    hip.lib().oat_attn_time_set_variant(gpw << 8)
    side.zero_()
    t = timeit(lambda: hip.attn_time_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, sc))
    print(f"time bwd mfma gpw={gpw}: {t:.1f} us")
hip.lib().oat_attn_time_set_variant(4 << 8)
