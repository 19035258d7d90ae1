"""Dev (GPU): one library per ablation (compile-time OAT_ATTN_ABL, built from a patched COPY of csrc/attn_space.hip outside
the product tree): times the three MFMA attention kernels of the loaded library.  Run once per OAT_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
B, T, N, H = 32, 8, 196, 12
D = H * 64; M = B * T * N + B; Mp = (M + 255) // 256 * 256
R = 3
mk = lambda c: [torch.zeros(Mp, c, dtype=torch.bfloat16, device="cuda") for _ in range(R)]
qkv, out, dout, dqkv = mk(3 * D), mk(D), mk(D), mk(3 * D)
for r in range(R):
    qkv[r][:M] = torch.randn(M, 3 * D, device="cuda").bfloat16(); dout[r][:M] = torch.randn(M, D, device="cuda").bfloat16()
    out[r][:M] = torch.randn(M, D, device="cuda").bfloat16()
lse = [torch.randn(Mp, H, device="cuda") + 5 for _ in range(R)]
side = torch.zeros(B, H, 3, 64, device="cuda")
sc = 0.125
def timeit(fn, n=12):
    for k in range(3): fn(k % R)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(n): fn(k % R)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
tf = min(timeit(lambda k: hip.attn_space_fwd(qkv[k], out[k], lse[k], B, T, N, H, D, sc)) for _ in range(2))
tb = min(timeit(lambda k: hip.attn_space_bwd(qkv[k], out[k], lse[k], dout[k], dqkv[k], side, B, T, N, H, D, sc)) for _ in range(2))
tt = min(timeit(lambda k: hip.attn_time_bwd(qkv[k], out[k], lse[k], dout[k], dqkv[k], side, B, T, N, H, D, sc)) for _ in range(2))
print(f"{os.environ.get('TAG', 'product'):24s} space fwd {tf:6.1f} us   space bwd {tb:6.1f} us   time bwd {tt:6.1f} us", flush=True)
