"""Dev: the four patch-attention launches of a ViT block alone on the GPU (B = 32, T = 8, N = 196, H = 12), three rotating
buffer sets, microseconds per launch and checksums (A/B of two library builds: OAT_LIB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
torch.manual_seed(0)
B, T, N, H = 32, 8, 196, 12
D = H * 64; M = B * T * N + B; Mp = (M + 255) // 256 * 256
NS = 3
def mk(cols, rnd=True):
    t = torch.zeros(Mp, cols, dtype=torch.bfloat16, device="cuda")
    if rnd: t[:M] = torch.randn(M, cols, device="cuda").bfloat16()
    return t
qkv = [mk(3 * D) for _ in range(NS)]; out = [mk(D, False) for _ in range(NS)]; lse = [torch.zeros(Mp, H, device="cuda") for _ in range(NS)]
dout = [mk(D) for _ in range(NS)]; dqkv = [mk(3 * D, False) for _ in range(NS)]; side = torch.zeros(B, H, 3, 64, device="cuda")
def timeit(fn, n=30):
    for i in range(6): fn(i % NS)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i % NS)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
sc = 0.125
for rep in range(3):
    a = timeit(lambda i: hip.attn_space_fwd(qkv[i], out[i], lse[i], B, T, N, H, D, sc))
    b = timeit(lambda i: hip.attn_space_bwd(qkv[i], out[i], lse[i], dout[i], dqkv[i], side, B, T, N, H, D, sc))
    cs = dqkv[0][:M - B].float().abs().sum().item()
    for i in range(NS): hip.attn_time_fwd(qkv[i], out[i], lse[i], B, T, N, H, D, sc)
    c = timeit(lambda i: hip.attn_time_fwd(qkv[i], out[i], lse[i], B, T, N, H, D, sc))
    d = timeit(lambda i: hip.attn_time_bwd(qkv[i], out[i], lse[i], dout[i], dqkv[i], side, B, T, N, H, D, sc))
    ct = dqkv[0][:M - B].float().abs().sum().item()
    print(f"space fwd {a:6.1f}  space bwd {b:6.1f}  time fwd {c:6.1f}  time bwd {d:6.1f} us   checksums {cs:.6e} {ct:.6e}", flush=True)
