"""Dev: MFMA-bound GEMM (persistent grid of G workgroups) on one stream beside the memory-bound backward kernels (space / TIME attention
backward, LayerNorm backward) on another: does the pair finish sooner than one after the other?  Weight-gradient group of one ViT block
as the GEMM (TnGroup with a reduced grid) and the forward / data-gradient ping-pong GEMM."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
B, T, N, H = 32, 8, 196, 12
D = H * 64; M = B * T * N + B; Mp = (M + 255) // 256 * 256
rb = lambda r, c, s=1.0: (torch.randn(r, c, device="cuda") * s).bfloat16()
# memory-bound side
qkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda"); qkv[:M] = rb(M, 3 * D)
out = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(Mp, H, device="cuda")
dout = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda"); dout[:M] = rb(M, D)
dqkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda"); side = torch.zeros(B, H, 3, 64, device="cuda")
hip.attn_space_fwd(qkv, out, lse, B, T, N, H, D, 0.125); hip.attn_cls_fwd(qkv, out, lse, B, T, N, H, D, 0.125)
x16 = rb(Mp, D); a16 = rb(Mp, D); b16 = rb(Mp, D); o16 = torch.zeros_like(x16); d16 = rb(Mp, D); y = rb(Mp, D)
rstd = torch.rand(Mp, device="cuda") + 0.5
def mem_chain():          # the memory-bound kernels of one block's backward
    hip.layernorm_bwd_xhat(d16, y, rstd, M, D, dx16=o16, add_a=a16, add_b=b16)
    hip.attn_space_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, 0.125)
    hip.layernorm_bwd_xhat(d16, y, rstd, M, D, dx16=o16, add_a=a16)
    hip.attn_time_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, 0.125)
    hip.layernorm_bwd_xhat(d16, y, rstd, M, D, dx16=o16, add_a=a16, add_b=b16)
# MFMA side: the weight gradients of one block
def wprob(n1, n2):
    return (rb(Mp, n1, 0.5), rb(Mp, n2, 0.5), M, n1, n2, torch.zeros(n1, n2, device="cuda"), torch.zeros(n1, device="cuda"), False)
blk = [wprob(D, 4 * D), wprob(4 * D, D), wprob(3 * D, D), wprob(3 * D, D), wprob(D, D), wprob(D, D)]
groups = {g: hip.TnGroup(blk, grid=g, layers=[[0, 1, 2, 3], [4, 5]]) for g in (256, 192, 128)}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def wall(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for rep in range(2):
    tm = wall(mem_chain)
    print(f"memory-bound chain alone: {tm:7.1f} us")
    for g, grp in groups.items():
        tg = wall(grp.run)
        def both():
            with torch.cuda.stream(s1): grp.run()
            with torch.cuda.stream(s2): mem_chain()
        def both_rev():
            with torch.cuda.stream(s2): mem_chain()
            with torch.cuda.stream(s1): grp.run()
        tb, tr = wall(both), wall(both_rev)
        print(f"  wgrad grid {g}: alone {tg:7.1f} us ({grp.grid} wgs); one after the other {tg + tm:7.1f}; side by side {tb:7.1f} (GEMM issued first) / {tr:7.1f} (chain first)  -> {100 * (1 - min(tb, tr) / (min(wall(groups[256].run), tg) + tm)):+.1f} % vs best sequential")
