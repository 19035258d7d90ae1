"""Dev: a few launches of each hot kernel at the bench shapes, alone on the GPU, for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
B, T, N, H = 32, 8, 196, 12
D = H * 64; M = B * T * N + B; Mp = (M + 255) // 256 * 256
rb = lambda r, c: (torch.randn(r, c, device="cuda")).bfloat16()
for (n, k) in [(2304, 768), (768, 3072)]:
    A = rb(Mp, k); W = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16(); o = torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(n, device="cuda")
    for _ in range(3): hip.gemm_nt(A, W, M, n, k, hip.EPI_BF16, o, bias=bias)
    P = rb(Mp, n); out = torch.zeros(n, k, device="cuda"); bo = torch.zeros(n, device="cuda")
    for _ in range(3): hip.gemm_tn(P, A, M, n, k, out, bias_out=bo)
qkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda"); qkv[:M] = rb(M, 3 * D)
out = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(Mp, H, device="cuda")
dout = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda"); dout[:M] = rb(M, D)
dqkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda"); side = torch.zeros(B, H, 3, 64, device="cuda")
for _ in range(3):
    hip.attn_space_fwd(qkv, out, lse, B, T, N, H, D, 0.125); hip.attn_cls_fwd(qkv, out, lse, B, T, N, H, D, 0.125)
    hip.attn_space_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, 0.125)
    hip.attn_time_fwd(qkv, out, lse, B, T, N, H, D, 0.125)
    hip.attn_time_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, 0.125)
torch.cuda.synchronize()
