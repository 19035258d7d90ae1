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
# round 2: fp8 forward GEMM, MLP-pair epilogues, LayerNorm
st = torch.zeros(2, 3, device="cuda")
for (n, k) in [(2304, 768), (768, 3072)]:
    A = rb(Mp, k); W = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16(); o = torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16)
    A8 = torch.empty(Mp, k, dtype=torch.uint8, device="cuda"); W8 = torch.empty(n, k, dtype=torch.uint8, device="cuda")
    for x, x8, r in ((A, A8, 0), (W, W8, 1)):
        st[r].zero_(); hip.fp8_amax(x, x.shape[0], k, st[r, 0:1]); hip.fp8_update_scales(st[r, 0:1], st[r, 1:2], st[r, 2:3], 1)
        hip.fp8_quant(x, x8, x.shape[0], k, st[r, 1:2])
    for _ in range(3): hip.gemm_nt_f8(A8, W8, M, n, k, hip.EPI_BF16, o, st[0, 2:3], st[1, 2:3])
A = rb(Mp, D); W = (torch.randn(4 * D, D, device="cuda") * D ** -0.5).bfloat16(); bias = torch.randn(4 * D, device="cuda")
h8 = torch.zeros(Mp, 4 * D, dtype=torch.uint8, device="cuda"); g = torch.zeros(Mp, 4 * D, dtype=torch.bfloat16, device="cuda"); dh = torch.zeros_like(g)
for _ in range(3):
    hip.gemm_nt(A, W, M, 4 * D, D, hip.EPI_GELU_GRAD | hip.EPI_U8, h8, out2=g, bias=bias)
    hip.gemm_nt(A, W, M, 4 * D, D, hip.EPI_MUL_AUX | hip.EPI_U8, dh, aux=h8)
x = torch.randn(Mp, D, device="cuda"); br = rb(Mp, D); s32 = torch.zeros(Mp, D, device="cuda"); y = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda")
gam = torch.ones(D, device="cuda"); bet = torch.zeros(D, device="cuda"); mean = torch.zeros(Mp, device="cuda"); rstd = torch.zeros(Mp, device="cuda")
G = torch.zeros(Mp, D, device="cuda"); dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
for _ in range(3):
    hip.add_layernorm_fwd(x, br, s32, gam, bet, M, D, 1e-6, y=y, mean=mean, rstd=rstd)
    hip.layernorm_bwd(br, s32, mean, rstd, gam, M, D, dx=G, dx16=y, dres=G, dgamma=dg, dbeta=db)
# round 4: the grouped weight gradients of one ViT block (gemm_tn_sk), the N = 768 GEMM shapes, the LayerNorms of the bf16 stream
def wprob(n1, n2):
    P = (torch.randn(Mp, n1, device="cuda") * 0.5).bfloat16(); Q = (torch.randn(Mp, n2, device="cuda") * 0.5).bfloat16()
    return (P, Q, M, n1, n2, torch.zeros(n1, n2, device="cuda"), torch.zeros(n1, device="cuda"), False)
blk = [wprob(D, 4 * D), wprob(4 * D, D), wprob(3 * D, D), wprob(3 * D, D), wprob(D, D), wprob(D, D)]
grp = hip.TnGroup(blk, layers=[[0, 1, 2, 3], [4, 5]])
for _ in range(3): grp.run()
del blk, grp
for (n, k) in [(768, 768), (768, 2304), (3072, 768)]:
    A = rb(Mp, k); W = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16(); o = torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(n, device="cuda")
    for _ in range(3): hip.gemm_nt(A, W, M, n, k, hip.EPI_BF16, o, bias=bias)
x16 = rb(Mp, D); a16 = rb(Mp, D); b16 = rb(Mp, D); o16 = torch.zeros_like(x16); d16 = rb(Mp, D)
for _ in range(3):
    hip.layernorm_fwd_r16(x16, M, D, 1e-6, add_a=a16, add_b=b16, sum16=o16, y=y, mean=mean, rstd=rstd)
    hip.layernorm_bwd_xhat(d16, y, rstd, M, D, dx16=o16, add_a=a16, add_b=b16)
torch.cuda.synchronize()
