"""Dev: one measured case per process (CASE env var), a few launches of ONE kernel configuration at the bench shapes, for the
rocprofv3 timing / FETCH_SIZE / WRITE_SIZE passes of scripts/dev/r5_measure.sh ("Next round" item 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
case = os.environ["CASE"]
B, T, N, H = 32, 8, 196, 12
D = H * 64
M = B * T * N + B
REP = 6
rb = lambda r, c, s=1.0: (torch.randn(r, c, device="cuda") * s).bfloat16()
if case.startswith("space_bwd_v"):
    Mp = (M + 255) // 256 * 256
    qkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda"); qkv[:M] = rb(M, 3 * D)
    out = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(Mp, H, device="cuda")
    dout = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda"); dout[:M] = rb(M, D)
    dqkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda"); side = torch.zeros(B, H, 3, 64, device="cuda")
    hip.attn_space_fwd(qkv, out, lse, B, T, N, H, D, 0.125); hip.attn_cls_fwd(qkv, out, lse, B, T, N, H, D, 0.125)
    hip.lib().oat_attn_space_set_variant(int(case[-1]))
    for _ in range(REP):
        hip.attn_space_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, 0.125)
elif case.startswith(("fc1_gelu", "fc2_dgrad", "fc1_plain")):
    m = int(case.split("_M")[1])
    Mp = (m + 255) // 256 * 256
    A = rb(Mp, D); W = rb(4 * D, D, D ** -0.5); bias = torch.randn(4 * D, device="cuda")
    h8 = torch.zeros(Mp, 4 * D, dtype=torch.uint8, device="cuda"); g = torch.zeros(Mp, 4 * D, dtype=torch.bfloat16, device="cuda")
    A2 = rb(Mp, 4 * D); W2 = rb(4 * D, D, D ** -0.5)       # fc2 dgrad: dh[M, 3072] = dy[M, 768] @ W2^T[768 -> 3072] x gelu'(h)
    dh = torch.zeros(Mp, 4 * D, dtype=torch.bfloat16, device="cuda")
    hip.gemm_nt(A, W, m, 4 * D, D, hip.EPI_GELU_GRAD | hip.EPI_U8, h8, out2=g, bias=bias)
    for _ in range(REP):
        if case.startswith("fc1_gelu"):
            hip.gemm_nt(A, W, m, 4 * D, D, hip.EPI_GELU_GRAD | hip.EPI_U8, h8, out2=g, bias=bias)
        elif case.startswith("fc1_plain"):
            hip.gemm_nt(A, W, m, 4 * D, D, hip.EPI_BF16, g, bias=bias)
        else:
            hip.gemm_nt(A, W2, m, 4 * D, D, hip.EPI_MUL_AUX | hip.EPI_U8, dh, aux=h8)
elif case.startswith("tn_"):
    Mp = (M + 255) // 256 * 256
    def wprob(n1, n2):
        return (rb(Mp, n1, 0.5), rb(Mp, n2, 0.5), M, n1, n2, torch.zeros(n1, n2, device="cuda"), torch.zeros(n1, device="cuda"), False)
    blk = [wprob(D, 4 * D), wprob(4 * D, D), wprob(3 * D, D), wprob(3 * D, D), wprob(D, D), wprob(D, D)]
    if case == "tn_block":
        grp = hip.TnGroup(blk, layers=[[0, 1, 2, 3], [4, 5]])
    else:
        grp = hip.TnGroup(blk[:4], splits=int(case[-1]))
    for _ in range(REP):
        grp.run()
torch.cuda.synchronize()
