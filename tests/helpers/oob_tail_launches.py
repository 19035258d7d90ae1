"""Test helper (run as a subprocess by tests/test_kernels_gpu.py): the B-row launches of the pruned top block and of the per-clip tails (engine/video.py: _top_tail_fwd, _top_block_bwd_pruned, the
final LayerNorm) on operands that END at the last byte of their hipMalloc (PYTORCH_NO_CUDA_MEMORY_CACHING=1, allocations of whole 2 MiB pages): any row
touched past M - 1 is an illegal access.  Prints "ok <op>" per op."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd"))
import torch
from OATrans.ops import hip
M = int(sys.argv[1])
D, Hd = 768, 3072
def tail(rows, cols, dtype, fill=None):
    """the last `rows` rows of an allocation of whole 2 MiB pages"""
    esz = torch.empty(0, dtype=dtype).element_size()
    per = (2 << 20) // (cols * esz) if (2 << 20) % (cols * esz) == 0 else None
    R = 4096
    while (R * cols * esz) % (2 << 20) or R < rows: R += 4096
    big = torch.zeros(R, cols, dtype=dtype, device="cuda")
    t = big[R - rows:]
    if fill is not None: t.copy_(fill)
    return t
bf, f32 = torch.bfloat16, torch.float32
rb = lambda r, c: torch.randn(r, c, device="cuda").to(bf)
W1 = (torch.randn(Hd, D, device="cuda") * D ** -0.5).to(bf); W2 = (torch.randn(D, Hd, device="cuda") * Hd ** -0.5).to(bf)
Wp = (torch.randn(D, D, device="cuda") * D ** -0.5).to(bf)
b1 = torch.randn(Hd, device="cuda"); b2 = torch.randn(D, device="cuda")
# forward tail: proj, LN, fc1 + GELU (bf16 derivative), fc2
o_s = tail(M, D, bf, rb(M, D)); brs = tail(M, D, bf); x = tail(M, D, bf, rb(M, D)); a2 = tail(M, D, bf)
mean, rstd = tail(M, 1, f32).view(-1), tail(M, 1, f32).view(-1)
h, g, br = tail(M, Hd, bf), tail(M, Hd, bf), tail(M, D, bf)
hip.gemm_nt(o_s, Wp, M, D, D, hip.EPI_BF16, brs, bias=b2); torch.cuda.synchronize(); print("ok gemm_nt proj")
hip.layernorm_fwd_r16(x, M, D, 1e-6, add_a=brs, y=a2, mean=mean, rstd=rstd); torch.cuda.synchronize(); print("ok layernorm_fwd_r16")
hip.gemm_nt(a2, W1, M, Hd, D, hip.EPI_GELU_GRAD, h, out2=g, bias=b1); torch.cuda.synchronize(); print("ok gemm_nt fc1 gelu")
hip.gemm_nt(g, W2, M, D, Hd, hip.EPI_BF16, br, bias=b2); torch.cuda.synchronize(); print("ok gemm_nt fc2")
# backward tail
ga = tail(M, D, bf, rb(M, D)); d_h = tail(M, Hd, bf); d_a = tail(M, D, bf); gb = tail(M, D, bf)
hip.gemm_nt(ga, W2.t().contiguous(), M, Hd, D, hip.EPI_MUL_AUX, d_h, aux=h); torch.cuda.synchronize(); print("ok gemm_nt fc2 dgrad x aux")
hip.gemm_nt(d_h, W1.t().contiguous(), M, D, Hd, hip.EPI_BF16, d_a); torch.cuda.synchronize(); print("ok gemm_nt fc1 dgrad")
hip.layernorm_bwd_xhat(d_a, a2, rstd, M, D, dx16=gb, add_a=ga); torch.cuda.synchronize(); print("ok layernorm_bwd_xhat")
dW = torch.zeros(D, Hd, device="cuda"); db = torch.zeros(D, device="cuda")
hip.gemm_tn(ga, g, M, D, Hd, dW, bias_out=db); torch.cuda.synchronize(); print("ok gemm_tn fc2 wgrad")
# final LayerNorm on the CLS rows (fp32 output)
y32 = tail(M, D, f32); s16 = tail(M, D, bf)
gam, bet = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
hip.layernorm_fwd_r16(x, M, D, 1e-6, add_a=brs, add_b=br, sum16=s16, gamma=gam, beta=bet, y32=y32, mean=mean, rstd=rstd)
torch.cuda.synchronize(); print("ok final layernorm")
print("done")
