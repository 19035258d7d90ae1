"""Dev: what an [M, N] bf16 read in the ping-pong GEMM's epilogue costs (EPI_MUL_AUX with a bf16 aux tensor against EPI_BF16), at the N = 768
shapes whose output feeds a residual add + LayerNorm - the price a fused residual epilogue would pay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M = 50208; Mp = (M + 255) // 256 * 256
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (n, k) in [(768, 768), (768, 3072), (768, 2304)]:
    A = torch.randn(Mp, k, device="cuda").bfloat16(); W = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    o = torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16); bias = torch.randn(n, device="cuda")
    aux = torch.randn(Mp, n, device="cuda").bfloat16()
    hip.gemm_set_variant(4)
    for rep in range(3):
        t0 = timeit(lambda: hip.gemm_nt(A, W, M, n, k, hip.EPI_BF16, o, bias=bias))
        t1 = timeit(lambda: hip.gemm_nt(A, W, M, n, k, hip.EPI_MUL_AUX, o, bias=bias, aux=aux))
        print(f"N{n} K{k}: EPI_BF16 {t0:6.1f} us   EPI_MUL_AUX (bf16 aux read in the epilogue) {t1:6.1f} us   delta {t1 - t0:+5.1f}")
hip.gemm_set_variant(0)
