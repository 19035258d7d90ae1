"""Dev microbench for the GEMM kernels on the hot-path shapes (random bf16 operands)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M = int(os.environ.get("M", 50208))
shapes = [(M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 3072), (M, 768, 2304)]
variants = [int(v) for v in os.environ.get("VARIANTS", "1,2").split(",")]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e-3
Mp = (M + 255) // 256 * 256
for (m, n, k) in shapes:
    A = torch.randn(Mp, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda"); out16 = torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16)
    out32 = torch.zeros(Mp, n, device="cuda"); res = torch.randn(Mp, n, device="cuda")
    ref = None
    for v in variants:
        hip.gemm_set_variant(v)
        t = timeit(lambda: hip.gemm_nt(A, B, m, n, k, hip.EPI_BF16, out16, bias=bias))
        if ref is None: ref = out16.clone()
        ok = torch.equal(ref, out16)
        t2 = timeit(lambda: hip.gemm_nt(A, B, m, n, k, hip.EPI_F32, out32, bias=bias, resid=res))
        print(f"NT v{v} M={m} N={n} K={k}: bf16-out {2*m*n*k/t/1e12:7.1f} TF/s ({t*1e6:7.1f} us)  f32+resid {2*m*n*k/t2/1e12:7.1f} TF/s  same_as_first={ok}")
    hip.gemm_set_variant(0)
    # wgrad: out[n,k] = dY[m,n]^T X[m,k]
    P = torch.randn(Mp, n, device="cuda").bfloat16(); out = torch.zeros(n, k, device="cuda")
    bo = torch.zeros(n, device="cuda")
    for tv in (1, 2):
        hip.gemm_tn_set_variant(tv)
        t = timeit(lambda: hip.gemm_tn(P, A, m, n, k, out, bias_out=bo))
        print(f"TN v{tv}   M={m} N1={n} N2={k}: {2*m*n*k/t/1e12:7.1f} TF/s ({t*1e6:7.1f} us)")
    hip.gemm_tn_set_variant(0)
