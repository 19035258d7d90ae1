"""Dev: main-loop ablations of gemm_nt through the dbg bits (1 = no epilogue, 16 = 2-stage A instead of the 3-stage ring).  Results are wrong by construction; only the timing matters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M = 50208
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e-3
Mp = (M + 255) // 256 * 256
for (m, n, k) in [(M, 768, 3072), (M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 2304)]:
    A = torch.randn(Mp, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda"); out16 = torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16)
    for rep in range(2):
      for VAR in (2,):
        for dbg in [int(x) for x in os.environ.get("DBGS", "128,0").split(",")]:
            hip.gemm_set_variant(VAR | (dbg << 8))
            t = timeit(lambda: hip.gemm_nt(A, B, m, n, k, hip.EPI_BF16, out16, bias=bias))
            print(f"N={n} K={k} variant={VAR & 0xff} persist={VAR >> 16} dbg={dbg}: {2*m*n*k/t/1e12:7.1f} TF/s ({t*1e6:7.1f} us)")
hip.gemm_set_variant(0)
