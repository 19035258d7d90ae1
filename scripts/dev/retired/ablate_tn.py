"""Dev: gemm_tn ablations (dbg 1 = no global loads after the first ring fill, 2 = no slab store); timing only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M = 50208; Mp = 50432
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e-3
for (n1, n2) in [(2304, 768), (768, 3072), (768, 768)]:
    P = torch.randn(Mp, n1, device="cuda").bfloat16(); Q = torch.randn(Mp, n2, device="cuda").bfloat16()
    out = torch.zeros(n1, n2, device="cuda"); bo = torch.zeros(n1, device="cuda")
    for rep in range(2):
        for dbg in (0, 1, 2, 3):
            hip.gemm_tn_set_variant(dbg << 8)
            t = timeit(lambda: hip.gemm_tn(P, Q, M, n1, n2, out, bias_out=bo))
            t2 = timeit(lambda: hip.gemm_tn(P, Q, M, n1, n2, out))
            print(f"TN N1={n1} N2={n2} dbg={dbg}: {2*M*n1*n2/t/1e12:7.1f} TF/s ({t*1e6:7.1f} us) | no bias {2*M*n1*n2/t2/1e12:7.1f} TF/s ({t2*1e6:7.1f} us)")
hip.gemm_tn_set_variant(0)
