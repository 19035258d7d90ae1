"""Dev: cache policy of the bf16 epilogue stores of gemm_nt_pp (plain / nt / sc1 / sc0 sc1: one library per policy, OAT_LIB): the
EPI_BF16 launches of a ViT block alone on the GPU, three rotating buffer sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M = 50208; Mp = (M + 255) // 256 * 256
NS = 3
def timeit(fn, reps=30):
    for i in range(6): fn(i % NS)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for i in range(reps): fn(i % NS)
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / reps * 1e3
out = []
for n, k in ((768, 768), (2304, 768), (3072, 768), (768, 2304), (768, 3072)):
    A = [torch.randn(Mp, k, device="cuda").bfloat16() for _ in range(NS)]
    B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda")
    o = [torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
    t = min(timeit(lambda i: hip.gemm_nt(A[i], B, M, n, k, hip.EPI_BF16, o[i], bias=bias)) for _ in range(3))
    out.append(f"N{n}/K{k} {t:6.1f}")
    del A, o
print(os.path.basename(os.environ.get("OAT_LIB", "product")), " ".join(out), "us", flush=True)
