"""dev: the fc1 forward launch (EPI_GELU_GRAD | EPI_U8, M = 50208, N = 3072, K = 768) and its plain-bf16 twin, alone, under the
library named by OAT_LIB - the in-tree build or one of the ablation builds (compile-time -DOAT_ABL bits in a COPY of
csrc/gemm_nt_pp.hip: 1 = no derivative-block store, 2 = no GELU arithmetic, 4 = no gelu(h) store; never part of the product
build).  Prints microseconds per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M, N, K = 50208, 3072, 768
Mp = (M + 255) // 256 * 256
torch.manual_seed(0)
A = torch.randn(Mp, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
bias = torch.randn(N, device="cuda")
g = torch.empty(Mp, N, device="cuda", dtype=torch.bfloat16)
d8 = torch.empty(Mp, N, device="cuda", dtype=torch.uint8)
o = torch.empty(Mp, N, device="cuda", dtype=torch.bfloat16)
def timeit(fn, n=30):
    for _ in range(5): fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3
res = []
for rnd in range(3):
    t1 = timeit(lambda: hip.gemm_nt(A, W, M, N, K, hip.EPI_GELU_GRAD | hip.EPI_U8, d8, out2=g, bias=bias))
    t0 = timeit(lambda: hip.gemm_nt(A, W, M, N, K, hip.EPI_BF16, o, bias=bias))
    res.append((t1, t0))
print(os.path.basename(os.environ.get("OAT_LIB", "in-tree")), "fc1+GELU us:", [round(a, 1) for a, _ in res], "plain bf16 us:", [round(b, 1) for _, b in res], flush=True)
