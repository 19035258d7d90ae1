"""Dev: the two MLP GEMMs with fused activation epilogues, alone on the GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M = 50208; Mp = (M + 255) // 256 * 256
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e-3
n, k = 3072, 768
A = torch.randn(Mp, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
bias = torch.randn(n, device="cuda"); h = torch.randn(Mp, n, device="cuda").bfloat16()
o1 = torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16); o2 = torch.zeros_like(o1)
for rep in range(2):
    t = timeit(lambda: hip.gemm_nt(A, B, M, n, k, hip.EPI_BF16, o1, bias=bias))
    print(f"N=3072 K=768 EPI_BF16     : {2*M*n*k/t/1e12:7.1f} TF/s ({t*1e6:7.1f} us)")
    t = timeit(lambda: hip.gemm_nt(A, B, M, n, k, hip.EPI_GELU_DUAL, o1, out2=o2, bias=bias))
    print(f"N=3072 K=768 EPI_GELU_DUAL: {2*M*n*k/t/1e12:7.1f} TF/s ({t*1e6:7.1f} us)")
    t = timeit(lambda: hip.gemm_nt(A, B, M, n, k, hip.EPI_DGELU, o1, aux=h))
    print(f"N=3072 K=768 EPI_DGELU    : {2*M*n*k/t/1e12:7.1f} TF/s ({t*1e6:7.1f} us)")
    t = timeit(lambda: hip.gemm_nt(A, B, M, n, k, hip.EPI_GELU_GRAD, o1, out2=o2, bias=bias))
    print(f"N=3072 K=768 EPI_GELU_GRAD: {2*M*n*k/t/1e12:7.1f} TF/s ({t*1e6:7.1f} us)")
    t = timeit(lambda: hip.gemm_nt(A, B, M, n, k, hip.EPI_MUL_AUX, o1, aux=h))
    print(f"N=3072 K=768 EPI_MUL_AUX  : {2*M*n*k/t/1e12:7.1f} TF/s ({t*1e6:7.1f} us)")
    print("checksum", o1.float().abs().sum().item())
