"""Dev: the MLP pair of a ViT block alone on the GPU at the bench shape (M = 50208, N = 3072, K = 768): plain bf16 epilogue,
fc1 forward with the GELU epilogue + 8-bit derivative, fc2 data gradient x saved derivative.  Three rotating buffer sets
(nothing is L2 / Infinity-Cache warm from the previous launch), HIP-event timing over 30 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M = int(os.environ.get("M", 50208)); Mp = (M + 255) // 256 * 256
n, k = 3072, 768
NS = 3
A = [torch.randn(Mp, k, device="cuda").bfloat16() for _ in range(NS)]
B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
bias = torch.randn(n, device="cuda")
o1 = [torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
o2 = [torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
d8 = [torch.zeros(Mp, n, device="cuda", dtype=torch.uint8) for _ in range(NS)]
def timeit(fn, reps=30):
    for i in range(6): fn(i % NS)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for i in range(reps): fn(i % NS)
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / reps * 1e-3
for rep in range(3):
    t0 = timeit(lambda i: hip.gemm_nt(A[i], B, M, n, k, hip.EPI_BF16, o1[i], bias=bias))
    t1 = timeit(lambda i: hip.gemm_nt(A[i], B, M, n, k, hip.EPI_GELU_GRAD | hip.EPI_U8, d8[i], out2=o2[i], bias=bias))
    t2 = timeit(lambda i: hip.gemm_nt(A[i], B, M, n, k, hip.EPI_MUL_AUX | hip.EPI_U8, o1[i], aux=d8[i]))
    f = 2 * M * n * k / 1e12
    print(f"plain {t0*1e6:6.1f} us ({f/t0:6.0f} TF/s)   gelu+u8 {t1*1e6:6.1f} us ({f/t1:6.0f})   mul_aux u8 {t2*1e6:6.1f} us ({f/t2:6.0f})", flush=True)
print("checksum", o1[0].float().abs().sum().item(), d8[0].float().sum().item())
