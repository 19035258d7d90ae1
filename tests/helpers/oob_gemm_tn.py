"""Test helper (run as a subprocess by tests/test_kernels_gpu.py): oat_gemm_tn on operands that END at the last byte of their hipMalloc (run with PYTORCH_NO_CUDA_MEMORY_CACHING=1: every
tensor its own allocation, sized to a multiple of 2 MiB) - any row fetched past M - 1 is an illegal access.  tests/test_kernels_gpu.py runs it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd"))
import torch
from OATrans.ops import hip
M = int(sys.argv[1]); R = (M + 4095) // 4096 * 4096      # 4096 rows of 768 / 256 bf16 = 6 MiB / 2 MiB: the allocations end on a 2 MiB boundary
N1, N2 = 768, 256
bigP = torch.zeros(R, N1, dtype=torch.bfloat16, device="cuda"); bigQ = torch.zeros(R, N2, dtype=torch.bfloat16, device="cuda")
bigP[R - M:] = torch.randn(M, N1, device="cuda").bfloat16(); bigQ[R - M:] = torch.randn(M, N2, device="cuda").bfloat16()
P, Q = bigP[R - M:], bigQ[R - M:]
out = torch.zeros(N1, N2, device="cuda"); bias = torch.zeros(N1, device="cuda")
for _ in range(20):
    hip.gemm_tn(P, Q, M, N1, N2, out, bias_out=bias)
torch.cuda.synchronize()
ref = P.float().t() @ Q.float()
print("ok", (out - ref).abs().max().item())
