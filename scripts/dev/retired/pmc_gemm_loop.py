"""Dev: gemm_nt alone (N768/K3072 and N2304/K768), with and without the epilogue, for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M = 50208; Mp = (M + 255) // 256 * 256
for (n, k) in [(768, 3072), (2304, 768)]:
    A = torch.randn(Mp, k, device="cuda").bfloat16(); W = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    o = torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16); bias = torch.randn(n, device="cuda")
    for dbg in (0, 1):
        hip.gemm_set_variant(2 | (dbg << 8))
        for _ in range(3): hip.gemm_nt(A, W, M, n, k, hip.EPI_BF16, o, bias=bias)
torch.cuda.synchronize()
