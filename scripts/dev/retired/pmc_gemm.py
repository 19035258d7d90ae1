"""Dev: a few launches per (shape, variant) of gemm_nt EPI_BF16, in a fixed order, for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M = int(os.environ.get("M", 50208))
shapes = [(M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 3072), (M, 768, 2304)]
variants = [int(v) for v in os.environ.get("VARIANTS", "2,514").split(",")]
Mp = (M + 255) // 256 * 256
for (m, n, k) in shapes:
    A = torch.randn(Mp, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda"); out16 = torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16)
    for v in variants:
        hip.gemm_set_variant(v)
        for _ in range(3):
            hip.gemm_nt(A, B, m, n, k, hip.EPI_BF16, out16, bias=bias)
    torch.cuda.synchronize()
