import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M = 50208; Mp = 50432
which = os.environ.get("WHICH", "tn")
for (n, k) in [(2304, 768), (768, 3072)]:
    P = torch.randn(Mp, n, device="cuda").bfloat16(); A = torch.randn(Mp, k, device="cuda").bfloat16()
    out = torch.zeros(n, k, device="cuda")
    W = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16(); o16 = torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16)
    for v in (1, 2):
        if which == "tn":
            hip.gemm_tn_set_variant(v)
            for _ in range(3): hip.gemm_tn(P, A, M, n, k, out)
        else:
            hip.gemm_set_variant(v)
            for _ in range(3): hip.gemm_nt(A, W, M, n, k, hip.EPI_BF16, o16)
torch.cuda.synchronize()
