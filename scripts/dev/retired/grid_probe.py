"""Dev: does a power-bound GEMM keep its throughput on fewer CUs?  Persistent gemm_nt_pp grids of 128 .. 256 workgroups at the bench shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
M = 50208; Mp = (M + 255) // 256 * 256
def timeit(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (n, k) in [(768, 3072), (768, 2304), (2304, 768), (768, 768)]:
    A = torch.randn(Mp, k, device="cuda").bfloat16(); W = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    o = torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16); bias = torch.randn(n, device="cuda")
    for rep in range(2):
        row = []
        for grid in (256, 240, 224, 192, 160, 128):
            hip.gemm_set_variant(grid << 16)
            t = timeit(lambda: hip.gemm_nt(A, W, M, n, k, hip.EPI_BF16, o, bias=bias))
            row.append(f"{grid}: {t:6.1f} us ({2 * M * n * k / t / 1e6:5.0f} TF)")
        print(f"N{n} K{k}  " + "  ".join(row))
hip.gemm_set_variant(0)
