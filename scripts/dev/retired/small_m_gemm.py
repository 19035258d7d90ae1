"""Dev: which gemm_nt configuration serves mid-size M best (the object clip of the OA variants: M = 32 x 197 = 6304)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip


def timeit(fn, n=20):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); st.record()
    for _ in range(n):
        fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e-3


for M in (1024, 2048, 6304, 12608, 25216):
    Mp = (M + 255) // 256 * 256
    for (n, k) in [(2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)]:
        A = torch.randn(Mp, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
        bias = torch.randn(n, device="cuda"); out = torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16)
        res = {}
        for name, v in (("128x128", 1), ("256 lockstep", 2), ("256 ping-pong", 4), ("pp 1wg/tile", 4 | (0xffff << 16))):
            hip.gemm_set_variant(v)
            res[name] = timeit(lambda: hip.gemm_nt(A, B, M, n, k, hip.EPI_BF16, out, bias=bias))
        hip.gemm_set_variant(0)
        best = min(res, key=res.get)
        print(f"M={M:6d} N={n:5d} K={k:5d}: " + "  ".join(f"{nm} {t*1e6:6.1f} us" for nm, t in res.items()) + f"   tiles256={((M+255)//256)*(n//256)}  best: {best}")
