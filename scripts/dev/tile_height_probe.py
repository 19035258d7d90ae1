"""Dev: per-tile time of the ping-pong gemm_nt at tile heights 256 / 224 / 192 (EPI_BF16, N = 768): M is chosen so that every
height runs EXACTLY `R` full rounds on 256 CUs (M = R * 256 / 3 row tiles), i.e. time / R = the time of one tile of that
height including its share of the epilogue.  Decides whether slabs of mixed heights (2 x 192 + 224 per workgroup instead of
3 x 224 at M = 50208) can pay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
lib = hip.lib()
n = 768
NS = 3
def timeit(fn, reps=30):
    for i in range(6): fn(i % NS)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for i in range(reps): fn(i % NS)
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / reps * 1e3
for k in (768, 2304, 3072):
    B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    for mode, tm in ((0, 256), (2, 224), (3, 192)):
        lib.oat_gemm_set_m224(mode)
        R = 3
        M = tm * (R * 256 // 3)
        Mp = (M + 255) // 256 * 256
        A = [torch.randn(Mp, k, device="cuda").bfloat16() for _ in range(NS)]
        o = [torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
        ts = [timeit(lambda i: hip.gemm_nt(A[i], B, M, n, k, hip.EPI_BF16, o[i])) for _ in range(3)]
        t = min(ts)
        print(f"K {k:5d} tile {tm}: M {M:6d} {t:7.1f} us = {t / R:6.2f} us per tile, {t / R / (tm // 32):5.2f} per 32 rows, {2 * M * n * k / t / 1e6:7.0f} TF/s", flush=True)
        del A, o
lib.oat_gemm_set_m224(1)
