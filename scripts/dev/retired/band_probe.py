"""Dev: band-grouped tile walk of the ping-pong gemm_nt (oat_gemm_set_band) vs the row-major walk on the short-K, wide-N
launches of a step: bit-identical outputs, interleaved timing (median of ROUNDS)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip

M = int(os.environ.get("M", 50208)); Mp = (M + 255) // 256 * 256
ROUNDS = int(os.environ.get("ROUNDS", 5))
lib = hip.lib()


def timeit(fn, n=10):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3


torch.manual_seed(0)
ok = True
cases = [("bf16", hip.EPI_BF16, 2304, 768), ("bf16", hip.EPI_BF16, 3072, 768), ("gelu_grad_u8", hip.EPI_GELU_GRAD | hip.EPI_U8, 3072, 768),
         ("mul_aux_u8", hip.EPI_MUL_AUX | hip.EPI_U8, 3072, 768), ("bf16", hip.EPI_BF16, 768, 3072), ("bf16", hip.EPI_BF16, 768, 2304)]
for name, epi, n, k in cases:
    A = torch.randn(Mp, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda")
    u8 = bool(epi & hip.EPI_U8)
    base = epi & 0xff
    mk = lambda: (torch.full((Mp, n), 7, device="cuda", dtype=torch.uint8 if (u8 and base == hip.EPI_GELU_GRAD) else torch.bfloat16),
                  torch.full((Mp, n), 7.0, device="cuda", dtype=torch.bfloat16))
    aux = (torch.randint(0, 255, (Mp, n), device="cuda", dtype=torch.uint8) if u8 else torch.randn(Mp, n, device="cuda").bfloat16())

    def run(o, o2):
        hip.gemm_nt(A, B, M, n, k, epi, o, out2=o2 if base == hip.EPI_GELU_GRAD else None, bias=bias,
                    aux=aux if base == hip.EPI_MUL_AUX else None)
    bands = [0, 3, 4] if n // 256 == 12 else ([0, 3] if n // 256 == 9 else [0, 1, 2])
    if n // 256 == 12:
        bands.append(6)
    outs, ts = {}, {b: [] for b in bands}
    for b in bands:
        lib.oat_gemm_set_band(b)
        o, o2 = mk()
        run(o, o2)
        outs[b] = (o, o2)
    for r in range(ROUNDS):
        for b in bands:
            lib.oat_gemm_set_band(b)
            ts[b].append(timeit(lambda: run(*outs[b])))
    same = all(torch.equal(outs[b][0], outs[0][0]) and torch.equal(outs[b][1], outs[0][1]) for b in bands)
    ok &= same
    line = "  ".join(f"band {b}: {sorted(ts[b])[ROUNDS // 2]:6.1f} us" for b in bands)
    print(f"{name:13s} N={n:5d} K={k:5d}: identical={same}  {line}")
lib.oat_gemm_set_band(0)
print("BAND", "PASSED" if ok else "FAILED")
