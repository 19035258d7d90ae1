"""Dev: fp8 ping-pong gemm_nt (oat_gemm_nt_f8) - correctness against an fp32 matmul of the SAME quantised operands
(exact up to fp32 summation order) and against the unquantised product (fp8 error), then timing next to the bf16 kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip

M = int(os.environ.get("M", 50208))
Mp = (M + 255) // 256 * 256


def timeit(fn, n=10):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); st.record()
    for _ in range(n):
        fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e-3


def quantise(x, margin=1.0):
    """-> (uint8 e4m3 tensor, dq scalar tensor) through the HIP kernels (amax -> scales -> quant)."""
    R, C = x.shape
    st = torch.zeros(3, device="cuda")
    hip.fp8_amax(x, R, C, st[0:1])
    hip.fp8_update_scales(st[0:1], st[1:2], st[2:3], 1, margin)
    q = torch.empty(R, C, dtype=torch.uint8, device="cuda")
    hip.fp8_quant(x, q, R, C, st[1:2])
    return q, st[2:3], st[1:2]


def check():
    torch.manual_seed(0)
    ok = True
    for (m, n, k) in [(512, 256, 256), (1000, 768, 768), (4096 + 17, 512, 512), (M, 2304, 768), (M, 768, 3072)]:
        mp = (m + 255) // 256 * 256
        A = torch.randn(mp, k, device="cuda").bfloat16()
        B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
        bias = torch.randn(n, device="cuda")
        A8, dqa, qa = quantise(A)
        B8, dqb, qb = quantise(B)
        # the conversion itself against torch's e4m3fn
        want = (A.float() * qa).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        conv_ok = torch.equal(A8, want)
        Aq = A8.view(torch.float8_e4m3fn).float() * dqa
        Bq = B8.view(torch.float8_e4m3fn).float() * dqb
        ref_q = Aq[:m] @ Bq.t() + bias
        ref = A[:m].float() @ B.float().t() + bias
        out = torch.full((mp, n), 7.0, device="cuda", dtype=torch.bfloat16)
        hip.gemm_nt_f8(A8, B8, m, n, k, hip.EPI_BF16, out, dqa, dqb, bias=bias)
        e_exact = (out[:m].float() - ref_q).abs().max().item()
        e_fp8 = ((out[:m].float() - ref).norm() / ref.norm()).item()
        untouched = bool((out[m:] == 7.0).all())
        scale = ref_q.abs().max().item()
        good = conv_ok and untouched and e_exact < 2e-2 * max(1.0, scale / 4)          # bf16 output rounding
        print(f"M={m} N={n} K={k}: conv==torch {conv_ok}  max|out - fp32(quantised)| {e_exact:.4f} (|ref|max {scale:.1f})  "
              f"rel-L2 vs unquantised {e_fp8:.4f}  pad rows untouched {untouched} -> {'ok' if good else 'BAD'}")
        ok &= good
        if k == 768 and n % 256 == 0:                     # GELU_GRAD epilogue (fc1)
            o1 = torch.empty(mp, n, device="cuda", dtype=torch.bfloat16); o2 = torch.empty_like(o1)
            hip.gemm_nt_f8(A8, B8, m, n, k, hip.EPI_GELU_GRAD, o1, dqa, dqb, out2=o2, bias=bias)
            gl = torch.nn.functional.gelu(ref_q)
            eg = (o2[:m].float() - gl).abs().max().item()
            print(f"   gelu epilogue max err {eg:.4f}")
            ok &= eg < 5e-2 * max(1.0, scale / 4)
    print("CHECK", "PASSED" if ok else "FAILED")
    return ok


def bench():
    for (m, n, k) in [(M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 3072)]:
        A = torch.randn(Mp, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
        bias = torch.randn(n, device="cuda"); out = torch.zeros(Mp, n, device="cuda", dtype=torch.bfloat16)
        A8, dqa, qa = quantise(A); B8, dqb, _ = quantise(B)
        t8 = timeit(lambda: hip.gemm_nt_f8(A8, B8, m, n, k, hip.EPI_BF16, out, dqa, dqb, bias=bias))
        t16 = timeit(lambda: hip.gemm_nt(A, B, m, n, k, hip.EPI_BF16, out, bias=bias))
        am = torch.zeros(1, device='cuda')
        tq = timeit(lambda: hip.fp8_quant(A, A8, Mp, k, qa, am))
        print(f"N={n:5d} K={k:5d}: fp8 {2*m*n*k/t8/1e12:7.1f} TF/s ({t8*1e6:6.1f} us)   bf16 {2*m*n*k/t16/1e12:7.1f} TF/s ({t16*1e6:6.1f} us)   "
              f"quantise A {tq*1e6:6.1f} us ({Mp*k*3/tq/1e12:.2f} TB/s)")


if __name__ == "__main__":
    if check() or os.environ.get("FORCE_BENCH"):
        bench()
