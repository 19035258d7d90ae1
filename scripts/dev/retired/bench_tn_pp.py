"""Dev: ping-pong gemm_tn (variant 4) vs the lockstep 256x256 kernel (variant 2) and an fp64 reference: correctness on
the hot shapes (ragged M, NaN-poisoned pad rows, tiny M, bias sums, accumulate), repeat-run determinism, timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip

M = int(os.environ.get("M", 50208))
SHAPES = [(M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 3072), (M, 768, 2304)]
ROUNDS = int(os.environ.get("ROUNDS", 4))
NOSTAG, NOSTORE = 2, 8


def timeit(fn, n=10):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e-3


def one(m, n1, n2, slots=0, poison=True):
    mp = (m + 255) // 256 * 256
    P = torch.randn(mp, n1, device="cuda").bfloat16()
    Q = torch.randn(mp, n2, device="cuda").bfloat16()
    if poison and mp > m:
        P[m:] = float("nan")
        Q[m:] = float("nan")
    ref = (P[:m].double().t() @ Q[:m].double())
    refb = P[:m].double().sum(0)
    res = {}
    for v in (2, 4):
        hip.gemm_tn_set_variant(v | (slots << 16))
        out = torch.full((n1, n2), 3.0, device="cuda")
        bo = torch.full((n1,), 3.0, device="cuda")
        hip.gemm_tn(P, Q, m, n1, n2, out, bias_out=bo)
        out2 = torch.full((n1, n2), 1.0, device="cuda")
        bo2 = torch.full((n1,), 1.0, device="cuda")
        hip.gemm_tn(P, Q, m, n1, n2, out2, bias_out=bo2, accumulate=True)
        res[v] = (out, bo, out2, bo2)
    scale = ref.abs().max().item() + 1e-9
    e_new = (res[4][0].double() - ref).abs().max().item() / scale
    e_old = (res[2][0].double() - ref).abs().max().item() / scale
    eb = (res[4][1].double() - refb).abs().max().item() / (refb.abs().max().item() + 1e-9)
    eacc = (res[4][2].double() - 1.0 - ref).abs().max().item() / scale
    ebacc = (res[4][3].double() - 1.0 - refb).abs().max().item() / (refb.abs().max().item() + 1e-9)
    ok = e_new < 2e-5 and eb < 2e-5 and eacc < 2e-5 and ebacc < 2e-5 and bool(torch.isfinite(res[4][0]).all())
    print(f"check M={m} N1={n1} N2={n2} slots={slots}: rel err new {e_new:.2e} old {e_old:.2e} bias {eb:.2e} acc {eacc:.2e}/{ebacc:.2e} -> {'ok' if ok else 'BAD'}")
    return ok


def check():
    torch.manual_seed(0)
    ok = True
    for (m, n1, n2) in SHAPES:
        ok &= one(m, n1, n2)
        ok &= one(m, n1, n2, slots=192)
    for m in (1, 31, 64, 65, 100, 128, 129, 192, 200, 256, 1000, 4096, 4097 + 64):
        ok &= one(m, 256, 512)
    ok &= one(8000, 768, 256)
    # determinism
    m, n1, n2 = SHAPES[0]
    mp = (m + 255) // 256 * 256
    P = torch.randn(mp, n1, device="cuda").bfloat16(); Q = torch.randn(mp, n2, device="cuda").bfloat16()
    hip.gemm_tn_set_variant(4 | (192 << 16))
    ref = torch.empty(n1, n2, device="cuda"); rb = torch.empty(n1, device="cuda")
    hip.gemm_tn(P, Q, m, n1, n2, ref, bias_out=rb)
    side = torch.cuda.Stream(); junk = torch.randn(64 << 20, device="cuda")
    bad = 0
    for it in range(30):
        with torch.cuda.stream(side):
            junk.mul_(1.0001)
        o = torch.empty(n1, n2, device="cuda"); b = torch.empty(n1, device="cuda")
        hip.gemm_tn(P, Q, m, n1, n2, o, bias_out=b)
        bad += int(not (torch.equal(o, ref) and torch.equal(b, rb)))
    torch.cuda.synchronize()
    print(f"race screen: {bad} of 30 launches differ")
    ok &= bad == 0
    hip.gemm_tn_set_variant(0)
    print("CHECK", "PASSED" if ok else "FAILED")
    return ok


def bench():
    res = {}
    mp = (M + 255) // 256 * 256
    variants = [("lockstep", 2), ("pp", 4), ("pp-nostagger", 4 | (NOSTAG << 8)), ("pp-nostore", 4 | (NOSTORE << 8)),
                ("lockstep-192", 2 | (192 << 16)), ("pp-192", 4 | (192 << 16))]
    for (m, n1, n2) in SHAPES:
        P = torch.randn(mp, n1, device="cuda").bfloat16(); Q = torch.randn(mp, n2, device="cuda").bfloat16()
        out = torch.zeros(n1, n2, device="cuda"); bo = torch.zeros(n1, device="cuda")
        for name, v in variants:
            hip.gemm_tn_set_variant(v)
            hip.gemm_tn(P, Q, m, n1, n2, out, bias_out=bo)
        for r in range(ROUNDS):
            for name, v in variants:
                hip.gemm_tn_set_variant(v)
                t = timeit(lambda: hip.gemm_tn(P, Q, m, n1, n2, out, bias_out=bo))
                res.setdefault((n1, n2, name), []).append(t)
        for name, v in variants:
            ts = sorted(res[(n1, n2, name)]); med = ts[len(ts) // 2]
            print(f"N1={n1:5d} N2={n2:5d} {name:14s}: median {2*m*n1*n2/med/1e12:7.1f} TF/s ({med*1e6:7.1f} us incl. reduce)  best {2*m*n1*n2/ts[0]/1e12:7.1f}")
    hip.gemm_tn_set_variant(0)


if __name__ == "__main__":
    ok = check()
    if ok or os.environ.get("FORCE_BENCH"):
        bench()
