"""Dev: grouped weight gradients (hip.TnGroup, csrc/gemm_tn_sk.hip) vs the per-problem gemm_tn + tn_reduce launches at the
shapes of one ViT block (M = 50208) and of one DistilBERT backward (36 problems, M = 1024): interleaved timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
hip.lib()
ROUNDS = int(os.environ.get("ROUNDS", 5))


def timeit(fn, n=5):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3


def problems(M, shapes):
    Mp = (M + 255) // 256 * 256
    out = []
    for n1, n2 in shapes:
        P = (torch.randn(Mp, n1, device="cuda") * 0.5).bfloat16(); Q = (torch.randn(Mp, n2, device="cuda") * 0.5).bfloat16()
        out.append((P, Q, M, n1, n2, torch.zeros(n1, n2, device="cuda"), torch.zeros(n1, device="cuda"), False))
    return out


D = 768
M = int(os.environ.get("M", 50208))
blk = problems(M, [(D, 4 * D), (4 * D, D), (3 * D, D), (3 * D, D), (D, D), (D, D)])
single = lambda ps: [hip.gemm_tn(P, Q, m, n1, n2, o, bias_out=b) for P, Q, m, n1, n2, o, b, _ in ps]
gA, gB = hip.TnGroup(blk[:4]), hip.TnGroup(blk[4:])
g6s = hip.TnGroup(blk, splits=0)            # all six, stream mode (no operand locality): for the record
variants = {"6 x (gemm_tn + reduce)": lambda: single(blk), "2 groups (4 x 2-way + 2 x 14-way)": lambda: (gA.run(), gB.run()),
            "1 group, stream mode": g6s.run}
for sp in (1, 2, 3, 4):
    g = hip.TnGroup(blk[:4], splits=sp)
    variants[f"group A alone, {sp}-way ({g.grid} wgs)"] = g.run
variants["group B alone"] = gB.run
for v in variants.values():
    v()
ts = {k: [] for k in variants}
for r in range(ROUNDS):
    for k, v in variants.items():
        ts[k].append(timeit(v))
gf = sum(2 * M * n1 * n2 for _, _, _, n1, n2, _, _, _ in blk) / 1e9
for k in variants:
    t = sorted(ts[k])[ROUNDS // 2]
    print(f"block wgrads M={M}: {k:42s} {t:8.1f} us" + (f"  ({gf / t / 1e3:6.1f} TF/s)" if "alone" not in k else ""))
# DistilBERT
txt = problems(1024, [(D, 4 * D), (4 * D, D), (D, D), (D, D), (D, D), (D, D)] * 6)
gt = hip.TnGroup(txt, splits=0)
variants = {"36 x (gemm_tn + reduce)": lambda: single(txt), "1 group (stream, whole tiles)": gt.run}
for v in variants.values():
    v()
ts = {k: [] for k in variants}
for r in range(ROUNDS):
    for k, v in variants.items():
        ts[k].append(timeit(v))
for k in variants:
    print(f"DistilBERT wgrads M=1024: {k:34s} {sorted(ts[k])[ROUNDS // 2]:8.1f} us")
