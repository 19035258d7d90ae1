#!/bin/bash
# Dev, on the GPU box: the measured statements VERDICT round 4 asks for (item 2): per case the kernel's duration (kernel trace) and its
# L2-miss read / write bytes (FETCH_SIZE, WRITE_SIZE: separate passes, gfx950 correction) -> gpurun_out/r5m/summary.md
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5m; rm -rf "$OUT"; mkdir -p "$OUT"
CASES="space_bwd_v0 space_bwd_v2 fc1_gelu_M50208 fc1_gelu_M16384 fc1_plain_M50208 fc1_plain_M16384 fc2_dgrad_M50208 fc2_dgrad_M16384 tn_block tn_A_splits1 tn_A_splits2 tn_A_splits3 tn_A_splits4"
for c in $CASES; do
  CASE=$c timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$c/t" -o run -- python scripts/dev/r5_cases.py > "$OUT/$c.log" 2>&1
  CASE=$c timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/$c/f" -o run -- python scripts/dev/r5_cases.py >> "$OUT/$c.log" 2>&1
  CASE=$c timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/$c/w" -o run -- python scripts/dev/r5_cases.py >> "$OUT/$c.log" 2>&1
done
python - "$OUT" $CASES <<'PY'
import glob, sys, pandas as pd
out, cases = sys.argv[1], sys.argv[2:]
pick = {"space": "attn_space_bwd", "fc1": "gemm_nt_pp", "fc2": "gemm_nt_pp", "tn": "gemm_tn_sk_kernel"}
rows = []
for c in cases:
    pat = pick[c.split("_")[0]]
    def one(kind):
        f = glob.glob(f"{out}/{c}/{kind}/**/*" + ("kernel_trace.csv" if kind == "t" else "counter_collection.csv"), recursive=True)
        return pd.read_csv(f[0]) if f else None
    t, f, w = one("t"), one("f"), one("w")
    if t is None or f is None or w is None:
        rows.append((c, "missing", 0, 0, 0, 0)); continue
    t = t[t.Kernel_Name.str.contains(pat)]
    t = t.iloc[1:] if len(t) > 1 else t                              # drop the first (cold) launch
    us = ((t.End_Timestamp - t.Start_Timestamp).mean()) / 1e3
    name = t.Kernel_Name.iloc[0].replace("void ", "").replace("oat::(anonymous namespace)::", "").replace("oat::", "")[:56]
    def cnt(df, n):
        df = df[(df.Counter_Name == n) & df.Kernel_Name.str.contains(pat)]
        d = df.groupby("Dispatch_Id").Counter_Value.sum()
        return float(d.iloc[1:].mean() if len(d) > 1 else d.mean()) * 1024
    rd, wr = 2 * cnt(f, "FETCH_SIZE"), cnt(w, "WRITE_SIZE")
    rows.append((c, name, us, rd / 1e6, wr / 1e6, len(t)))
with open(f"{out}/summary.md", "w") as fh:
    fh.write("| case | kernel | us per launch (kernel trace) | read MB (2 x FETCH_SIZE) | write MB | launches |\n|---|---|---|---|---|---|\n")
    for r in rows:
        fh.write(f"| {r[0]} | `{r[1]}` | {r[2]:.1f} | {r[3]:.1f} | {r[4]:.1f} | {r[5]} |\n")
print(open(f"{out}/summary.md").read())
PY
for c in $CASES; do rm -rf "$OUT/$c"; done
