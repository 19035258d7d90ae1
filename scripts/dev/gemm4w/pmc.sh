#!/bin/bash
# Runs ON THE GPU BOX: matrix-pipe busy share and effective clock of the 4-wave probe kernel (counters in their own passes).
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/gemm4w_pmc; rm -rf "$OUT"; mkdir -p "$OUT"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/k1" -o run -- scripts/dev/gemm4w/${BIN:-gemm4w} 1 > "$OUT/k1.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$OUT/k2" -o run -- scripts/dev/gemm4w/${BIN:-gemm4w} 1 > "$OUT/k2.log" 2>&1
python - "$OUT" <<'PY'
import glob, sys, pandas as pd
out = sys.argv[1]
res = {}
for p in ("k1", "k2"):
    fs = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    tr = glob.glob(f"{out}/{p}/**/*kernel_trace.csv", recursive=True)
    if not fs or not tr: print(p, "missing", open(f"{out}/{p}.log").read()[-400:]); continue
    d = pd.read_csv(fs[0]); kt = pd.read_csv(tr[0])
    kt["us"] = (kt.End_Timestamp - kt.Start_Timestamp) / 1e3
    c = d.groupby(["Dispatch_Id", "Counter_Name"]).Counter_Value.sum().unstack()
    c = c.join(kt.set_index("Dispatch_Id")[["us"]])
    for did, row in c.iterrows():
        res.setdefault(did, {}).update(row.to_dict())
rows = [r for r in res.values()]
for i, r in enumerate(rows):
    g = r.get("GRBM_GUI_ACTIVE"); m = r.get("SQ_VALU_MFMA_BUSY_CYCLES")
    mf = m / (1024 * g / 8) if g and m == m else float("nan"); print(i, f"us {r.get('us', 0):8.1f} MFMA util {mf:.3f} clock {g / 8 / (r.get('us', 1) * 1e3):.2f} GHz wait_inst {r.get('SQ_WAIT_INST_ANY', 0) / max(1, r.get('SQ_WAVE_CYCLES', 1)):.2f}")
PY
