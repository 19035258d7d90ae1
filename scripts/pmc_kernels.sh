#!/bin/bash
# Runs ON THE GPU BOX: MFMA-pipe / wait / LDS counters of every hot kernel alone (scripts/pmc_kernels.py), counters in their own
# passes with --kernel-trace only.   bash scripts/pmc_kernels.sh <tag>  ->  gpurun_out/<tag>/pmc_mfma_lds.md
set -u
TAG=${1:-pmc}
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/k1" -o run -- python scripts/pmc_kernels.py > "$OUT/k1.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$OUT/k2" -o run -- python scripts/pmc_kernels.py > "$OUT/k2.log" 2>&1
python - "$OUT" "$TAG" <<'PY'
import glob, sys, pandas as pd
out, tag = sys.argv[1], sys.argv[2]
cnt, dur = {}, {}
for p in ("k1", "k2"):
    fs = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(p, "no counters:", open(f"{out}/{p}.log").read()[-600:]); continue
    d = pd.read_csv(fs[0])
    d["key"] = d.Kernel_Name.str.replace("void ", "").str.replace("oat::", "").str.replace("(anonymous namespace)::", "").str.slice(0, 60) + " grid " + d.Grid_Size.astype(str)
    t = d.groupby(["key", "Dispatch_Id", "Counter_Name"]).Counter_Value.sum().unstack().groupby("key").mean()
    for k, row in t.iterrows():
        cnt.setdefault(k, {}).update(row.to_dict())
    tr = glob.glob(f"{out}/{p}/**/*kernel_trace.csv", recursive=True)
    if tr:
        kt = pd.read_csv(tr[0])
        kt["key"] = kt.Kernel_Name.str.replace("void ", "").str.replace("oat::", "").str.replace("(anonymous namespace)::", "").str.slice(0, 60) + " grid " + (kt.Grid_Size_X if "Grid_Size_X" in kt else kt.Grid_Size).astype(str)
        for k, v in ((kt.End_Timestamp - kt.Start_Timestamp).groupby(kt.key).mean() / 1e3).items():
            dur[k] = v
with open(f"{out}/pmc_mfma_lds.md", "w") as fh:
    fh.write(f"# rocprofv3 --kernel-trace --pmc (two passes: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY | SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE) — `python scripts/pmc_kernels.py` ({tag})\n\n"
             "Each hot kernel alone on the GPU at the bench shapes (B=32, 8 frames, M=50208), 3 launches each; durations under the counter pass.\n"
             "`MFMA util` = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): share of the launch's cycles the matrix pipes were busy; `wait` columns are fractions of SQ_WAVE_CYCLES; "
             "`LDS conflict` = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS; `clock` = GRBM_GUI_ACTIVE / 8 XCDs / duration.\n\n"
             "| kernel | µs | clock GHz | MFMA util | wait any | wait inst | LDS conflict |\n|---|---|---|---|---|---|---|\n")
    for k in sorted(cnt):
        c = cnt[k]; g = lambda n: float(c.get(n, float("nan")))
        us = dur.get(k, float("nan"))
        if k.startswith("at::"):
            continue
        mf = g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * g("GRBM_GUI_ACTIVE") / 8) if g("GRBM_GUI_ACTIVE") else float("nan")
        fh.write(f"| `{k}` | {us:.1f} | {g('GRBM_GUI_ACTIVE') / 8 / (us * 1e3):.2f} | {mf:.3f} | {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.2f} | {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.2f} | "
                 f"{(g('SQ_LDS_BANK_CONFLICT') / g('SQ_ACTIVE_INST_LDS')) if g('SQ_ACTIVE_INST_LDS') else 0:.2f} |\n")
print(open(f"{out}/pmc_mfma_lds.md").read())
PY
rm -rf "$OUT/k1" "$OUT/k2"
