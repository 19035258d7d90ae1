#!/bin/bash
# Dev, runs on the GPU box: in-step durations of selected kernels (KERNELS=a,b,..) under rocprofv3 --stats for two library builds:
# oa-transformer_amd/_ab/lib_base.so (OAT_LIB) against the product library.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; export TMPDIR=/tmp
for l in base new; do
  OUT=$PWD/gpurun_out/fix_$l; rm -rf "$OUT"; mkdir -p "$OUT"
  if [ $l = base ]; then export OAT_LIB=$PWD/oa-transformer_amd/_ab/lib_base.so; else unset OAT_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > "$OUT/log" 2>&1
  python - "$OUT" $l <<'PY'
import glob, sys, pandas as pd
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
d = pd.read_csv(f[0])
import os
pats = os.environ.get("KERNELS", "tn_sk_fix,ln_fold_grads,gemm_tn_sk_kernel").split(",")
for k in pats:
    r = d[d.Name.str.contains(k, regex=False)]
    print(sys.argv[2], k, r.Calls.sum(), round(r.TotalDurationNs.sum() / max(1, r.Calls.sum()) / 1e3, 1), "us", round(r.TotalDurationNs.sum() / 1e6 / 8, 2), "ms per step")
PY
done
