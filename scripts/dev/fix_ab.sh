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
for k in ("tn_sk_fix", "ln_fold_grads", "gemm_tn_sk_kernel"):
    r = d[d.Name.str.contains(k)]
    print(sys.argv[2], k, r.Calls.sum(), round(r.TotalDurationNs.sum() / max(1, r.Calls.sum()) / 1e3, 1), "us")
PY
done
