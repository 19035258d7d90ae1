cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/fp8prof; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python bench.py --dtype fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/bench.log 2>&1
python - <<PY
import pandas as pd
d=pd.read_csv("$OUT/stats/run_kernel_stats.csv")
d["Name"]=d["Name"].str.replace("void oat::","").str.replace("(anonymous namespace)::","").str.slice(0,70)
print(d[["Name","Calls","TotalDurationNs","AverageNs","Percentage"]].head(32).to_string())
PY
rm -rf $OUT/stats
