# Runs ON THE GPU BOX: L2-miss read traffic of the grouped weight-gradient launch as a function of the plan (splits over M).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/tn_traffic; rm -rf $OUT; mkdir -p $OUT
ROUNDS=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o run -- python scripts/dev/tn_group_time.py > $OUT/log.txt 2>&1
python - <<PY
import glob, pandas as pd
f = glob.glob("$OUT/f/**/*counter_collection.csv", recursive=True)[0]
d = pd.read_csv(f)
d = d[d.Counter_Name == "FETCH_SIZE"]
g = d.groupby(["Dispatch_Id", "Kernel_Name", "Grid_Size"], as_index=False).Counter_Value.sum()
g = g[g.Kernel_Name.str.contains("gemm_tn_sk|gemm_tn_pp")]
g["MB"] = 2 * g.Counter_Value * 1024 / 1e6
print(g.groupby(["Kernel_Name", "Grid_Size"]).MB.agg(["mean", "count"]).to_string())
PY
tail -12 $OUT/log.txt
rm -rf $OUT/f
