cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2a; rm -rf $OUT; mkdir -p $OUT
timeout 300 python scripts/step_breakdown.py > $OUT/breakdown.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats_bench.log 2>&1
T=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
python scripts/trace_gaps.py $T > $OUT/gaps.log 2>&1
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $S $OUT/kernel_stats.csv
# keep the trace of one step only (size)
python - <<PY
import pandas as pd
d = pd.read_csv("$T").sort_values("Start_Timestamp")
ad = d[d.Kernel_Name.str.contains("adamw")].Start_Timestamp.values
b=[ad[0]]
for x,y in zip(ad,ad[1:]):
    if y-x>20e6: b.append(y)
lo,hi=b[3],b[4]
w=d[(d.Start_Timestamp>=lo)&(d.Start_Timestamp<hi)][["Kernel_Name","Stream_Id","Start_Timestamp","End_Timestamp","Workgroup_Size_X","Grid_Size_X"]].copy()
w["Start_Timestamp"]-=lo; w["End_Timestamp"]-=lo
w["Kernel_Name"]=w.Kernel_Name.str.slice(0,70)
w.to_csv("$OUT/one_step_trace.csv",index=False)
PY
rm -rf $OUT/stats
tail -6 $OUT/breakdown.log; head -12 $OUT/gaps.log
