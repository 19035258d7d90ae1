cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4oa; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python bench.py --variant global_local --frames 8 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/stats_bench.log 2>&1
T=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
python - <<PY
import pandas as pd, numpy as np
d = pd.read_csv("$T").sort_values("Start_Timestamp")
ad = d[d.Kernel_Name.str.contains("adamw")].Start_Timestamp.values
b=[ad[0]]
for x,y in zip(ad,ad[1:]):
    if y-x>20e6: b.append(y)
lo,hi=b[3],b[4]
w=d[(d.Start_Timestamp>=lo)&(d.Start_Timestamp<hi)].copy()
ms=w[w.Kernel_Name.str.contains("adamw")].Stream_Id.iloc[0]
m=w[w.Stream_Id==ms].sort_values("Start_Timestamp")
busy=(m.End_Timestamp-m.Start_Timestamp).sum()/1e6
gaps=(m.Start_Timestamp.values[1:]-m.End_Timestamp.values[:-1])
print("step %.2f ms; main stream: %d kernels, busy %.2f ms, gaps %.2f ms" % ((hi-lo)/1e6, len(m), busy, gaps[gaps>0].sum()/1e6))
m["name"]=m.Kernel_Name.str.replace("void oat::","").str.replace("(anonymous namespace)::","").str.slice(0,44)
m["dur"]=(m.End_Timestamp-m.Start_Timestamp)/1e3
m["gap_before"]=np.concatenate([[0],gaps])/1e3
g=m.groupby("name").agg(n=("dur","size"),total_ms=("dur",lambda x:x.sum()/1e3),avg_us=("dur","mean"),gap_us=("gap_before","mean")).sort_values("total_ms",ascending=False)
print(g.head(30).to_string())
big=m[m.gap_before>50][["name","gap_before"]]
print(big.head(20).to_string())
for sid,gg in w.groupby("Stream_Id"):
    if sid!=ms: print("other stream",sid,"kernels",len(gg),"busy %.2f ms"%((gg.End_Timestamp-gg.Start_Timestamp).sum()/1e6), "from %.1f to %.1f ms"%((gg.Start_Timestamp.min()-lo)/1e6,(gg.End_Timestamp.max()-lo)/1e6))
PY
rm -rf $OUT/stats
