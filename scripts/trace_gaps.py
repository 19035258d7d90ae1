"""Dev: summarise a rocprofv3 kernel-trace CSV of bench.py: per-stream busy time and the idle gaps of the main stream."""
import sys
import pandas as pd
d = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp")
ad = d[d.Kernel_Name.str.contains("adamw")].Start_Timestamp.values
bounds = [ad[0]]
for a, b in zip(ad, ad[1:]):
    if b - a > 20e6: bounds.append(b)
lo, hi = bounds[3], bounds[4]
print("step ms", (hi - lo) / 1e6)
w = d[(d.Start_Timestamp >= lo) & (d.Start_Timestamp < hi)]
for sid, g in w.groupby("Stream_Id"):
    print(sid, "first %.2f last %.2f busy %.2f ms n=%d" % ((g.Start_Timestamp.min() - lo) / 1e6, (g.End_Timestamp.max() - lo) / 1e6, ((g.End_Timestamp - g.Start_Timestamp).sum()) / 1e6, len(g)))
main = w[w.Stream_Id == w.Stream_Id.value_counts().index[0]] if False else w[w.Stream_Id == w[w.Kernel_Name.str.contains("adamw")].Stream_Id.iloc[0]]
prev_end, prev_name = lo, "step start"
tot = 0
for _, r in main.iterrows():
    gap = r.Start_Timestamp - prev_end
    if gap > 30e3:
        tot += gap
        print(f"gap {gap/1e3:8.1f} us at {(prev_end-lo)/1e6:7.2f} ms  after {prev_name[:45]:45s} before {r.Kernel_Name[:45]}")
    prev_end, prev_name = max(prev_end, r.End_Timestamp), r.Kernel_Name
print("total main-stream gaps > 30us: %.2f ms" % (tot / 1e6))
