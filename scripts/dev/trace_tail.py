"""Dev: the kernels around the end of backward / gradient sync / AdamW of the second-to-last step of a kernel trace, all queues
interleaved by start time (queue, start offset us, duration us, gap to the previous kernel on the same queue, name)."""
import csv, gzip, re, sys
rows = list(csv.DictReader(gzip.open(sys.argv[1], "rt")))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("oat::(anonymous namespace)::", "").replace("oat::", ""))[:70]
adam = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]
hi = ends[-2] + 1
first_adam = max(i for i in adam if i < hi and (i - 1 not in adam))
lo = max(0, first_adam - int(sys.argv[2]) if len(sys.argv) > 2 else first_adam - 60)
t0 = rows[lo]["s"]; last = {}
for r in rows[lo:hi + 8]:
    q = r["Queue_Id"]
    gap = (r["s"] - last[q]) / 1e3 if q in last else 0.0
    last[q] = r["e"]
    print(f"q{q} +{(r['s'] - t0) / 1e3:9.1f} us  dur {(r['e'] - r['s']) / 1e3:8.1f}  gap {gap:8.1f}  {short(r['Kernel_Name'])}")
