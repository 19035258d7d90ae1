"""Dev: where the wall time of a step goes, from a rocprofv3 kernel trace (scripts/dev/trace_step.sh).
Per hardware queue: busy time, idle gaps between consecutive kernels, the largest gap sources; overlap between queues."""
import csv, gzip, sys, collections, re
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace/kernel_trace.csv.gz"
rows = list(csv.DictReader(gzip.open(path, "rt")))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
def short(n):
    n = n.replace("void ", "").replace("oat::(anonymous namespace)::", "").replace("oat::", ""); n = re.sub(r"\(.*", "", n)
    return n[:60]
# the last step: from the last adamw block backwards to the previous one
adam = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]
_k = int(__import__('os').environ.get('STEP_BACK', '1'))
lo, hi = ends[-1 - _k] + 1, ends[-_k] + 1
step = rows[lo:hi]
t0, t1 = step[0]["s"], max(r["e"] for r in step)
print(f"last step: {len(step)} kernels, wall {(t1 - t0) / 1e6:.2f} ms")
byq = collections.defaultdict(list)
for r in step:
    byq[(r["Queue_Id"], r["Stream_Id"])].append(r)
for q, ks in sorted(byq.items(), key=lambda kv: -sum(r["e"] - r["s"] for r in kv[1])):
    busy = sum(r["e"] - r["s"] for r in ks)
    gaps = [(ks[i]["s"] - ks[i - 1]["e"], short(ks[i - 1]["Kernel_Name"]), short(ks[i]["Kernel_Name"])) for i in range(1, len(ks))]
    small = sum(g for g, _, _ in gaps if 0 < g < 20000)
    big = [(g, a, b) for g, a, b in gaps if g >= 20000]
    print(f"queue {q}: {len(ks)} kernels, busy {busy / 1e6:.2f} ms, gaps < 20 us: {small / 1e6:.2f} ms in {sum(1 for g, _, _ in gaps if 0 < g < 20000)}"
          f" (median {sorted(g for g, _, _ in gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us), gaps >= 20 us: {sum(g for g, _, _ in big) / 1e6:.2f} ms in {len(big)}")
    for g, a, b in sorted(big, reverse=True)[:12]:
        print(f"   BIG {g / 1e3:8.1f} us  {a}  ->  {b}")
    agg = collections.Counter()
    for g, a, b in gaps:
        if 0 < g < 20000: agg[(a, b)] += g
    for (a, b), g in agg.most_common(8):
        print(f"      {g / 1e3:8.1f} us  {a}  ->  {b}")
# union busy time over all queues (any kernel running)
ev = sorted([(r["s"], 1) for r in step] + [(r["e"], -1) for r in step])
run, last, any_busy, multi = 0, t0, 0, 0
for t, d in ev:
    if run > 0: any_busy += t - last
    if run > 1: multi += t - last
    run += d; last = t
print(f"any kernel running {any_busy / 1e6:.2f} ms, nothing running {(t1 - t0 - any_busy) / 1e6:.2f} ms, >= 2 kernels running {multi / 1e6:.2f} ms")
# per kernel name on the main queue: mean duration
mainq = max(byq.items(), key=lambda kv: sum(r["e"] - r["s"] for r in kv[1]))[1]
agg = collections.defaultdict(lambda: [0, 0])
for r in mainq:
    a = agg[short(r["Kernel_Name"])]; a[0] += 1; a[1] += r["e"] - r["s"]
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"   {t / 1e6:7.2f} ms {c:5d} x {t / c / 1e3:7.1f} us  {n}")
