"""Dev: the last step of a kernel trace as a per-queue timeline: every kernel with start offset (us), duration, queue."""
import csv, gzip, sys, collections, re
path = sys.argv[1]
rows = list(csv.DictReader(gzip.open(path, "rt")))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
def short(n):
    n = n.replace("void ", "").replace("oat::(anonymous namespace)::", "").replace("oat::", ""); n = re.sub(r"\(.*", "", n) if len(n) < 60 or "at::" not in n else n
    return n[:int(__import__("os").environ.get("NAMELEN", "44"))]
adam = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]
_k = int(__import__('os').environ.get('STEP_BACK', '1'))      # 1 = the last step (bench.py: the instrumented one), 2 = the step before it
step = rows[ends[-1 - _k] + 1:ends[-_k] + 1]
t0 = step[0]["s"]
qs = {}
for r in step:
    q = (r["Queue_Id"], r["Stream_Id"])
    qs.setdefault(q, len(qs))
print("queues:", {v: k for k, v in qs.items()})
for q, i in qs.items():
    ks = [r for r in step if (r["Queue_Id"], r["Stream_Id"]) == q]
    print(f"queue {i}: {len(ks)} kernels, first start {(ks[0]['s']-t0)/1e3:.0f} us, last end {(max(r['e'] for r in ks)-t0)/1e3:.0f} us, busy {sum(r['e']-r['s'] for r in ks)/1e3:.0f} us")
for r in step:
    print(f"{(r['s']-t0)/1e3:9.1f} {(r['e']-r['s'])/1e3:8.1f} q{qs[(r['Queue_Id'], r['Stream_Id'])]} {short(r['Kernel_Name'])}")
