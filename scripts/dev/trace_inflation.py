"""Dev: does side-stream work inflate the main stream's kernels?  From a rocprofv3 kernel trace (scripts/dev/trace_step.sh):
per kernel class on the main queue, the duration of every launch of the last step in order, with the number of OTHER-queue
kernels that were running at its start (static tile assignment: a persistent GEMM workgroup that cannot be placed because a
side-stream workgroup holds its CU starts late and ends late)."""
import csv, gzip, sys, collections, re, bisect
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace/kernel_trace.csv.gz"
rows = list(csv.DictReader(gzip.open(path, "rt")))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
def short(n):
    n = n.replace("void ", "").replace("oat::(anonymous namespace)::", "").replace("oat::", ""); n = re.sub(r"\(.*", "", n)
    return n[:48]
adam = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]
_k = int(__import__('os').environ.get('STEP_BACK', '1'))      # 1 = the last step (bench.py: the instrumented one), 2 = the step before it
step = rows[ends[-1 - _k] + 1:ends[-_k] + 1]
byq = collections.defaultdict(list)
for r in step: byq[(r["Queue_Id"], r["Stream_Id"])].append(r)
mainkey = max(byq, key=lambda k: sum(r["e"] - r["s"] for r in byq[k]))
side = [r for k, v in byq.items() if k != mainkey for r in v]
def overlap(r):      # side-queue kernel time inside r's interval / r's duration
    t = sum(max(0, min(r["e"], s["e"]) - max(r["s"], s["s"])) for s in side)
    return t / max(1, r["e"] - r["s"])
cls = collections.defaultdict(list)
for r in byq[mainkey]:
    cls[(short(r["Kernel_Name"]), r.get("Grid_Size", ""))].append(r)
tot_excess = 0.0
for (n, g), ks in sorted(cls.items(), key=lambda kv: -sum(r["e"] - r["s"] for r in kv[1])):
    if len(ks) < 6 or not ("gemm" in n or "attn" in n or "ln_" in n): continue
    d = [(r["e"] - r["s"]) / 1e3 for r in ks]
    ov = [overlap(r) for r in ks]
    quiet = sorted(x for x, o in zip(d, ov) if o < 0.05) or sorted(d)
    base = quiet[len(quiet) // 2]
    excess = sum(x - base for x in d)
    tot_excess += excess
    print(f"{n:50s} grid {g:>8s} n {len(d):3d} median-quiet {base:7.1f} us  mean {sum(d)/len(d):7.1f}  excess over quiet median {excess/1e3:6.2f} ms")
    print("     " + " ".join(f"{x:.0f}{'*' if o > 0.3 else ''}" for x, o in zip(d, ov)))
print(f"sum of excess: {tot_excess/1e3:.2f} ms per step   (* = side-queue kernels ran during > 30 % of the launch)")
