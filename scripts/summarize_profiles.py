"""Turns the rocprofv3 CSVs written by scripts/make_profiles.sh into the small summaries kept under profiles/:
   <tag>_bench_kernel_stats.{csv,md}, <tag>_pmc_hbm_traffic.md, <tag>_pmc_traffic.json, <tag>_final_bench_line.json.
   Usage: python scripts/summarize_profiles.py gpurun_out/<tag> <tag>   (writes into gpurun_out/<tag>/profiles/)"""
import glob
import json
import os
import re
import shutil
import sys

import pandas as pd

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(src, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return hits[-1]


def bench_ms(log):
    for line in open(os.path.join(src, log)):
        if line.startswith("{"):
            return json.loads(line)["ms_per_step"]
    return float("nan")


# ---- kernel statistics
stats = pd.read_csv(one("stats/**/*kernel_stats.csv"))
shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
# executed steps = launches of a once-per-step kernel (warm-up + timed + the instrumented step; with the hipGraph step also
# its two eager warm-ups and first replay)
once = stats[stats.Name.str.contains("cast_bf16_multi")]
steps = int(once.Calls.iloc[0]) // 2 if len(once) else 5 + 2 + 1      # one cast launch per tower per step
total = stats.TotalDurationNs.sum() / 1e6
final = json.loads([l for l in open(os.path.join(src, "final_bench_line.json")) if l.startswith("{")][-1])
with open(os.path.join(dst, f"{tag}_final_bench_line.json"), "w") as fh:
    fh.write(json.dumps(final) + "\n")
with open(os.path.join(dst, f"{tag}_bench_kernel_stats.md"), "w") as fh:
    fh.write(f"# rocprofv3 --kernel-trace --stats — `python bench.py --steps 5 --warmup 2 --no-cpu-baseline` ({tag}, MI355X)\n\n")
    fh.write(f"{steps - 1} training steps + 1 instrumented step (B=32, 8 frames); {bench_ms('stats_bench.log'):.1f} ms/step under the profiler, "
             f"{final['ms_per_step']:.1f} ms without (`{tag}_final_bench_line.json`).  Kernels of the HIP streams overlap, so durations sum to "
             f"more than the wall time ({total:.1f} ms of kernel time over {steps} steps = {total / steps:.1f} ms per step).\n"
             f"Raw rocprofv3 table: `{tag}_bench_kernel_stats.csv`.\n\n| kernel | calls | total ms | avg µs | % |\n|---|---|---|---|---|\n")
    for _, r in stats.sort_values("TotalDurationNs", ascending=False).head(32).iterrows():
        fh.write(f"| `{r.Name[:96]}` | {r.Calls} | {r.TotalDurationNs / 1e6:.2f} | {r.AverageNs / 1e3:.1f} | {r.Percentage:.1f} |\n")

# ---- PMC traffic
def counters(kind, name):
    df = pd.read_csv(one(f"{kind}/**/*counter_collection.csv"))
    df = df[df.Counter_Name == name]
    per_dispatch = df.groupby(["Dispatch_Id", "Kernel_Name"], as_index=False).Counter_Value.sum()
    return per_dispatch.groupby("Kernel_Name").Counter_Value.agg(["mean", "count"])

rd, wr = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
rows = []
for k in rd.index:
    r = 2 * rd.loc[k, "mean"] * 1024                  # gfx950: FETCH_SIZE tallies 128-B requests at 64 B; unit KiB
    w = wr.loc[k, "mean"] * 1024 if k in wr.index else 0.0
    rows.append((k, int(rd.loc[k, "count"]), r, w))
rows.sort(key=lambda t: -(t[2] + t[3]) * t[1])
with open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.md"), "w") as fh:
    fh.write(f"# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) — `python bench.py --steps 1 --warmup 1 --no-cpu-baseline` ({tag})\n\n"
             "Per-launch averages over every step of the run (B=32, 8 frames).  Correction per `MI355X_MICROARCH.md` (HBM section): counters are in KiB;\n"
             "on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so reads = 2 x FETCH_SIZE x 1024; writes = WRITE_SIZE x 1024.\n"
             "The memory-side counters include Infinity-Cache hits, so `read MB` is L2-miss traffic, an upper bound on HBM reads.\n\n"
             "| kernel | launches | read MB | write MB | total MB per launch |\n|---|---|---|---|---|\n")
    for k, n, r, w in rows[:24]:
        fh.write(f"| `{k[:84]}` | {n} | {r / 1e6:.1f} | {w / 1e6:.1f} | {(r + w) / 1e6:.1f} |\n")
out = []
for k, n, r, w in rows:
    if "gemm_nt_pp_kernel<0" in k.replace(" ", "") or "gemm_tn_pp_kernel" in k or "gemm_tn_sk_kernel" in k:
        out.append({"kernel": "gemm_nt_pp_kernel<EPI_BF16>" if "gemm_nt" in k else ("gemm_tn_sk_kernel" if "_sk_" in k else "gemm_tn_pp_kernel"),
                    "rocprof_name": k, "read_bytes_per_launch": r,
                   "write_bytes_per_launch": w, "bytes_per_launch": r + w, "launches": n,
                   "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 1; "
                             "reads = 2*FETCH_SIZE KiB (gfx950 correction), writes = WRITE_SIZE KiB",
                   "workload": "frozen B=32 T=8"})
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))
