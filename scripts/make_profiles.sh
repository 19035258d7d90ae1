#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the three rocprofv3 passes the committed profiles/ summaries come from.
#   bash scripts/make_profiles.sh <tag>      e.g. round1e
# Counters are collected in their own passes with --kernel-trace only (never combined with other trace domains).
set -u
TAG=${1:-roundX}
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$PWD/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- \
    python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > "$OUT/stats_bench.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o run -- \
    python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > "$OUT/fetch_bench.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o run -- \
    python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > "$OUT/write_bench.log" 2>&1
timeout 300 python bench.py > "$OUT/final_bench_line.json" 2> "$OUT/final_bench.err"
python scripts/summarize_profiles.py "$OUT" "$TAG" > "$OUT/summary.log" 2>&1
tail -3 "$OUT/summary.log"; tail -1 "$OUT/final_bench_line.json" | cut -c1-400
