#!/bin/bash
# Dev, runs on the GPU box: kernel trace with timestamps of a short bench run -> gpurun_out/trace/ (analysed by
# scripts/dev/trace_gaps.py).  Only the kernel-trace domain.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$PWD/gpurun_out/trace
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o run -- \
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs "$@" > "$OUT/bench.log" 2>&1
find "$OUT" -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'gzip -c {} > '"$OUT"'/kernel_trace.csv.gz; rm {}'
ls -la "$OUT"; tail -1 "$OUT/bench.log" | cut -c1-200
