#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$PWD/gpurun_out/gaptrace
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o run -- python scripts/dev/gap_probe.py > "$OUT/probe.log" 2>&1
find "$OUT" -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'gzip -c {} > '"$OUT"'/kernel_trace.csv.gz; rm {}'
tail -14 "$OUT/probe.log"
