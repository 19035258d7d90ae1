#!/bin/bash
# Dev, runs on the GPU box: kernel traces of (a) the headline step, (b) the same step on the W > 1 launch path of a 1-rank RCCL
# group, (c) the global_local step; gap / queue analysis of each (scripts/dev/trace_gaps.py) -> gpurun_out/trace3/*.txt
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/trace3; rm -rf "$OUT"; mkdir -p "$OUT"
run() {  # name, bench args...
  local name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$name" -o run -- \
      python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-traffic "$@" > "$OUT/$name.log" 2>&1
  f=$(find "$OUT/$name" -name "*kernel_trace.csv" | head -1)
  gzip -c "$f" > "$OUT/$name.csv.gz"; rm -rf "$OUT/$name"
  STEP_BACK=2 python scripts/dev/trace_gaps.py "$OUT/$name.csv.gz" > "$OUT/$name.txt" 2>&1
  python scripts/dev/trace_tail.py "$OUT/$name.csv.gz" 70 > "$OUT/$name.tail.txt" 2>&1
  tail -1 "$OUT/$name.log" | cut -c1-160
}
run frozen --no-forced-w1
run frozen_w1 --force-w1-main
run gl --variant global_local --no-forced-w1
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-traffic 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d[\"ms_per_step\"], d.get(\"w1_forced\"))"; done > "$OUT/plain.txt"
# traces are small (a few thousand rows): kept
