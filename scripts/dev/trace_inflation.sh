#!/bin/bash
# Dev, runs on the GPU box: kernel traces of short bench runs (frozen and global_local) -> per-launch durations of the main
# queue's kernels (trace_inflation.py) and the queue / gap summary (trace_gaps.py) under gpurun_out/trace_<variant>/.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
for v in ${VARIANTS:-frozen global_local}; do
  OUT=$PWD/gpurun_out/trace_$v
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o run -- \
      python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs --variant $v > "$OUT/bench.log" 2>&1
  find "$OUT" -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'gzip -c {} > '"$OUT"'/kernel_trace.csv.gz; rm {}'
  python scripts/dev/trace_inflation.py $OUT/kernel_trace.csv.gz > $OUT/inflation.txt 2>&1
  python scripts/dev/trace_gaps.py $OUT/kernel_trace.csv.gz > $OUT/gaps.txt 2>&1
  python scripts/dev/trace_timeline.py $OUT/kernel_trace.csv.gz > $OUT/timeline.txt 2>&1
  rm -f $OUT/kernel_trace.csv.gz
  find "$OUT" -type f -size +2M -delete
done
