#!/bin/bash
# Dev, on the GPU box: the W > 1 launch path on a 1-rank RCCL group with 4 (default) and 8 hardware queues; plain run beside it.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/w1q; mkdir -p "$OUT"; : > "$OUT/res.txt"
one() { # label, env..., -- args
  local label=$1; shift
  ms=$(env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-traffic --no-forced-w1 $EXTRA 2>/dev/null | grep '^{' | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "$label $ms" | tee -a "$OUT/res.txt"
}
for rep in 1 2; do
  EXTRA="" one "plain" A=1
  EXTRA="--force-w1-main" one "w1 grid=auto q=default" OAT_BWD_NT_GRID=auto
  EXTRA="--force-w1-main" one "w1 grid=auto q=8" OAT_BWD_NT_GRID=auto GPU_MAX_HW_QUEUES=8
  EXTRA="--force-w1-main" one "w1 grid=tile q=default" A=1
  EXTRA="--force-w1-main" one "w1 grid=tile q=8" GPU_MAX_HW_QUEUES=8
  EXTRA="" one "plain q=8" GPU_MAX_HW_QUEUES=8
done
