#!/bin/bash
# Dev, on the GPU box: text backward issued first (OAT_TEXT_BWD_FIRST=1, default) against the earlier order, on the plain step and on
# the W > 1 launch path of a 1-rank RCCL group; global_local beside it.  Interleaved, two rounds.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/w1ab; mkdir -p "$OUT"; : > "$OUT/res.txt"
one() { # label, env..., (EXTRA = bench args)
  local label=$1; shift
  ms=$(env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-traffic --no-forced-w1 $EXTRA 2>/dev/null | grep '^{' | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "$label $ms" | tee -a "$OUT/res.txt"
}
for rep in 1 2; do
  for first in 1 0; do
    EXTRA="" one "plain first=$first" OAT_TEXT_BWD_FIRST=$first
    EXTRA="--force-w1-main" one "w1 grid=auto first=$first" OAT_BWD_NT_GRID=auto OAT_TEXT_BWD_FIRST=$first
    EXTRA="--force-w1-main" one "w1 grid=tile first=$first" OAT_TEXT_BWD_FIRST=$first
    EXTRA="--variant global_local" one "gl first=$first" OAT_TEXT_BWD_FIRST=$first
    EXTRA="--variant region_mem" one "rm first=$first" OAT_TEXT_BWD_FIRST=$first
  done
done
