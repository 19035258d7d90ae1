#!/bin/bash
# Dev, on the GPU box: interleaved A/B of two builds of the library on the headline step.
#   A = oa-transformer_amd/liboatrans_base.so (built from the previous commit), B = the in-tree library.   bash scripts/dev/ab.sh [rounds] [bench args...]
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; export TMPDIR=/tmp
R=${1:-3}; shift || true
OUT=$PWD/gpurun_out/ab; mkdir -p "$OUT"; : > "$OUT/res.txt"
run() { env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-traffic --no-forced-w1 $ARGS 2>/dev/null | grep '^{' | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
ARGS="$*"
for i in $(seq "$R"); do
  a=$(run OAT_LIB=$PWD/oa-transformer_amd/liboatrans_base.so); b=$(run A=1)
  echo "base $a  new $b" | tee -a "$OUT/res.txt"
done
