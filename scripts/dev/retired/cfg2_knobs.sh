#!/bin/bash
# Dev, on the GPU box: config 2 (4 frames, bs 32: M = 25 120 rows = 1.16 rounds of 256-row tiles at N = 768) under the tile / split-K knobs
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; export TMPDIR=/tmp
run() { env "$@" python bench.py --frames 4 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-traffic --no-forced-w1 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; }
for i in 1 2; do
echo "default $(run A=1)"
echo "splitk $(run OAT_SPLITK=1)"
echo "m224=0 $(run OAT_GEMM_M224=0)"
echo "m224=2 $(run OAT_GEMM_M224=2)"
echo "tail_split $(run OAT_TAIL_SPLIT=1)"
echo "band0 $(run OAT_GEMM_BAND=0)"
done
