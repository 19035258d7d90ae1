#!/bin/bash
# Runs ON THE GPU BOX: the headline step under single-knob changes, baseline interleaved (box clocks drift).
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/knob_sweep.log
: > $OUT
run() {  # name, env...
  local name=$1; shift
  local line=$(env "$@" timeout 120 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>/dev/null | tail -1)
  echo "$name $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>/dev/null)" >> $OUT
}
for rep in 1 2; do
run base X=1
run m224_0 OAT_GEMM_M224=0
run m224_2 OAT_GEMM_M224=2
run band0 OAT_GEMM_BAND=0
run band2 OAT_GEMM_BAND=2
run band4 OAT_GEMM_BAND=4
run lnf4096 OAT_LN_FWD_BLOCKS=4096
run lnf16384 OAT_LN_FWD_BLOCKS=16384
run lnbx4096 OAT_LN_BWDX_BLOCKS=4096
run lnbx16384 OAT_LN_BWDX_BLOCKS=16384
run base X=1
run prune OAT_PRUNE_TOP=1
run tailsplit OAT_TAIL_SPLIT=1
run wg0 OAT_GROUP_WGRADS=0
run lane0 OAT_CLS_LANE=0
run textstream0 OAT_TEXT_STREAM=0
done
cat $OUT
