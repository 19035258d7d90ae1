#!/bin/bash
# Dev, runs on the GPU box: board power / clocks as rocm-smi reports them while the training step loops (the cap: --showmaxpower).
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$PWD/gpurun_out/power; mkdir -p "$OUT"
{
echo "== idle"; rocm-smi --showpower --showmaxpower --showperflevel 2>&1 | grep -i "power\|level"
python bench.py --steps ${STEPS:-1500} --warmup 3 --no-cpu-baseline --no-other-configs "$@" > "$OUT/bench.log" 2>&1 &
BP=$!
sleep 22
for i in $(seq 1 14); do echo "== t=$((22 + 2 * i)) s"; rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "package power\|sclk\|junction\|hotspot"; sleep 1.6; done
wait $BP
tail -1 "$OUT/bench.log" | cut -c1-200
} > "$OUT/power.log" 2>&1
