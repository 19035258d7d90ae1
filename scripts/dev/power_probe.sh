#!/bin/bash
# Dev, runs on the GPU box: board power / clocks / caps as rocm-smi reports them, idle and while the training step loops.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$PWD/gpurun_out/power; mkdir -p "$OUT"
{
echo "== idle"; rocm-smi --showpower --showmaxpower --showclocks --showperflevel 2>&1 | grep -v "^=\|^$"
python bench.py --steps 400 --warmup 3 --no-cpu-baseline --no-other-configs > "$OUT/bench.log" 2>&1 &
BP=$!
sleep 45
for i in 1 2 3 4 5 6; do echo "== step loop, sample $i"; rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk\|fclk"; sleep 1.5; done
wait $BP
tail -1 "$OUT/bench.log" | cut -c1-160
} > "$OUT/power.log" 2>&1
cat "$OUT/power.log"
