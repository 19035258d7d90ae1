#!/bin/bash
# Runs ON THE GPU BOX: one bench line per BASELINE.json configuration other than the headline, for profiles/<tag>_other_config_bench_lines.jsonl
#   bash scripts/other_configs.sh <tag>
set -u
TAG=${1:-roundX}
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
F="$OUT/${TAG}_other_config_bench_lines.jsonl"
: > "$F"
run() { timeout 600 python bench.py --no-cpu-baseline --no-other-configs "$@" 2>> "$OUT/other_configs.err" | tail -1 >> "$F"; }
run --frames 4                                                        # config 2
run --variant global_local --frames 8                                 # config 3 as worded (8-frame + 10 obj)
run --variant region_mem --frames 8
run --batch 64                                                        # config 4's per-GPU shape
run --res 336 --frames 16 --variant global_local --batch 8            # config 5's geometry, bf16
run --res 336 --frames 16 --variant global_local --batch 8 --dtype fp8
run --dtype fp8                                                       # headline shape, fp8 forward
python - "$F" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(f"{d['value']:8.2f} pairs/s {d['ms_per_step']:8.2f} ms  {d['dtype']}  {d['config']['workload'][:90]}")
PY
