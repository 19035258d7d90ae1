#!/bin/bash
# Scaling curve in one command, on a node with N >= 1 MI355X:   bash scripts/scale.sh [max_gpus] [steps] [warmup]
# Runs bench.py at 1, 2, 4, 8 ranks (up to max_gpus, default = GPUs visible), one process per GPU over RCCL/xGMI (plain
# `python bench.py --gpus N`: it spawns its ranks itself; under torch.distributed.run it uses the launcher's env), and prints per N: RCCL rank count, whole-job pairs/s, ms/step and the per-rank step time spread.
# OAT_GRAD_DTYPE=bf16 halves the bytes of the gradient exchange (parallel.GradSync docstring).
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0           # dmabuf IPC: without it RCCL fails with hipIpcGetMemHandle: invalid argument
VISIBLE=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
MAX=${1:-$VISIBLE}; STEPS=${2:-20}; WARMUP=${3:-5}
OUT=${OUT:-gpurun_out/scale}; mkdir -p "$OUT"
echo "# GPUs visible: $VISIBLE ; running N in {1,2,4,8} up to $MAX ; steps=$STEPS warmup=$WARMUP grad_dtype=${OAT_GRAD_DTYPE:-fp32}"
printf "%-3s %-6s %-12s %-10s %-24s\n" N ranks pairs/s ms/step "per-rank ms (min..max)"
BASE=""
for N in 1 2 4 8; do
  [ "$N" -gt "$MAX" ] && break
  LOG="$OUT/n$N.log"
  # plain form at every N: bench.py starts its own ranks when no launcher set RANK / WORLD_SIZE (bench.py:_self_launch)
  timeout 900 python bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARMUP" --no-cpu-baseline --no-other-configs --no-traffic --no-forced-w1 > "$LOG" 2> "$OUT/n$N.err"
  python - "$LOG" "$N" "${BASE:-0}" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line:
    print(f"{sys.argv[2]:<3} FAILED (see {sys.argv[1].replace('.log', '.err')})"); sys.exit(0)
d = json.loads(line[-1]); n = int(sys.argv[2]); base = float(sys.argv[3])
rt = d.get("rank_ms_per_step") or [d["ms_per_step"]]
eff = f"  eff {d['value'] / (n * base):.3f}" if base else ""
print(f"{n:<3} {d.get('ranks_in_group', d['n_gpus']):<6} {d['value']:<12} {d['ms_per_step']:<10} {min(rt):.2f}..{max(rt):.2f}{eff}")
PY
  [ "$N" -eq 1 ] && BASE=$(python -c "import json,sys; print([json.loads(l)['value'] for l in open('$LOG') if l.startswith('{')][-1])" 2>/dev/null)
done
