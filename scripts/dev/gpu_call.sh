cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s4g; rm -rf $O; mkdir -p $O
for r in 1 2 3; do
for v in 0 0x100; do
OAT_SPACE_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_$v_$r.log 2>&1
python - $O/bench_$v_$r.log $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("variant",sys.argv[2],d["value"],d["ms_per_step"])
PY
done; done
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -2
