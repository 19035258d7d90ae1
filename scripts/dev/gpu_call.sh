cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s4j; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > $O/pytest_attn.log 2>&1; tail -5 $O/pytest_attn.log
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_abi_cpu.py -x -q > $O/pytest_engine.log 2>&1; tail -3 $O/pytest_engine.log
for r in 1 2 3; do for v in 1 0; do
OAT_FUSED_FINALIZE=$v timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_${v}_$r.log 2>&1
python - $O/bench_${v}_$r.log $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("fused",sys.argv[2],d["value"],d["ms_per_step"])
PY
done; done
