cd "${GRAFT_REPO_ROOT:-$(pwd)}"; export TMPDIR=/tmp
run() { env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-traffic --no-forced-w1 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['host_loop_ms_per_step'], d['launch_mode'][:30])"; }
for i in 1 2; do
echo "default: $(run A=1)"
echo "graph: $(run OAT_GRAPH=1)"
echo "side: $(run OAT_BWD_SIDE=1)"
echo "side+graph: $(run OAT_BWD_SIDE=1 OAT_GRAPH=1)"
echo "side+graph cus160: $(run OAT_BWD_SIDE=1 OAT_GRAPH=1 OAT_WGRAD_CUS=160)"
echo "side+graph cus224: $(run OAT_BWD_SIDE=1 OAT_GRAPH=1 OAT_WGRAD_CUS=224)"
done
