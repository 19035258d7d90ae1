"""bench.py's contract: one JSON line on rank 0 with the fields the driver reads, at one rank and at two ranks.
The two-rank run is the dry-run form (two processes on the one GPU over gloo, OAT_BENCH_ONE_DEVICE / OAT_BENCH_BACKEND):
it exercises every `world > 1` branch of bench.py - barriers, max-over-ranks time, the instrumented step that EVERY rank
has to run because it contains collectives - so that a hang there is caught before the multi-GPU scaling run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-3000:]                       # exactly one JSON line (rank 0 only)
    return json.loads(lines[0])


def _check(rec, n):
    for k in REQUIRED:
        assert k in rec, k
    assert rec["n_gpus"] == n and rec["unit"] == "pairs/s" and rec["higher_is_better"] is True and rec["scaling"] == "weak"
    assert rec["vs_baseline"] is None and rec["dtype"] == "bf16" and rec["data"] == "synthetic" and "workload" in rec["config"]
    assert rec["value"] > 0 and abs(rec["value"] - n * rec["config"]["per_gpu_batch"] / (rec["ms_per_step"] * 1e-3)) < 0.01 * rec["value"]
    r = rec["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def test_bench_single_rank_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4", "--frames", "2",
                        "--other-configs"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rec = _line(r.stdout)
    _check(rec, 1)
    # config 3 as worded (object-aware variants) rides along outside `value`
    oc = rec["other_configs"]
    # ... and the two classes whose DEFAULT schedule prunes the top block (VideoEngine.prune_top) once more on the full graph
    # ... and the per-GPU shapes of BASELINE.json's configs 2, 4 and 5 (config 5 in both dtypes), scaled with the command line
    assert [o["workload"].split("]")[0] for o in oc] == [
        "[config 3, global_local", "[config 3, region_mem", "[frozen, full graph", "[region_mem, full graph",
        "[config 2", "[config 4, per-GPU shape", "[config 5 geometry, bf16", "[config 5 geometry, bf16, twice the batch",
        "[config 5 geometry, fp8 forward"], oc
    for o in oc:
        assert "error" not in o, o
        assert o["value"] > 0 and o["ms_per_step"] > 0 and 0 < o["step_mfma_frac"] < 1 and o["unit"] == "pairs/s"
        assert o["dtype"] and o["per_gpu_batch"] > 0 and o["frames"] > 0 and "gflop_per_pair" in o
    assert (oc[4]["frames"], oc[5]["per_gpu_batch"], oc[6]["res"], oc[6]["frames"], oc[7]["per_gpu_batch"]) == (1, 8, 336, 4, 2 * oc[6]["per_gpu_batch"])
    assert oc[6]["dtype"] == "bf16" and oc[8]["dtype"].startswith("fp8") and oc[6]["gflop_per_pair"] == oc[8]["gflop_per_pair"]
    # default schedule = pruned top block: the line counts EXECUTED FLOPs and names the full graph's beside them; the full-graph
    # entries of other_configs count everything
    assert rec["config"]["gflop_per_pair"] < rec["config"]["gflop_per_pair_full_graph"] == oc[2]["gflop_per_pair"]
    assert oc[1]["gflop_per_pair"] < oc[1]["gflop_per_pair_full_graph"] == oc[3]["gflop_per_pair"]      # region_mem: pruned by default
    assert "gflop_per_pair_full_graph" not in oc[0]                                                      # global_local reads the patch rows
    # the W > 1 launch path on a 1-rank RCCL group, beside the headline
    w1 = rec["w1_forced"]
    assert "error" not in w1, w1
    assert rec["ms_per_step_w1_forced"] == w1["ms_per_step_w1_forced"] > 0
    assert all(v["async_all_reduces_per_step"] >= 3 for v in w1["variants"].values()), w1     # text tower + ViT blocks + tables
    # roofline.traffic: PMC bytes per launch of the roofline kernel, or a stated reason
    tr = rec["roofline"]["traffic"]
    if tr is None:
        assert rec["roofline"]["traffic_skipped"]
    else:
        assert tr["read_mb"] > 0 and tr["write_mb"] > 0 and tr["algorithmic_mb"] > 0 and tr["launches_counted"] > 0
        assert abs(tr["ratio"] - tr["total_mb"] / tr["algorithmic_mb"]) < 0.01 * tr["ratio"] + 1e-3
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "pairs/s" and cb["sample"]


def test_bench_prints_its_line_when_the_optional_legs_run_out_of_budget():
    """The one JSON line comes last; optional legs (PMC traffic, w1_forced, other_configs) are skipped with a reason once the
    wall-time budget is used up, and the headline fields are all there."""
    env = dict(os.environ, OAT_BENCH_BUDGET_S="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4", "--frames", "2",
                        "--other-configs", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rec = _line(r.stdout)
    _check(rec, 1)
    assert rec["roofline"]["traffic"] is None and "budget" in rec["roofline"]["traffic_skipped"]
    assert "w1_forced" not in rec
    assert rec["other_configs"] and all("budget" in o["skipped"] for o in rec["other_configs"])


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher and no rendezvous variables: bench.py starts the two ranks itself
    (dry-run form: both on the one GPU, over gloo) and rank 0 prints one line for the group."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OAT_BENCH_ONE_DEVICE="1", OAT_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--frames", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rec = _line(r.stdout)
    _check(rec, 2)
    assert rec["ranks_in_group"] == 2 and len(rec["rank_ms_per_step"]) == 2 and rec["config"]["global_batch"] == 8


def test_bench_gpus_beyond_the_visible_devices_fails_loudly():
    n = __import__("torch").cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OAT_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--batch", "2",
                        "--frames", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and "are visible" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_two_rank_dry_run_terminates():
    env = dict(os.environ, OAT_BENCH_ONE_DEVICE="1", OAT_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "4", "--frames", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    _check(_line(r.stdout), 2)
