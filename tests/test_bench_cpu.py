"""bench.py's launcher and the small dispatch rules around the step, without a GPU.

`python bench.py --gpus N` (N > 1, no RANK / WORLD_SIZE in the environment) must start N ranks ITSELF with the env rendezvous the
reference's entry points read (train_dist_multi.py:35-38,127-132).  Here, with no GPU, every rank stops at "needs an MI355X":
what is checked is that N ranks were started with the right environment and that the launcher reports the failure."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


@pytest.mark.skipif(torch.cuda.is_available(), reason="the no-GPU failure path")
def test_bench_gpus_3_starts_three_ranks_and_propagates_their_failure():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=_clean_env(OAT_BENCH_ECHO_RANK="1"))
    assert r.returncode != 0
    seen = sorted(l for l in r.stderr.splitlines() if l.startswith("bench.py rank "))
    assert [l.split()[2] for l in seen] == ["0/3", "1/3", "2/3"], r.stderr[-2000:]
    assert all("HSA_ENABLE_IPC_MODE_LEGACY=0" in l and "MASTER_ADDR=127.0.0.1" in l for l in seen), seen
    assert len({l.split("MASTER_PORT=")[1].split()[0] for l in seen}) == 1                    # one rendezvous for all ranks
    assert "needs an MI355X" in r.stderr and "stopping the other ranks" in r.stderr or "exited with status" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]                         # no line on failure


def test_self_launch_is_a_no_op_under_a_launcher_and_at_one_gpu(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1"])
    assert bench._self_launch() is None
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    monkeypatch.setenv("WORLD_SIZE", "8")                 # torch.distributed.run already made us one of its ranks
    monkeypatch.setenv("RANK", "5")
    assert bench._self_launch() is None
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"          # set by `import bench` / `import OATrans` before any GPU call


def by_variant(plan):
    return {kw.get("label", v): v for v, kw in plan}


def test_other_config_plan_names_every_baseline_config():
    import argparse
    import bench
    plan = bench.other_config_plan(argparse.Namespace(batch=32, frames=8, res=224))
    labels = [kw.get("label", v) for v, kw in plan]
    assert labels == ["config 3, global_local", "config 3, region_mem", "frozen, full graph", "region_mem, full graph", "config 2",
                      "config 4, per-GPU shape", "config 5 geometry, bf16", "config 5 geometry, bf16, twice the batch",
                      "config 5 geometry, fp8 forward"]
    assert by_variant(plan)["frozen, full graph"] == "frozen_full" and by_variant(plan)["region_mem, full graph"] == "region_mem_full"
    by = {kw.get("label", v): (v, kw) for v, kw in plan}
    assert by["config 2"][1]["frames"] == 4 and by["config 4, per-GPU shape"][1]["batch"] == 64
    c5 = by["config 5 geometry, fp8 forward"]
    assert c5[0] == "global_local" and (c5[1]["frames"], c5[1]["res"], c5[1]["batch"], c5[1]["dtype"]) == (16, 336, 8, "fp8")
    assert by["config 5 geometry, bf16, twice the batch"][1]["batch"] == 16
    # FLOP model of config 5's geometry: one object frame + 16 frames of 441 patches, two text passes
    assert 5000 < bench.flops_per_pair(16, N=441, clips=(1, 16), text_passes=2) / 1e9 < 6500


def test_fused_loss_dispatch_accepts_either_import_path_and_rejects_subclasses():
    from OATrans.model.loss import NormSoftmaxLoss
    from OATrans.trainer import step
    assert step._is_norm_softmax(NormSoftmaxLoss(0.07))

    class Mine(NormSoftmaxLoss):                         # may override forward: must take the general path
        pass
    assert not step._is_norm_softmax(Mine())
    # the same class object reached through the other import root (entry points run from inside OATrans/ import `model.loss`)
    twin = type("NormSoftmaxLoss", (torch.nn.Module,), {"__module__": "model.loss", "temperature": 0.05})
    assert step._is_norm_softmax(twin())
    other = type("NormSoftmaxLoss", (torch.nn.Module,), {"__module__": "somewhere.else", "temperature": 0.05})
    assert not step._is_norm_softmax(other())


def test_backward_grid_default_at_more_than_one_rank(monkeypatch):
    from OATrans import parallel
    monkeypatch.delenv("OAT_BWD_NT_GRID", raising=False)
    assert parallel.bwd_nt_grid_default() == 0xffff          # one workgroup per tile: adapts to the CUs RCCL holds
    monkeypatch.setenv("OAT_BWD_NT_GRID", "240")
    assert parallel.bwd_nt_grid_default() == 240
