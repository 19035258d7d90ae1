"""End-to-end run of the reference-named entry point on one MI355X:
train_dist_multi.py -c <config> trains an epoch, validates, writes a checkpoint in the reference's layout
(base_trainer.py:163-244), and -r <checkpoint> resumes from it."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "oa-transformer_amd", "OATrans")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(args, cwd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(PKG, "train_dist_multi.py")] + args, cwd=cwd, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout + r.stderr


def test_entry_point_trains_checkpoints_and_resumes(tmp_path):
    cfg = json.load(open(os.path.join(PKG, "configs/pt/synthetic/frozen_1f_bs2.json")))
    cfg["arch"]["args"]["video_params"]["arch_kwargs"] = {"depth": 2}            # small towers: the plumbing is the subject
    cfg["arch"]["args"]["text_params"]["config"] = {"n_layers": 1}
    cfg["trainer"].update(epochs=1, max_samples_per_epoch=8, save_dir=str(tmp_path / "exps"), save_period=1)
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    out = _run(["-c", str(path)], str(tmp_path))
    assert "val_loss_0" in out and "Saving checkpoint" in out, out[-2000:]
    ckpts = [os.path.join(d, f) for d, _, fs in os.walk(tmp_path / "exps") for f in fs if f == "checkpoint-epoch1.pth"]
    assert len(ckpts) == 1, ckpts
    ck = torch.load(ckpts[0], map_location="cpu", weights_only=False)
    assert sorted(ck) == ["arch", "config", "epoch", "monitor_best", "optimizer", "state_dict"]      # reference layout
    assert ck["arch"] == "FrozenInTime" and ck["epoch"] == 1
    # the distributed trainer saves the wrapped model: `module.` prefix, as the reference's DDP checkpoints have
    keys = [k[len("module."):] if k.startswith("module.") else k for k in ck["state_dict"]]
    assert "video_model.blocks.0.timeattn.qkv.weight" in keys and "text_model.transformer.layer.0.attention.q_lin.weight" in keys
    assert "txt_proj.1.weight" in keys and "vid_proj.0.weight" in keys
    assert all(torch.isfinite(v).all() for v in ck["state_dict"].values() if v.is_floating_point())
    out = _run(["-r", ckpts[0]], str(tmp_path))
    assert "Checkpoint loaded. Resume training from epoch 2" in out, out[-2000:]
