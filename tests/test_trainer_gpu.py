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


def test_entry_point_two_ranks_on_one_gpu(tmp_path):
    """The same entry point at world size 2 (two processes share the GPU over gloo; RCCL refuses two ranks on one device):
    DistributedSampler shards, the packed gather, the overlapped gradient all-reduce, validation with the raw all_gather
    and rank-0-only logging / checkpointing (base_trainer.py:95,118,142) all run."""
    cfg = json.load(open(os.path.join(PKG, "configs/pt/synthetic/frozen_1f_bs2.json")))
    cfg["arch"]["args"]["video_params"]["arch_kwargs"] = {"depth": 2}
    cfg["arch"]["args"]["text_params"]["config"] = {"n_layers": 1}
    cfg["trainer"].update(epochs=1, max_samples_per_epoch=8, save_dir=str(tmp_path / "exps"), save_period=1)
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   OAT_ONE_DEVICE="1", OAT_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(PKG, "train_dist_multi.py"), "-c", str(path)], cwd=str(tmp_path),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][-3000:] + outs[1][-3000:]
    assert "val_loss_0" in outs[0] and "Saving checkpoint" in outs[0], outs[0][-2000:]
    assert "Saving checkpoint" not in outs[1], outs[1][-2000:]                      # rank 0 only
    ckpts = [os.path.join(d, f) for d, _, fs in os.walk(tmp_path / "exps") for f in fs if f == "checkpoint-epoch1.pth"]
    assert len(ckpts) == 1, ckpts


@pytest.mark.parametrize("entry,config", [("train_dist_region_mem.py", "region_mem_2f.json"),
                                           ("train_dist_multi_global_local.py", "global_local_2f.json")])
@pytest.mark.parametrize("world", [1, 2])
def test_oa_entry_points_train_an_epoch(tmp_path, entry, config, world):
    """The object-aware entry points through their trainers: one epoch on the synthetic loader, at one
    rank and at two ranks sharing the GPU over gloo (4 and 6 tensors in ONE packed all-gather per step)."""
    cfg = json.load(open(os.path.join(PKG, "configs/pt/synthetic", config)))
    depth = 6 if "region" in config else 2                   # the region variant taps block 6
    cfg["arch"]["args"]["video_params"]["arch_kwargs"] = {"depth": depth}
    cfg["arch"]["args"]["text_params"]["config"] = {"n_layers": 1}
    cfg["trainer"].update(epochs=1, max_samples_per_epoch=16, save_dir=str(tmp_path / "exps"), save_period=1)
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    port = str(_free_port())
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   OAT_ONE_DEVICE="1", OAT_DIST_BACKEND="gloo" if world > 1 else "nccl")
        procs.append(subprocess.Popen([sys.executable, os.path.join(PKG, entry), "-c", str(path)], cwd=str(tmp_path), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n=====\n".join(o[-3000:] for o in outs)
    assert "Saving checkpoint" in outs[0], outs[0][-2000:]


def test_validation_epoch_reports_retrieval_metrics(tmp_path):
    """trainer_dist.py:201-281 of the reference: _valid_epoch gathers the embeddings of the whole validation set, runs
    config['metrics'] on ONE sim matrix and returns `nested_val_metrics`; train() flattens them to
    val_{loader}_{metric}_{name} so that a monitor such as 'max val_0_t2v_metrics_R1' finds its key.  The numbers must
    equal model/metric.py applied to the embeddings of the HIP path."""
    import argparse
    import sys as _sys
    _sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd"))
    from OATrans import model as module_arch
    from OATrans.model import metric as module_metric
    from OATrans.optim import AdamW
    from OATrans.parse_config import ConfigParser
    from OATrans.trainer.trainer_dist import Multi_Trainer_dist

    cfg = json.load(open(os.path.join(PKG, "configs/pt/synthetic/frozen_1f_bs2.json")))
    cfg["arch"]["args"]["video_params"]["arch_kwargs"] = {"depth": 2}
    cfg["arch"]["args"]["text_params"]["config"] = {"n_layers": 1}
    cfg["trainer"].update(epochs=1, max_samples_per_epoch=8, save_dir=str(tmp_path / "exps"), save_period=1,
                          monitor="max val_0_t2v_metrics_R1")
    cfg["metrics"] = ["t2v_metrics", "v2t_metrics"]
    config = ConfigParser(cfg)
    args = argparse.Namespace(world_size=1, rank=0, local_rank=0, learning_rate1=1e-5, schedule=[60, 80])
    torch.manual_seed(0)
    model = config.initialize("arch", module_arch)

    class Loader(list):
        batch_size, n_samples, dataset_name = 2, 6, "SyntheticVal"

        class _S:
            @staticmethod
            def set_epoch(e):
                pass
        train_sampler = _S()

    def make(seed, n):
        g = torch.Generator().manual_seed(seed)
        return Loader({"video": torch.randn(2, 1, 3, 224, 224, generator=g),
                       "text": {"input_ids": torch.randint(1000, 30000, (2, 9), generator=g),
                                "attention_mask": torch.ones(2, 9, dtype=torch.int64)}} for _ in range(n))

    train_dl, val_dl = make(1, 2), make(2, 3)
    loss = module_arch.NormSoftmaxLoss()
    metrics = [getattr(module_metric, m) for m in cfg["metrics"]]
    opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-5)
    trainer = Multi_Trainer_dist(args, model, loss, metrics, opt, config=config, data_loader=[train_dl],
                                 valid_data_loader=[val_dl], tokenizer=None, max_samples_per_epoch=8)
    res = trainer._valid_epoch(0)
    assert set(res) == {"val_loss_0", "nested_val_metrics"}
    nested = res["nested_val_metrics"][0]
    assert set(nested) == {"t2v_metrics", "v2t_metrics"}
    # the same numbers from the embeddings of the HIP path, batch by batch, through metric.py
    trainer.model.eval()
    with torch.no_grad():
        ts, vs = zip(*[trainer.model.module(trainer._to_device(dict(d, text=dict(d["text"]))), return_embeds=True) for d in val_dl])
    sims = module_arch.sim_matrix(torch.cat(ts), torch.cat(vs)).cpu().numpy()
    assert sims.shape == (6, 6)
    for name, fn in (("t2v_metrics", module_metric.t2v_metrics), ("v2t_metrics", module_metric.v2t_metrics)):
        want = fn(sims)
        for k, v in want.items():
            assert abs(nested[name][k] - v) < 1e-9, (name, k, nested[name][k], v)
    # and train() exposes them under the monitor's key
    trainer.init_val = False
    trainer.train()
    assert trainer.mnt_best != -float("inf"), "monitor 'max val_0_t2v_metrics_R1' never saw its key"


def test_train_py_config1_full_depth(tmp_path):
    """BASELINE config 1 through its own entry point: train.py -c frozen_1f_bs2.json (1 frame 224^2, ViT-B/16 with all 12
    blocks + DistilBERT-base with all 6 layers, bs 2) trains an epoch, validates with the retrieval metrics and writes a
    checkpoint (reference: train.py + trainer/trainer.py, the single-process path)."""
    cfg = json.load(open(os.path.join(PKG, "configs/pt/synthetic/frozen_1f_bs2.json")))
    cfg["trainer"].update(epochs=1, max_samples_per_epoch=8, save_dir=str(tmp_path / "exps"), save_period=1)
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    r = subprocess.run([sys.executable, os.path.join(PKG, "train.py"), "-c", str(path)], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "val_loss_0" in out and "Saving checkpoint" in out, out[-2000:]
    assert "t2v_metrics" in out, out[-2000:]
    ckpts = [os.path.join(d, f) for d, _, fs in os.walk(tmp_path / "exps") for f in fs if f == "checkpoint-epoch1.pth"]
    assert len(ckpts) == 1, ckpts
    ck = torch.load(ckpts[0], map_location="cpu", weights_only=False)
    keys = list(ck["state_dict"])
    assert "video_model.blocks.11.mlp.fc2.weight" in keys and "text_model.transformer.layer.5.ffn.lin2.weight" in keys
    assert all(torch.isfinite(v).all() for v in ck["state_dict"].values() if v.is_floating_point())
