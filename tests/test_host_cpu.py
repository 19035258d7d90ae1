"""Host-side logic that needs no GPU: config parser / reflection factory, synthetic loaders,
retrieval metrics, the seeded generator, checkpoint-key compatibility, positional inflation."""
import argparse
import collections
import json
import sys

import numpy as np
import pytest
import torch


def _parser():
    p = argparse.ArgumentParser()
    p.add_argument('-c', '--config', default=None, type=str)
    p.add_argument('-r', '--resume', default=None, type=str)
    p.add_argument('-d', '--device', default=None, type=str)
    return p


def test_config_parser_overrides_and_factory(tmp_path, monkeypatch):
    from OATrans.parse_config import ConfigParser
    from OATrans.data_loader import data_loader as module_data
    import os
    cfg_path = os.path.join(os.path.dirname(module_data.__file__), "..", "configs", "pt", "synthetic", "frozen_8f.json")
    cfg = json.load(open(cfg_path))
    cfg["trainer"]["save_dir"] = str(tmp_path)
    local = tmp_path / "cfg.json"
    local.write_text(json.dumps(cfg))
    monkeypatch.setattr(sys, "argv", ["x", "-c", str(local), "--lr", "0.5", "--bs", "4"])
    CustomArgs = collections.namedtuple('CustomArgs', 'flags type target')
    options = [CustomArgs(['--lr', '--learning_rate'], type=float, target=('optimizer', 'args', 'lr')),
               CustomArgs(['--bs', '--batch_size'], type=int, target=('data_loader', 0, 'args', 'batch_size'))]
    config = ConfigParser(_parser(), options)
    assert config['optimizer']['args']['lr'] == 0.5
    assert config['data_loader'][0]['args']['batch_size'] == 4
    assert (config.save_dir / 'config.json').exists()
    # reflection factory + `args` injection for the loader class (parse_config_dist_multi.py:93-98)
    dl = config.initialize('data_loader', module_data, index=0)
    assert dl.batch_size == 4 and dl.args is config.args and dl.dataset_name == "Synthetic"
    batch = next(iter(dl))
    assert batch['video'].shape == (4, 8, 3, 224, 224)
    assert batch['text']['input_ids'].shape == (4, 32) and batch['text']['input_ids'].dtype == torch.int64
    assert hasattr(dl, 'n_samples') and hasattr(dl.train_sampler, 'set_epoch')
    # kwargs may not overwrite config args
    with pytest.raises(AssertionError):
        config.initialize('optimizer', __import__('OATrans.optim', fromlist=['x']), [], lr=1.0)


def test_arch_factory_builds_contract_class(tmp_path, monkeypatch):
    from OATrans import model as module_arch
    from OATrans.parse_config import ConfigParser
    cfg = {"name": "t", "n_gpu": 1, "arch": {"type": "FrozenInTime", "args": {
        "video_params": {"model": "SpaceTimeTransformer", "arch_config": "base_patch16_224", "num_frames": 2, "pretrained": True, "time_init": "zeros"},
        "object_params": {"model": "", "input_objects": False},
        "text_params": {"model": "pretrained/distilbert-base-uncased", "pretrained": True, "input": "text"},
        "projection": "minimal", "load_checkpoint": ""}},
        "trainer": {"save_dir": str(tmp_path), "verbosity": 2}}
    p = tmp_path / "c.json"
    p.write_text(json.dumps(cfg))
    monkeypatch.setattr(sys, "argv", ["x", "-c", str(p)])
    m = ConfigParser(_parser()).initialize('arch', module_arch)
    keys = set(m.state_dict().keys())
    # reference checkpoint key names
    for k in ("video_model.cls_token", "video_model.pos_embed", "video_model.temporal_embed", "video_model.patch_embed.proj.weight",
              "video_model.blocks.11.timeattn.qkv.weight", "video_model.blocks.0.attn.proj.bias", "video_model.blocks.3.mlp.fc2.weight",
              "video_model.blocks.5.norm3.weight", "video_model.norm.bias", "text_model.embeddings.word_embeddings.weight",
              "text_model.transformer.layer.5.ffn.lin2.bias", "text_model.transformer.layer.0.attention.q_lin.weight",
              "txt_proj.1.weight", "vid_proj.0.bias"):
        assert k in keys, k
    assert sum(p.numel() for p in m.parameters()) == 180922112 + 2 * 768
    # time_init='zeros': qkv = 0, proj.weight = 1 (video_transformer.py:89-95)
    blk = m.video_model.blocks[0].timeattn
    assert torch.count_nonzero(blk.qkv.weight) == 0 and torch.all(blk.proj.weight == 1)
    # no CPU fallback: the product path refuses to run off-GPU
    from OATrans.ops.hip import OatError
    with pytest.raises(OatError):
        m.video_model(torch.zeros(1, 2, 3, 224, 224))


def test_inflate_positional_embeds():
    from OATrans import model as module_arch
    m = module_arch.FrozenInTime(dict(model="SpaceTimeTransformer", num_frames=4, pretrained=True),
                                 dict(model="", input_objects=False),
                                 dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text"))
    te = torch.arange(2 * 768, dtype=torch.float32).view(1, 2, 768)
    out = m._inflate_positional_embeds({"video_model.temporal_embed": te.clone()})["video_model.temporal_embed"]
    assert out.shape == (1, 4, 768) and torch.equal(out[:, :2], te) and torch.count_nonzero(out[:, 2:]) == 0
    te6 = torch.randn(1, 6, 768)
    out = m._inflate_positional_embeds({"video_model.temporal_embed": te6.clone()})["video_model.temporal_embed"]
    assert torch.equal(out, te6[:, :4])
    m.load_temporal_fix = 'interp'
    out = m._inflate_positional_embeds({"video_model.temporal_embed": te.clone()})["video_model.temporal_embed"]
    assert out.shape == (1, 4, 768) and torch.equal(out[0, 0], te[0, 0]) and torch.equal(out[0, 3], te[0, 1])
    with pytest.raises(NotImplementedError):
        m._inflate_positional_embeds({"video_model.pos_embed": torch.zeros(1, 50, 768)})


def test_state_dict_data_parallel_fix():
    from OATrans.utils.util import state_dict_data_parallel_fix
    cur = {"a.w": 0, "b.w": 0}
    assert list(state_dict_data_parallel_fix({"module.a.w": 1, "module.b.w": 2}, cur)) == ["a.w", "b.w"]
    assert list(state_dict_data_parallel_fix({"a.w": 1}, {"module.a.w": 0})) == ["module.a.w"]
    assert list(state_dict_data_parallel_fix({"a.w": 1}, cur)) == ["a.w"]


def test_retrieval_metrics_vs_reference_golden(golden_dir):
    """t2v / v2t metrics against the outputs of the reference's own metric.py (tests/golden/make_golden_metrics.py):
    optimistic ties for t2v, averaged ties for v2t, R1 = exact rank 0, several captions per video, query masks."""
    import os
    from OATrans.model.metric import t2v_metrics, v2t_metrics
    cases = torch.load(os.path.join(golden_dir, "metrics.pt"), map_location="cpu", weights_only=False)
    assert len(cases) >= 6
    for c in cases:
        sims = c["sims"].numpy()
        masks = None if c["masks"] is None else c["masks"].numpy()
        for fn, want in ((t2v_metrics, c["t2v"]), (v2t_metrics, c["v2t"])):
            got = fn(sims.copy(), None if masks is None else masks.copy())
            assert set(got) == set(want), c["name"]
            for k in want:
                assert abs(got[k] - want[k]) <= 1e-9 * max(1.0, abs(want[k])), (c["name"], fn.__name__, k, got[k], want[k])


def test_seeded_generator_is_a_pure_function_of_name():
    from OATrans.utils import seeded_init as si
    a = si.seeded_tensor(1, "x.weight", (5, 7), std=0.5)
    assert torch.equal(a, si.seeded_tensor(1, "x.weight", (5, 7), std=0.5))
    assert not torch.equal(a, si.seeded_tensor(1, "y.weight", (5, 7), std=0.5))
    assert not torch.equal(a, si.seeded_tensor(2, "x.weight", (5, 7), std=0.5))
    big = si.seeded_tensor(3, "z", (200000,), std=2.0, mean=1.0)
    assert abs(big.mean().item() - 1.0) < 0.02 and abs(big.std().item() - 2.0) < 0.02
    ints = si.seeded_ints(3, "i", (1000,), 5, 9)
    assert ints.min() >= 5 and ints.max() <= 8
    shapes = si.video_param_shapes(num_frames=8)
    text = si.text_param_shapes()
    n = sum(int(np.prod(s)) for s in shapes.values()) + sum(int(np.prod(s)) for s in text.values()) + 2 * (768 * 256 + 256)
    assert n == 180922112 + 8 * 768                     # == the instantiated reference at 8 frames


def test_checkpoint_interop_vs_reference_golden(golden_dir):
    """Temporal-embedding inflation (all three fill modes, more / fewer / equal frames) and the `module.` prefix fix
    against outputs of the reference's own functions (tests/golden/make_golden_ckpt.py)."""
    import os
    from collections import OrderedDict
    from OATrans.model.oa_model import FrozenInTime
    from OATrans.utils.util import state_dict_data_parallel_fix
    g = torch.load(os.path.join(golden_dir, "ckpt_interop.pt"), map_location="cpu", weights_only=False)

    class Self:
        def __init__(self, frames, fix, sd):
            self.video_params = {"model": "SpaceTimeTransformer", "num_frames": frames}
            self.load_temporal_fix, self._sd = fix, sd

        def state_dict(self):
            return self._sd

    for c in g["inflate"]:
        D = c["load"].shape[2]
        cur = {"video_model.temporal_embed": torch.zeros(1, c["frames"], D), "video_model.pos_embed": torch.zeros(1, 10, D)}
        out = FrozenInTime._inflate_positional_embeds(Self(c["frames"], c["mode"], cur),
                                                      {"video_model.temporal_embed": c["load"].clone(),
                                                       "video_model.pos_embed": torch.zeros(1, 10, D)})
        assert torch.equal(out["video_model.temporal_embed"], c["out"]), (c["frames"], c["mode"])
    for c in g["prefix_fix"]:
        out = state_dict_data_parallel_fix(OrderedDict((k, i) for i, k in enumerate(c["load"])), OrderedDict((k, 0) for k in c["cur"]))
        assert list(out.items()) == c["out"], c


REF_CONFIGS = "/root/reference/OATrans/configs"


@pytest.mark.skipif(not __import__("os").path.isdir(REF_CONFIGS), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("rel", ["pt/cc3m_webvid/norm.json", "pt/cc3m_webvid/local-region-loss.json",
                                 "ft/msrvtt/zsl/normal.json", "ft/msrvtt/fine_tune/normal_1_cl.json"])
def test_reference_shipped_configs_parse_and_build(tmp_path, monkeypatch, rel):
    """The four JSON configs the reference ships (read in place, not copied) go through ConfigParser unchanged: the
    schema, the `-c` path, the arch / loss / optimizer / data_loader factories with `args` injection, and the loader
    kwargs of data_loader/data_loader.py:165-227 (incl. `cut`) are all the reference's."""
    import os
    from OATrans import model as module_arch
    from OATrans import optim as module_optim
    from OATrans.data_loader import data_loader as module_data
    from OATrans.parse_config import ConfigParser
    cfg = json.load(open(os.path.join(REF_CONFIGS, rel)))
    cfg["trainer"]["save_dir"] = str(tmp_path)
    local = tmp_path / "cfg.json"
    local.write_text(json.dumps(cfg))
    monkeypatch.setattr(sys, "argv", ["x", "-c", str(local)])
    config = ConfigParser(_parser())
    assert config["arch"]["type"] == "FrozenInTime" and config["loss"]["type"] == "NormSoftmaxLoss"
    loss = config.initialize("loss", module_arch)
    assert type(loss).__name__ == "NormSoftmaxLoss"
    metrics = [getattr(__import__("OATrans.model.metric", fromlist=["x"]), m) for m in config["metrics"]]
    assert all(callable(m) for m in metrics)
    w = torch.nn.Parameter(torch.zeros(3))
    opt = config.initialize("optimizer", module_optim, [w])
    assert opt.param_groups[0]["lr"] == cfg["optimizer"]["args"]["lr"]
    n = len(config["data_loader"]) if isinstance(config["data_loader"], list) else 1
    for i in range(n):
        node = config["data_loader"][i] if isinstance(config["data_loader"], list) else config["data_loader"]
        assert node["type"] in ("MultiDistTextObjectVideoDataLoader", "TextObjectVideoDataLoader"), node["type"]
        sig = __import__("inspect").signature(getattr(module_data, node["type"]).__init__)
        unknown = [k for k in node["args"] if k not in sig.parameters]
        assert not unknown, (rel, unknown)                      # every kwarg the reference passes is accepted
    vp = config["arch"]["args"]["video_params"]
    assert vp["model"] == "SpaceTimeTransformer" and vp["arch_config"] == "base_patch16_224"


def test_frame_transform_host_logic_and_oracle_identities():
    """data_loader/frames.py host side (crop-box sampling, shorter-side resize arithmetic) and oracle/frames_oracle.py
    identities: resizing to the same size is the identity, the eval pipeline of a 256 x 256 frame is one resize."""
    import random
    import torch
    from OATrans.data_loader import frames as fr
    from oracle import frames_oracle as forc
    rng = random.Random(0)
    for _ in range(200):
        H, W = rng.randint(32, 720), rng.randint(32, 1280)
        x0, y0, w, h = fr.random_resized_crop_params(H, W, (0.5, 1.0), rng=rng)
        assert 0 <= x0 and 0 <= y0 and x0 + w <= W and y0 + h <= H and w > 0 and h > 0
        fallback = (w, h) in ((W, H), (W, int(round(W / 0.75))), (int(round(H * 4 / 3)), H))    # torchvision's central crop
        assert fallback or (0.45 * H * W <= w * h <= H * W and 3 / 4 - 0.05 <= w / h <= 4 / 3 + 0.05)
    assert fr.resize_shorter_side(360, 640, 256) == (256, 455) and fr.resize_shorter_side(640, 360, 256) == (455, 256)
    f = torch.randint(0, 256, (2, 224, 224, 3), dtype=torch.uint8)
    assert torch.allclose(forc.oa_clip(f, 224), forc.normalize(forc.to_float_chw(f)), atol=1e-6)
    f = torch.randint(0, 256, (2, 256, 256, 3), dtype=torch.uint8)
    assert torch.allclose(forc.eval_clip(f, 224), forc.normalize(forc.resize(forc.to_float_chw(f), (224, 224))), atol=1e-6)
