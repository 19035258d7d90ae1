"""Pin the CPU oracle against outputs of the REFERENCE ITSELF (tests/golden/*.pt,
made by tests/golden/make_golden.py inside the build container).  CPU-only."""
import os

import pytest
import torch

from OATrans.utils import seeded_init as si
from oracle import oatrans_oracle as orc

SEED = 20240917
SMALL_VIDEO = dict(embed_dim=128, depth=2, mlp_ratio=4, num_frames=3, patches_per_frame=9, patch=16)
SMALL_TEXT = dict(dim=128, n_layers=2, hidden_dim=512, vocab=1000, max_pos=64)


def _load(golden_dir, name):
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    return torch.load(path, map_location="cpu", weights_only=False)


def small_inputs(B=2, T=3, R=48, L=7):
    video = si.seeded_tensor(SEED, "in.video", (B, T, 3, R, R), std=1.0)
    ids = si.seeded_ints(SEED, "in.ids", (B, L), 1, SMALL_TEXT["vocab"])
    mask = torch.ones(B, L, dtype=torch.int64)
    mask[1, L - 2:] = 0
    return video, ids, mask


@pytest.mark.parametrize("T", [3, 2])
def test_small_video_forward_and_grads(golden_dir, T):
    g = _load(golden_dir, "small_video.pt")[f"T{T}"]
    p = si.seeded_state_dict(si.video_param_shapes(**SMALL_VIDEO), SEED, "video_model.")
    for v in p.values():
        v.requires_grad_(True)
    video, _, _ = small_inputs(T=T)
    cls, patches, blocks = orc.video_encoder(video, p, num_heads=2, return_blocks=True)
    for i, b in enumerate(blocks):
        assert torch.allclose(b, g["blocks"][i], atol=2e-5, rtol=1e-5), f"block {i}"
    assert torch.allclose(cls, g["cls"], atol=2e-5, rtol=1e-5)
    assert torch.allclose(patches, g["patches"], atol=2e-5, rtol=1e-5)
    gc = si.seeded_tensor(SEED, f"g.cls.{T}", cls.shape)
    gp = si.seeded_tensor(SEED, f"g.patches.{T}", patches.shape, std=0.1)
    ((cls * gc).sum() + (patches * gp).sum()).backward()
    for k, ref in g["grads"].items():
        mine = p["video_model." + k].grad
        if k == "temporal_embed" and T == 2:
            # frames beyond curr_frames get no gradient on either side
            assert torch.count_nonzero(ref[:, 2:]) == 0
        assert mine is not None, k
        scale = ref.abs().max().clamp_min(1e-4)   # k_lin.bias grads are analytically 0 (softmax shift invariance)
        assert (mine - ref).abs().max() / scale < 2e-4, k


def test_video_336_geometry_forward_and_grads(golden_dir):
    """The oracle against the REFERENCE run at 336^2 (441 patches per frame, BASELINE config 5's frame geometry:
    SpaceTimeTransformer(img_size=336), video_transformer.py:195,233-236) - until round 6 the oracle was pinned at 224^2 and 48^2 only."""
    g = _load(golden_dir, "video_336.pt")
    geo = dict(embed_dim=128, depth=2, mlp_ratio=4, num_frames=2, patches_per_frame=441, patch=16)
    p = si.seeded_state_dict(si.video_param_shapes(**geo), SEED, "video_model.")
    for v in p.values():
        v.requires_grad_(True)
    video = si.seeded_tensor(SEED, "in.video.336", (2, 2, 3, 336, 336))
    cls, patches, blocks = orc.video_encoder(video, p, num_heads=2, return_blocks=True)
    for i, b in enumerate(blocks):
        assert torch.allclose(b[:, 0], g["block_cls"][i], atol=2e-5, rtol=1e-5), f"block {i}"
    assert torch.allclose(cls, g["cls"], atol=2e-5, rtol=1e-5)
    assert torch.allclose(patches, g["patches"], atol=2e-5, rtol=1e-5)
    gc = si.seeded_tensor(SEED, "g.cls.336", cls.shape)
    gp = si.seeded_tensor(SEED, "g.patches.336", patches.shape, std=0.05)
    ((cls * gc).sum() + (patches * gp).sum()).backward()
    for k, ref in g["grads"].items():
        mine = p["video_model." + k].grad
        assert mine is not None, k
        scale = ref.abs().max().clamp_min(1e-4)
        assert (mine - ref).abs().max() / scale < 2e-4, k


def test_small_chain(golden_dir):
    g = _load(golden_dir, "small_chain.pt")
    p = si.frozen_state_dict(SEED, SMALL_VIDEO, SMALL_TEXT, proj_dim=64)
    for v in p.values():
        v.requires_grad_(True)
    video, ids, mask = small_inputs(B=4)
    mask[3, 3:] = 0
    assert torch.equal(mask, g["mask"])
    hidden = orc.distilbert(ids, mask, p, n_heads=2)
    # padded positions are don't-care for token-0 pooling but the published algorithm defines them too
    assert torch.allclose(hidden, g["text_hidden"], atol=2e-5, rtol=1e-5)
    loss, sim, t, v = orc.train_step_loss(p, video, ids, mask, num_heads=2, text_heads=2)
    assert torch.allclose(t, g["text"], atol=2e-5, rtol=1e-5)
    assert torch.allclose(v, g["video"], atol=2e-5, rtol=1e-5)
    assert torch.allclose(sim, g["sim"], atol=1e-5)
    assert torch.allclose(loss, g["loss"], atol=1e-5)
    loss.backward()
    for k, ref in g["grads"].items():
        mine = p[k].grad
        assert mine is not None, k
        scale = ref.abs().max().clamp_min(1e-4)   # k_lin.bias grads are analytically 0 (softmax shift invariance)
        assert (mine - ref).abs().max() / scale < 5e-4, k


@pytest.mark.parametrize("name,n", [("sq8", 8), ("sq1", 1), ("sq33", 33)])
def test_loss_cases(golden_dir, name, n):
    g = _load(golden_dir, "loss_cases.pt")[name]
    a = si.seeded_tensor(SEED, f"loss.a.{name}", (n, 16))
    b = si.seeded_tensor(SEED, f"loss.b.{name}", (n, 16))
    if name == "sq8":
        a[2] = 0.0
    a.requires_grad_(True)
    b.requires_grad_(True)
    sim = orc.sim_matrix(a, b)
    loss = orc.norm_softmax_loss(sim)
    loss.backward()
    assert torch.allclose(sim, g["sim"], atol=1e-6)
    assert torch.allclose(loss, g["loss"], atol=1e-5)
    assert torch.allclose(a.grad, g["ga"], atol=1e-5, rtol=1e-4)
    assert torch.allclose(b.grad, g["gb"], atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("T", [1])
def test_full_geometry(golden_dir, T):
    """ViT-B/16 + DistilBERT-base geometry against the reference's contract class
    oa_model.FrozenInTime (T=4 variant runs in the GPU suite; here T=1 keeps the
    CPU suite within minutes)."""
    g = _load(golden_dir, f"full_T{T}.pt")
    torch.set_num_threads(8)
    p = si.frozen_state_dict(SEED, dict(num_frames=T), {})
    for v in p.values():
        v.requires_grad_(True)
    B, L = g["B"], g["L"]
    video = si.seeded_tensor(SEED, f"full.video.{T}", (B, T, 3, 224, 224))
    ids = si.seeded_ints(SEED, f"full.ids.{T}", (B, L), 1000, 30000)
    ids[:, 0] = 101
    loss, sim, t, v = orc.train_step_loss(p, video, ids, g["mask"])
    assert torch.allclose(t, g["text"], atol=5e-5, rtol=1e-4)
    assert torch.allclose(v, g["video"], atol=5e-5, rtol=1e-4)
    assert torch.allclose(sim, g["sim"], atol=1e-5)
    assert torch.allclose(loss, g["loss"], atol=2e-5)
    loss.backward()
    for k, pr in g["grad_probe"].items():
        mine = p[k].grad.flatten()
        if pr["norm"] < 1e-6:        # analytically-zero grads (k_lin.bias): both sides are round-off
            assert mine.norm() < 1e-5, k
            continue
        assert abs(mine.norm() - pr["norm"]) <= 2e-3 * pr["norm"] + 1e-7, k
        scale = pr["norm"] / (mine.numel() ** 0.5) + 1e-9
        assert ((mine[pr["idx"]] - pr["val"]).abs() / scale).max() < 5e-2, k
