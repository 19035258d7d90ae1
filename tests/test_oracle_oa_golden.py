"""Pin the oracle's OA-variant restatements against outputs of the reference's own
oa_model_region_mem.FrozenInTime and oa_model_global_local.FrozenInTime (tests/golden/oa_*.pt). CPU-only."""
import os

import pytest
import torch

from OATrans.utils import seeded_init as si
from oracle import oatrans_oracle as orc

SEED = 20240917


def oa_inputs(B=2, F=2, L=8, Lp=12, O=3, K=5):
    video = si.seeded_tensor(SEED, "oa.video", (B, F, 3, 224, 224))
    ids = si.seeded_ints(SEED, "oa.ids", (B, L), 1000, 30000)
    ids[:, 0] = 101
    mask = torch.ones(B, L, dtype=torch.int64)
    mask[1, L - 2:] = 0
    pids = si.seeded_ints(SEED, "oa.pids", (B, Lp), 1000, 30000)
    pids[:, 0] = 101
    pmask = torch.ones(B, Lp, dtype=torch.int64)
    pmask[1, Lp - 1:] = 0
    patch_masks = (si.seeded_tensor(SEED, "oa.pm", (B, O, 196)) > 0.3).float()
    region_masks = (si.seeded_tensor(SEED, "oa.rm", (B, K, 196)) > 0.5).float()
    otm = torch.tensor([[1, 3, 4], [2, 3, 5]], dtype=torch.int64)[:B, :O]
    treg = si.seeded_tensor(SEED, "oa.treg", (B, K, 512))
    return dict(video=video, ids=ids, mask=mask, pids=pids, pmask=pmask, patch_masks=patch_masks,
                region_masks=region_masks, otm=otm, treg=treg)


def region_params():
    sd = si.frozen_state_dict(SEED, dict(num_frames=1), {})
    sd.update(si.seeded_state_dict({"video_model.region_norm.weight": (768,), "video_model.region_norm.bias": (768,),
                                    "video_model.object_embed.weight": (768, 2054), "video_model.object_embed.bias": (768,),
                                    "txt_proj_2.1.weight": (256, 512), "txt_proj_2.1.bias": (256,)}, SEED))
    return sd


def gl_params():
    sd = si.frozen_state_dict(SEED, dict(num_frames=1), {})
    sd.update(si.seeded_state_dict({"video_model.object_embed.weight": (768, 2054), "video_model.object_embed.bias": (768,),
                                    "text_local_proj.1.weight": (256, 768), "text_local_proj.1.bias": (256,),
                                    "vid_local_proj.0.weight": (256, 768), "vid_local_proj.0.bias": (256,)}, SEED))
    return sd


def check_probe(p, probe, skip=()):
    for k, pr in probe.items():
        if pr["norm"] < 1e-6 or k in skip:
            continue
        g = p[k].grad
        assert g is not None, k
        assert abs(g.norm() - pr["norm"]) <= 2e-3 * pr["norm"] + 1e-7, k
        scale = pr["norm"] / g.numel() ** 0.5 + 1e-9
        assert ((g.flatten()[pr["idx"]] - pr["val"]).abs() / scale).max() < 5e-2, k


def _golden(golden_dir, name):
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    return torch.load(path, map_location="cpu", weights_only=False)


def test_region_mem_vs_reference(golden_dir):
    g = _golden(golden_dir, "oa_region_mem.pt")
    torch.set_num_threads(8)
    p = region_params()
    for v in p.values():
        v.requires_grad_(True)
    d = oa_inputs()
    t, v, rsim = orc.region_mem_forward(p, d["video"], d["ids"], d["mask"], d["treg"])
    assert torch.allclose(t, g["text"], atol=5e-5, rtol=1e-4)
    assert torch.allclose(v, g["video"], atol=5e-5, rtol=1e-4)
    assert torch.allclose(rsim, g["region_sim"], atol=1e-5)
    loss = orc.region_mem_loss(t, v, rsim, d["region_masks"])
    assert torch.allclose(loss, g["loss"], atol=1e-4, rtol=1e-5)
    loss.backward()
    check_probe(p, g["grad_probe"])


def test_global_local_vs_reference(golden_dir):
    g = _golden(golden_dir, "oa_global_local.pt")
    torch.set_num_threads(8)
    p = gl_params()
    for v in p.values():
        v.requires_grad_(True)
    d = oa_inputs()
    t, pt, v, ov, rf, tf = orc.gl_forward(p, d["video"], (d["ids"], d["mask"]), (d["pids"], d["pmask"]),
                                          d["patch_masks"], d["otm"])
    for mine, key in ((t, "text"), (pt, "pad_text"), (v, "video"), (ov, "object_video"), (rf, "region_feat"), (tf, "tags_feat")):
        assert torch.allclose(mine, g[key], atol=2e-4, rtol=1e-4), key
    loss = orc.gl_loss(t, pt, v, rf, tf)
    assert torch.allclose(loss, g["loss"], atol=1e-4, rtol=1e-5)
    loss.backward()
    check_probe(p, g["grad_probe"])


def test_native_clip_layout_is_the_reference_at_two_frames():
    """The 'native' object-clip layout (object frame + T-frame video, BASELINE config 3) has no reference line of its
    own; at F = 2 it must be the same computation as the reference's view(2B, F/2) - which the goldens above pin."""
    torch.set_num_threads(8)
    d = oa_inputs()
    with torch.no_grad():
        p = gl_params()
        a = orc.gl_forward(p, d["video"], (d["ids"], d["mask"]), (d["pids"], d["pmask"]), d["patch_masks"], d["otm"])
        b = orc.gl_forward(p, d["video"], (d["ids"], d["mask"]), (d["pids"], d["pmask"]), d["patch_masks"], d["otm"],
                           object_clip="native")
        for x, y in zip(a, b):
            assert torch.allclose(x, y, atol=2e-5, rtol=1e-5)
        p = region_params()
        a = orc.region_mem_forward(p, d["video"], d["ids"], d["mask"], d["treg"])
        b = orc.region_mem_forward(p, d["video"], d["ids"], d["mask"], d["treg"], object_clip="native")
        for x, y in zip(a, b):
            assert torch.allclose(x, y, atol=2e-5, rtol=1e-5)


def test_tag_masks_matches_the_reference_loop():
    otm = torch.tensor([[1, 3, 4], [2, 3, 5]])
    n_txt = torch.tensor([8, 6])
    tm = orc.tag_masks(otm, n_txt, 12)
    ref = torch.zeros(2, 3, 12)
    for j in range(2):                       # literal semantics of oa_model_global_local.py:189-196
        start = 0
        for k in range(3):
            ref[j][k][int(n_txt[j]) - 1 + start:int(n_txt[j]) - 1 + int(otm[j][k])] = 1
            start = int(otm[j][k])
    assert torch.equal(tm, ref)


def test_patch_masks_from_bbox_vs_reference(golden_dir):
    """bbox -> 14x14 patch masks: the oracle against vectors produced by the reference's own method bodies
    (tests/golden/make_golden_masks.py)."""
    import os
    import torch
    from oracle import oatrans_oracle as orc
    g = torch.load(os.path.join(golden_dir, "oa_patch_masks.pt"), map_location="cpu", weights_only=False)
    for c in g["global_local"]:
        assert torch.equal(orc.patch_masks_from_bbox(c["bbox"]), c["masks"])
    for c in g["region_mem"]:
        assert torch.equal(orc.patch_masks_from_bbox(c["bbox"], box_class=c["box_class"], sel_class=c["sel_class"]), c["masks"])
