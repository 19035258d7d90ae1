#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.pt

The reference (/root/reference) is imported with import shims (SURVEY.md
Appendix B) - nothing from it is copied into the repo; the committed fixtures
hold only numeric inputs/outputs.  Weights and inputs are regenerated on both
sides from OATrans.utils.seeded_init (a pure function of tensor *name*), so the
fixtures stay small: expected outputs + metadata.

The GPU box never runs this file (it has no /root/reference).
"""
import os
import sys
import types
import tempfile
import importlib.util

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd"))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from OATrans.utils import seeded_init as si  # noqa: E402  (build-owned generator)


# ----------------------------------------------------------------------------- reference import shims
def import_reference():
    """Make `OATrans.model.*` of the reference importable (timm/ipdb/humanize stubs,
    synthetic OATrans.base that avoids cv2/av/decord)."""
    for k in [k for k in sys.modules if k == "OATrans" or k.startswith("OATrans.")]:
        del sys.modules[k]
    sys.path = [p for p in sys.path if "oa-transformer_amd" not in p]
    sys.path.insert(0, REF)
    import transformers  # noqa: F401  import BEFORE the timm stub exists (it probes find_spec('timm'))
    from transformers import AutoModel, DistilBertModel  # noqa: F401  force the lazy modules

    def trunc_normal_(t, std=1.0, **kw):
        return nn.init.trunc_normal_(t, std=std)

    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_layers = types.ModuleType("timm.models.layers")
    timm_layers.DropPath = nn.Identity
    timm_layers.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
    timm_layers.trunc_normal_ = trunc_normal_
    timm.models = timm_models
    timm_models.layers = timm_layers
    sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.layers": timm_layers})
    for name in ("ipdb", "humanize"):
        sys.modules.setdefault(name, types.ModuleType(name))

    import OATrans  # the reference package (namespace root)
    base = types.ModuleType("OATrans.base")
    base.__path__ = []
    for fname in ("base_model", "base_trainer"):
        spec = importlib.util.spec_from_file_location(f"OATrans.base.{fname}", f"{REF}/OATrans/base/{fname}.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"OATrans.base.{fname}"] = mod
        try:
            spec.loader.exec_module(mod)
        except Exception as e:  # base_trainer drags optional deps; only BaseModel is required
            print(f"[shim] {fname}: {type(e).__name__}: {e}")
            continue
        for attr in dir(mod):
            if attr.startswith(("Base", "Multi_Base")):
                setattr(base, attr, getattr(mod, attr))
    sys.modules["OATrans.base"] = base
    OATrans.base = base
    return OATrans


def restore_build_path():
    for k in [k for k in sys.modules if k == "OATrans" or k.startswith("OATrans.")]:
        del sys.modules[k]
    sys.path = [p for p in sys.path if p != REF]
    sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd"))


# ----------------------------------------------------------------------------- cases
SMALL_VIDEO = dict(embed_dim=128, depth=2, mlp_ratio=4, num_frames=3, patches_per_frame=9, patch=16)
SMALL_TEXT = dict(dim=128, n_layers=2, hidden_dim=512, vocab=1000, max_pos=64)
SMALL_HEADS = 2
SEED = 20240917


def small_inputs(B=2, T=3, R=48, L=7, seed=SEED):
    video = si.seeded_tensor(seed, "in.video", (B, T, 3, R, R), std=1.0)
    ids = si.seeded_ints(seed, "in.ids", (B, L), 1, SMALL_TEXT["vocab"])
    mask = torch.ones(B, L, dtype=torch.int64)
    mask[1, L - 2:] = 0              # ragged: second caption is 2 tokens shorter
    return video, ids, mask


def gen_small_video(SpaceTimeTransformer):
    """Reference SpaceTimeTransformer on the small geometry: outputs, per-block
    activations, and gradients of a fixed linear functional."""
    out = {}
    for T in (3, 2):                 # T=2 < num_frames exercises `curr_frames <= num_frames`
        torch.manual_seed(0)
        m = SpaceTimeTransformer(img_size=48, patch_size=16, embed_dim=128, depth=2, num_heads=SMALL_HEADS,
                                 num_frames=3, time_init="rand")
        m.head = nn.Identity()
        m.pre_logits = nn.Identity()
        sd = si.seeded_state_dict(si.video_param_shapes(**SMALL_VIDEO), SEED, "video_model.")
        sd = {k[len("video_model."):]: v for k, v in sd.items()}     # names are hashed WITH the prefix
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert all(k.startswith("head") for k in missing), missing
        m.eval()
        video, _, _ = small_inputs(T=T)
        blocks = []
        hooks = [b.register_forward_hook(lambda mod, i, o: blocks.append(o.detach().clone())) for b in m.blocks]
        for prm in m.parameters():
            prm.requires_grad_(True)
        cls, patches = m(video)
        for h in hooks:
            h.remove()
        gc = si.seeded_tensor(SEED, f"g.cls.{T}", cls.shape)
        gp = si.seeded_tensor(SEED, f"g.patches.{T}", patches.shape, std=0.1)
        (cls * gc).sum().add((patches * gp).sum()).backward()
        grads = {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}
        out[f"T{T}"] = dict(cls=cls.detach(), patches=patches.detach(), blocks=blocks, grads=grads)
    return out


def gen_video_336(SpaceTimeTransformer):
    """Reference SpaceTimeTransformer(img_size=336): 21 x 21 = 441 patches per frame (BASELINE config 5's frame geometry; the class
    takes img_size, video_transformer.py:195,233-236) on the narrow test width (128-d, 2 blocks, 2 heads, 2 frames, B = 2) - the
    inputs and the linear functional of tests/test_engine_gpu.py::test_336_geometry_vs_oracle.  Outputs, per-block activations of the
    CLS rows and every parameter gradient."""
    geo = dict(embed_dim=128, depth=2, mlp_ratio=4, num_frames=2, patches_per_frame=441, patch=16)
    torch.manual_seed(0)
    m = SpaceTimeTransformer(img_size=336, patch_size=16, embed_dim=128, depth=2, num_heads=2, num_frames=2, time_init="rand")
    m.head = nn.Identity()
    m.pre_logits = nn.Identity()
    sd = si.seeded_state_dict(si.video_param_shapes(**geo), SEED, "video_model.")
    missing, unexpected = m.load_state_dict({k[len("video_model."):]: v for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.startswith("head") for k in missing), (missing, unexpected)
    m.eval()
    video = si.seeded_tensor(SEED, "in.video.336", (2, 2, 3, 336, 336))
    blocks = []
    hooks = [b.register_forward_hook(lambda mod, i, o: blocks.append(o[:, 0].detach().clone())) for b in m.blocks]
    cls, patches = m(video)
    for h in hooks:
        h.remove()
    assert patches.shape == (2, 2 * 441, 128), patches.shape
    gc = si.seeded_tensor(SEED, "g.cls.336", cls.shape)
    gp = si.seeded_tensor(SEED, "g.patches.336", patches.shape, std=0.05)
    ((cls * gc).sum() + (patches * gp).sum()).backward()
    grads = {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}
    return dict(cls=cls.detach(), patches=patches.detach(), block_cls=blocks, grads=grads)


def gen_small_chain(SpaceTimeTransformer, sim_matrix, NormSoftmaxLoss):
    """Small full chain: ref video encoder + HF DistilBERT + ref sim_matrix + ref loss."""
    from transformers import DistilBertConfig, DistilBertModel
    cfg = DistilBertConfig(vocab_size=SMALL_TEXT["vocab"], max_position_embeddings=SMALL_TEXT["max_pos"],
                           n_layers=SMALL_TEXT["n_layers"], n_heads=SMALL_HEADS, dim=SMALL_TEXT["dim"],
                           hidden_dim=SMALL_TEXT["hidden_dim"], attn_implementation="eager")
    txt = DistilBertModel(cfg).eval()
    vid = SpaceTimeTransformer(img_size=48, patch_size=16, embed_dim=128, depth=2, num_heads=SMALL_HEADS,
                               num_frames=3, time_init="rand")
    vid.head = nn.Identity()
    vid.pre_logits = nn.Identity()
    sd = si.frozen_state_dict(SEED, SMALL_VIDEO, SMALL_TEXT, proj_dim=64)
    vid.load_state_dict({k[12:]: v for k, v in sd.items() if k.startswith("video_model.")}, strict=False)
    r = txt.load_state_dict({k[11:]: v for k, v in sd.items() if k.startswith("text_model.")}, strict=False)
    assert not r.unexpected_keys and not r.missing_keys, r
    txt_proj = nn.Sequential(nn.ReLU(), nn.Linear(128, 64))
    vid_proj = nn.Sequential(nn.Linear(128, 64))
    txt_proj[1].weight.data.copy_(sd["txt_proj.1.weight"]); txt_proj[1].bias.data.copy_(sd["txt_proj.1.bias"])
    vid_proj[0].weight.data.copy_(sd["vid_proj.0.weight"]); vid_proj[0].bias.data.copy_(sd["vid_proj.0.bias"])
    vid.eval()
    video, ids, mask = small_inputs(B=4)
    mask[3, 3:] = 0
    hidden = txt(input_ids=ids, attention_mask=mask).last_hidden_state
    t = txt_proj(hidden[:, 0].float())
    cls, _ = vid(video)
    v = vid_proj(cls)
    sim = sim_matrix(t, v)
    loss = NormSoftmaxLoss()(sim)
    loss.backward()
    grads = {}
    for pre, mod in (("video_model.", vid), ("text_model.", txt), ("txt_proj.", txt_proj), ("vid_proj.", vid_proj)):
        for k, prm in mod.named_parameters():
            if prm.grad is not None:
                grads[pre + k] = prm.grad.detach().clone()
    return dict(text_hidden=hidden.detach(), text=t.detach(), video=v.detach(), sim=sim.detach(),
                loss=loss.detach(), grads=grads, mask=mask)


def gen_loss_cases(sim_matrix, NormSoftmaxLoss):
    cases = {}
    for name, n, m in (("sq8", 8, 8), ("sq1", 1, 1), ("sq33", 33, 33)):
        a = si.seeded_tensor(SEED, f"loss.a.{name}", (n, 16))
        b = si.seeded_tensor(SEED, f"loss.b.{name}", (m, 16))
        if name == "sq8":
            a[2] = 0.0               # zero row -> norm clamp path (eps)
        a.requires_grad_(True); b.requires_grad_(True)
        sim = sim_matrix(a, b)
        loss = NormSoftmaxLoss()(sim)
        loss.backward()
        cases[name] = dict(sim=sim.detach(), loss=loss.detach(), ga=a.grad.clone(), gb=b.grad.clone())
    return cases


def gen_full(FrozenInTime, sim_matrix, NormSoftmaxLoss, T, B=2, L=12):
    """The contract class oa_model.FrozenInTime at ViT-B/16 + DistilBERT-base geometry."""
    from transformers import DistilBertConfig, DistilBertModel
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            os.makedirs("pretrained")
            DistilBertModel(DistilBertConfig()).save_pretrained("pretrained/distilbert-base-uncased")
            torch.save({}, "pretrained/jx_vit_base_p16_224-80ecf9dd.pth")
            m = FrozenInTime(
                video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=T,
                                  pretrained=True, time_init="rand"),
                object_params=dict(model="", input_objects=False),
                text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text"),
                projection="minimal", load_checkpoint="")
        finally:
            os.chdir(cwd)
    sd = si.frozen_state_dict(SEED, dict(num_frames=T), {})
    r = m.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys, r.unexpected_keys
    assert not r.missing_keys, r.missing_keys
    m.eval()                          # DistilBERT dropout off (SURVEY parity traps)
    video = si.seeded_tensor(SEED, f"full.video.{T}", (B, T, 3, 224, 224))
    ids = si.seeded_ints(SEED, f"full.ids.{T}", (B, L), 1000, 30000)
    ids[:, 0] = 101
    mask = torch.ones(B, L, dtype=torch.int64)
    mask[1, L - 3:] = 0
    t, v = m({"video": video, "text": {"input_ids": ids, "attention_mask": mask}})
    sim = sim_matrix(t, v)
    loss = NormSoftmaxLoss()(sim)
    loss.backward()
    probe = {}
    for k, prm in m.named_parameters():
        if prm.grad is None:
            continue
        g = prm.grad.flatten()
        idx = si.seeded_ints(SEED, "probe." + k, (8,), 0, g.numel())
        probe[k] = dict(idx=idx, val=g[idx].clone(), norm=g.norm().clone())
    return dict(text=t.detach(), video=v.detach(), sim=sim.detach(), loss=loss.detach(), grad_probe=probe,
                T=T, B=B, L=L, mask=mask)


def _oa_inputs(B=2, F=2, L=8, Lp=12, O=3, K=5):
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
    otm = torch.tensor([[1, 3, 4], [2, 3, 5]], dtype=torch.int64)[:B, :O]      # cumulative tag-token ends
    treg = si.seeded_tensor(SEED, "oa.treg", (B, K, 512))
    return dict(video=video, ids=ids, mask=mask, pids=pids, pmask=pmask, patch_masks=patch_masks,
                region_masks=region_masks, otm=otm, treg=treg)


def _probe(module):
    probe = {}
    for k, prm in module.named_parameters():
        if prm.grad is None:
            continue
        g = prm.grad.flatten()
        idx = si.seeded_ints(SEED, "probe." + k, (8,), 0, g.numel())
        probe[k] = dict(idx=idx, val=g[idx].clone(), norm=g.norm().clone())
    return probe


def _pretrained_dir():
    from transformers import DistilBertConfig, DistilBertModel
    os.makedirs("pretrained", exist_ok=True)
    DistilBertModel(DistilBertConfig()).save_pretrained("pretrained/distilbert-base-uncased")
    torch.save({}, "pretrained/jx_vit_base_p16_224-80ecf9dd.pth")


def gen_region_mem(NormSoftmaxLoss, sim_matrix):
    """The reference's own oa_model_region_mem.FrozenInTime at ViT-B/16 geometry, F=2 (one object frame +
    one video frame), with the region loss of trainer_region_mem.py:151-167."""
    from OATrans.model.oa_model_region_mem import FrozenInTime
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            _pretrained_dir()
            m = FrozenInTime(video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=1,
                                               pretrained=True, time_init="rand"),
                             object_params=dict(model="", input_objects=False),
                             text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text"),
                             projection="minimal", load_checkpoint="")
        finally:
            os.chdir(cwd)
    sd = si.frozen_state_dict(SEED, dict(num_frames=1), {})
    sd.update(si.seeded_state_dict({"video_model.region_norm.weight": (768,), "video_model.region_norm.bias": (768,),
                                    "video_model.object_embed.weight": (768, 2054), "video_model.object_embed.bias": (768,),
                                    "txt_proj_2.1.weight": (256, 512), "txt_proj_2.1.bias": (256,)}, SEED))
    r = m.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and not r.missing_keys, r
    m.eval()
    d = _oa_inputs()
    t, v, rsim = m({"video": d["video"], "text": {"input_ids": d["ids"], "attention_mask": d["mask"]},
                    "text_region_embedding": d["treg"]})
    loss = NormSoftmaxLoss()(sim_matrix(t, v))
    rs, pm = rsim.view(-1, rsim.size(-1)), d["region_masks"].view(-1, 196)
    loss = loss + 0.1 * torch.nn.BCELoss(reduction='sum')(rs, pm) / rs.size(0)
    loss.backward()
    return dict(text=t.detach(), video=v.detach(), region_sim=rsim.detach(), loss=loss.detach(), grad_probe=_probe(m))


def gen_global_local(NormSoftmaxLoss, sim_matrix):
    """The reference's oa_model_global_local.FrozenInTime (needs cwd-style imports + an Identity shim for the
    undefined CrossModalityFusion) with the 3-term loss of trainer_global_local.py:187-211."""
    import OATrans.base as refbase
    sys.path.insert(0, f"{REF}/OATrans")
    for name in ("cv2",):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["base"] = refbase
    spec = importlib.util.spec_from_file_location("model.oa_video_transformer_global_local",
                                                  f"{REF}/OATrans/model/oa_video_transformer_global_local.py")
    vt = importlib.util.module_from_spec(spec)
    pkg = types.ModuleType("model")
    pkg.__path__ = []
    sys.modules["model"] = pkg
    sys.modules["model.oa_video_transformer_global_local"] = vt
    spec.loader.exec_module(vt)
    spec = importlib.util.spec_from_file_location("ref_oa_gl", f"{REF}/OATrans/model/oa_model_global_local.py")
    gl = importlib.util.module_from_spec(spec)
    gl.CrossModalityFusion = nn.Identity          # undefined in the reference; never used in forward
    gl.__dict__["CrossModalityFusion"] = nn.Identity
    spec.loader.exec_module(gl)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            _pretrained_dir()
            m = gl.FrozenInTime(video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=1,
                                                  pretrained=True, time_init="rand", two_outputs=False),
                                object_params=dict(model="", input_objects=False),
                                text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text"),
                                projection="minimal", load_checkpoint="")
        finally:
            os.chdir(cwd)
    sd = si.frozen_state_dict(SEED, dict(num_frames=1), {})
    sd.update(si.seeded_state_dict({"video_model.object_embed.weight": (768, 2054), "video_model.object_embed.bias": (768,),
                                    "text_local_proj.1.weight": (256, 768), "text_local_proj.1.bias": (256,),
                                    "vid_local_proj.0.weight": (256, 768), "vid_local_proj.0.bias": (256,)}, SEED))
    r = m.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and not r.missing_keys, r
    m.eval()
    m.set_device("cpu")
    d = _oa_inputs()
    t, pt, v, ov, extra = m({"video": d["video"], "text": {"input_ids": d["ids"], "attention_mask": d["mask"]},
                             "pad_text": {"input_ids": d["pids"], "attention_mask": d["pmask"]},
                             "patch_masks": d["patch_masks"], "object_token_masks": d["otm"],
                             "object_token_len": d["otm"][:, -1]})
    region_feat, tags_feat = extra[4], extra[5]
    L = NormSoftmaxLoss()
    loss = L(sim_matrix(t, v)) + L(sim_matrix(pt, v)) + L(sim_matrix(region_feat.mean(1), tags_feat.mean(1)))
    loss.backward()
    return dict(text=t.detach(), pad_text=pt.detach(), video=v.detach(), object_video=ov.detach(),
                region_feat=region_feat.detach(), tags_feat=tags_feat.detach(), loss=loss.detach(), grad_probe=_probe(m))


def main():
    import_reference()
    from OATrans.model.video_transformer import SpaceTimeTransformer
    from OATrans.model.oa_model import FrozenInTime, sim_matrix
    from OATrans.model.loss import NormSoftmaxLoss
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["small", "loss", "full", "oa", "v336"]
    if "oa" in which:
        torch.save(gen_region_mem(NormSoftmaxLoss, sim_matrix), os.path.join(HERE, "oa_region_mem.pt"))
        print("region_mem done")
        torch.save(gen_global_local(NormSoftmaxLoss, sim_matrix), os.path.join(HERE, "oa_global_local.pt"))
        print("global_local done")
    if "small" in which:
        torch.save(gen_small_video(SpaceTimeTransformer), os.path.join(HERE, "small_video.pt"))
        torch.save(gen_small_chain(SpaceTimeTransformer, sim_matrix, NormSoftmaxLoss),
                   os.path.join(HERE, "small_chain.pt"))
        print("small done")
    if "v336" in which:
        torch.save(gen_video_336(SpaceTimeTransformer), os.path.join(HERE, "video_336.pt"))
        print("video 336 done")
    if "loss" in which:
        torch.save(gen_loss_cases(sim_matrix, NormSoftmaxLoss), os.path.join(HERE, "loss_cases.pt"))
        print("loss done")
    if "full" in which or "full8" in which:
        for T in ((8,) if "full8" in which else (1, 4, 8)):
            torch.save(gen_full(FrozenInTime, sim_matrix, NormSoftmaxLoss, T), os.path.join(HERE, f"full_T{T}.pt"))
            print("full", T, "done")
    restore_build_path()


if __name__ == "__main__":
    main()
