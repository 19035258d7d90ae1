"""Counter-based, wheel-independent weight generator.

Parity tests must build *identical* weights on both sides of a comparison
(reference / oracle in the build container, HIP engine on the GPU box) without
shipping 700 MB of tensors and without depending on ``torch.manual_seed``
streams.  Every tensor element is ``f(seed, fnv1a(name), flat_index)`` with
``f`` = splitmix64 -> uniform(-1, 1) * sqrt(3) * std, so a tensor is
reproducible from its *name* alone, in any order, on any machine.

This is build-owned code (no reference counterpart): the reference initialises
from ImageNet / HF checkpoints that do not exist offline
(/root/reference/OATrans/model/oa_model.py:27,42).
"""
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for c in name.encode():
        h ^= c
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def unit_uniform(seed: int, name: str, n: int) -> np.ndarray:
    """n float64 values in [-1, 1), a pure function of (seed, name, index)."""
    base = np.uint64((_fnv1a(name) ^ (seed * 0x2545F4914F6CDD1D)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base
    z = _splitmix64(idx)
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u * 2.0 - 1.0


def seeded_tensor(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    """fp32 tensor with the given mean/std (uniform distribution)."""
    n = int(np.prod(shape)) if len(shape) else 1
    v = unit_uniform(seed, name, n) * (np.sqrt(3.0) * std) + mean
    return torch.from_numpy(v.astype(np.float32)).reshape(tuple(shape))


def seeded_ints(seed: int, name: str, shape, lo: int, hi: int) -> torch.Tensor:
    """int64 tensor uniform in [lo, hi)."""
    n = int(np.prod(shape))
    u = (unit_uniform(seed, name, n) + 1.0) * 0.5
    v = np.minimum((u * (hi - lo)).astype(np.int64) + lo, hi - 1)
    return torch.from_numpy(v).reshape(tuple(shape))


def video_param_shapes(embed_dim=768, depth=12, mlp_ratio=4, num_frames=8, patches_per_frame=196,
                       patch=16, in_chans=3):
    """state_dict key -> shape for SpaceTimeTransformer
    (/root/reference/OATrans/model/video_transformer.py:195-262)."""
    D, Hd = embed_dim, int(embed_dim * mlp_ratio)
    s = {
        "cls_token": (1, 1, D),
        "pos_embed": (1, patches_per_frame + 1, D),
        "temporal_embed": (1, num_frames, D),
        "patch_embed.proj.weight": (D, in_chans, patch, patch),
        "patch_embed.proj.bias": (D,),
        "norm.weight": (D,),
        "norm.bias": (D,),
    }
    for i in range(depth):
        b = f"blocks.{i}."
        for ln in ("norm1", "norm2", "norm3"):
            s[b + ln + ".weight"] = (D,)
            s[b + ln + ".bias"] = (D,)
        for at in ("attn", "timeattn"):
            s[b + at + ".qkv.weight"] = (3 * D, D)
            s[b + at + ".qkv.bias"] = (3 * D,)
            s[b + at + ".proj.weight"] = (D, D)
            s[b + at + ".proj.bias"] = (D,)
        s[b + "mlp.fc1.weight"] = (Hd, D)
        s[b + "mlp.fc1.bias"] = (Hd,)
        s[b + "mlp.fc2.weight"] = (D, Hd)
        s[b + "mlp.fc2.bias"] = (D,)
    return s


def text_param_shapes(dim=768, n_layers=6, hidden_dim=3072, vocab=30522, max_pos=512):
    """state_dict key -> shape for HF DistilBertModel (transformers, third party;
    call site /root/reference/OATrans/model/oa_model.py:27,113)."""
    s = {
        "embeddings.word_embeddings.weight": (vocab, dim),
        "embeddings.position_embeddings.weight": (max_pos, dim),
        "embeddings.LayerNorm.weight": (dim,),
        "embeddings.LayerNorm.bias": (dim,),
    }
    for i in range(n_layers):
        b = f"transformer.layer.{i}."
        for lin in ("q_lin", "k_lin", "v_lin", "out_lin"):
            s[b + f"attention.{lin}.weight"] = (dim, dim)
            s[b + f"attention.{lin}.bias"] = (dim,)
        s[b + "sa_layer_norm.weight"] = (dim,)
        s[b + "sa_layer_norm.bias"] = (dim,)
        s[b + "ffn.lin1.weight"] = (hidden_dim, dim)
        s[b + "ffn.lin1.bias"] = (hidden_dim,)
        s[b + "ffn.lin2.weight"] = (dim, hidden_dim)
        s[b + "ffn.lin2.bias"] = (dim,)
        s[b + "output_layer_norm.weight"] = (dim,)
        s[b + "output_layer_norm.bias"] = (dim,)
    return s


def _std_for(name: str, shape) -> tuple:
    """(mean, std) by parameter role: LN gains around 1, biases small, matrices
    fan-in scaled so activations stay O(1) through 12 blocks.  time-attention
    weights are non-zero on purpose (SURVEY.md 'parity traps':
    time_init='zeros' would never exercise the temporal kernel)."""
    leaf = name.split(".")[-1]
    if ("norm" in name.lower() or "LayerNorm" in name) and leaf == "weight":
        return 1.0, 0.1
    if leaf == "bias":
        return 0.0, 0.05
    if "embed" in name or "cls_token" in name:
        return 0.0, 0.2 if "word" not in name and "position" not in name else 0.5
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return 0.0, 1.0 / np.sqrt(fan_in)
    return 0.0, 0.02


def seeded_state_dict(shapes: dict, seed: int, prefix: str = "") -> dict:
    out = {}
    for k, shp in shapes.items():
        mean, std = _std_for(k, shp)
        out[prefix + k] = seeded_tensor(seed, prefix + k, shp, std=std, mean=mean)
    return out


def frozen_state_dict(seed, video_kw=None, text_kw=None, proj_dim=256):
    """Full FrozenInTime state dict with the reference's key names
    (SURVEY.md 5.4): video_model.*, text_model.*, txt_proj.1.*, vid_proj.0.*"""
    video_kw = video_kw or {}
    text_kw = text_kw or {}
    sd = {}
    sd.update(seeded_state_dict(video_param_shapes(**video_kw), seed, "video_model."))
    sd.update(seeded_state_dict(text_param_shapes(**text_kw), seed, "text_model."))
    D = video_kw.get("embed_dim", 768)
    Dt = text_kw.get("dim", 768)
    sd.update(seeded_state_dict({
        "txt_proj.1.weight": (proj_dim, Dt), "txt_proj.1.bias": (proj_dim,),
        "vid_proj.0.weight": (proj_dim, D), "vid_proj.0.bias": (proj_dim,),
    }, seed))
    return sd
