"""Input pipeline, device side: decoded frames -> the clip tensor, and a token cache.

What the reference does per sample on CPU workers (base/base_dataset.py:519-545: frames.float() / 255, permute; then the
torchvision transforms of data_loader/transforms.py:4-31, or Resize((224, 224)) + Normalize in the OA datasets,
base_dataset_global_local.py:251-257) runs here as one or two launches of oat_frames_resize per BATCH on uint8 frames that
were copied to the GPU as they left the decoder (a quarter of the bytes of the float clip over PCIe).

  train   RandomResizedCrop(input_res, scale) + RandomHorizontalFlip + Normalize   -> 1 launch
  val / test   Resize(center_crop) + CenterCrop(center_crop) + Resize(input_res) + Normalize   -> 2 launches
  oa      Resize((input_res, input_res)) + Normalize   -> 1 launch

Decoding itself (cv2 / decord / av) stays with the caller: there is no codec on the device side of this repository.
The random parameters follow torchvision's algorithms (RandomResizedCrop.get_params).  rng="torch" draws them from torch's
global generator in torchvision 0.9.1's own call order (crop area, aspect ratio, top, left, flip, ColorJitter's order draw), so
that under the same torch.manual_seed a sample gets the crop box and flip the reference's Compose would give it
(torchvision_train_draws; known-answer test in tests/test_frames_cpu.py); the default draws from Python's `random`.

TokenCache: the reference tokenises the captions of every batch inside the training step (trainer_dist.py:151-153);
captions repeat every epoch, so their ids are cached per string and only padded / stacked per batch."""
import math
import random

import torch

from ..ops import hip


def random_resized_crop_params(H, W, scale=(0.5, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), rng=random):
    """torchvision RandomResizedCrop.get_params: (x0, y0, w, h) of the crop box."""
    area = H * W
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target = area * rng.uniform(scale[0], scale[1])
        ar = math.exp(rng.uniform(log_ratio[0], log_ratio[1]))
        w, h = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
        if 0 < w <= W and 0 < h <= H:
            return rng.randint(0, W - w), rng.randint(0, H - h), w, h
    in_ratio = W / H                                   # fallback: central crop at the nearest allowed ratio
    if in_ratio < ratio[0]:
        w, h = W, int(round(W / ratio[0]))
    elif in_ratio > ratio[1]:
        h, w = H, int(round(H * ratio[1]))
    else:
        w, h = W, H
    return (W - w) // 2, (H - h) // 2, w, h


def torchvision_train_draws(H, W, scale=(0.5, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """The draws of the reference's 'train' Compose (data_loader/transforms.py:11-16) on torch's global generator, in the order
    torchvision 0.9.1 (environment.yml:162) makes them: RandomResizedCrop.get_params (area ~ U(scale), log-aspect ~ U(log ratio) in
    float32, top then left by torch.randint; ten tries, then the central fallback), RandomHorizontalFlip (torch.rand(1) < 0.5),
    ColorJitter (its torch.randperm(4), drawn although all factors are zero).  -> ((x0, y0, w, h), flip)"""
    area, box = H * W, None
    lo, hi = torch.log(torch.tensor(ratio)).tolist()
    for _ in range(10):
        target = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        ar = torch.exp(torch.empty(1).uniform_(lo, hi)).item()
        w, h = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
        if 0 < w <= W and 0 < h <= H:
            y0 = torch.randint(0, H - h + 1, size=(1,)).item()
            x0 = torch.randint(0, W - w + 1, size=(1,)).item()
            box = (x0, y0, w, h)
            break
    if box is None:
        in_ratio = W / H
        if in_ratio < ratio[0]:
            w, h = W, int(round(W / ratio[0]))
        elif in_ratio > ratio[1]:
            h, w = H, int(round(H * ratio[1]))
        else:
            w, h = W, H
        box = ((W - w) // 2, (H - h) // 2, w, h)
    flip = bool(torch.rand(1) < 0.5)
    torch.randperm(4)
    return box, flip


def resize_shorter_side(H, W, size):
    """torchvision Resize(int): the shorter side becomes `size`, the other int(size * long / short)."""
    if W <= H:
        return int(size * H / W), size
    return size, int(size * W / H)


def clip_from_frames(frames, split="train", input_res=224, center_crop=256, randcrop_scale=(0.5, 1.0), dtype=torch.bfloat16,
                     rng=random, out=None):
    """frames: uint8 [F, H, W, 3] on the GPU (RGB, as the decoders deliver them) -> normalised clip [F, 3, R, R].
    split: 'train' | 'val' | 'test' (data_loader/transforms.py) | 'oa' (the OA datasets' Resize((R, R)) + Normalize)."""
    F, H, W, _ = frames.shape
    R = input_res
    if split == "train":
        if rng == "torch":
            crop, flip = torchvision_train_draws(H, W, randcrop_scale)
        else:
            crop = random_resized_crop_params(H, W, randcrop_scale, rng=rng)
            flip = rng.random() < 0.5
        return hip.frames_resize(frames, (R, R), crop=crop, flip=flip, out=out, dtype=dtype)
    if split == "oa":
        return hip.frames_resize(frames, (R, R), out=out, dtype=dtype)
    if split not in ("val", "test"):
        raise ValueError(split)
    h1, w1 = resize_shorter_side(H, W, center_crop)
    mid = hip.frames_resize(frames, (h1, w1), mean=None, std=None, dtype=torch.float32)       # [F, 3, h1, w1] in [0, 1]
    top, left = int(round((h1 - center_crop) / 2.0)), int(round((w1 - center_crop) / 2.0))
    return hip.frames_resize(mid, (R, R), crop=(left, top, center_crop, center_crop), out=out, dtype=dtype)


class TokenCache:
    """caption -> token ids, tokenised once.  `tokenizer(list_of_str, ...)` is any HF-style callable returning
    {'input_ids': [...], 'attention_mask': [...]} per caption (no padding requested); batches are padded to their
    longest caption like tokenizer(..., padding=True) does (trainer_dist.py:151-153)."""

    def __init__(self, tokenizer, max_length=None, pad_id=0):
        self.tokenizer, self.max_length, self.pad_id = tokenizer, max_length, pad_id
        self.ids = {}
        self.hits = self.misses = 0

    def __call__(self, captions, device=None):
        new = [c for c in dict.fromkeys(captions) if c not in self.ids]
        if new:
            enc = self.tokenizer(new, truncation=True, max_length=self.max_length) if self.max_length else self.tokenizer(new, truncation=True)
            for c, ids in zip(new, enc["input_ids"]):
                self.ids[c] = torch.as_tensor(ids, dtype=torch.int64)
        self.misses += len(new)
        self.hits += len(captions) - len(new)
        L = max(self.ids[c].numel() for c in captions)
        ids = torch.full((len(captions), L), self.pad_id, dtype=torch.int64)
        mask = torch.zeros(len(captions), L, dtype=torch.int64)
        for i, c in enumerate(captions):
            t = self.ids[c]
            ids[i, :t.numel()] = t
            mask[i, :t.numel()] = 1
        if device is not None:
            ids, mask = ids.to(device, non_blocking=True), mask.to(device, non_blocking=True)
        return {"input_ids": ids, "attention_mask": mask}
