"""TEST INFRASTRUCTURE - CPU restatement of the reference's frame transforms.  Only tests/ may import it.

The reference composes torchvision transforms on float tensors (data_loader/transforms.py:4-31;
base_dataset_global_local.py:251-257) after frames.float() / 255 (base/base_dataset.py:519-545).  It pins
torchvision==0.9.1 (environment.yml:162), a third-party dependency that is absent from /root/reference and not installed in
this image, so the pipeline cannot be RUN here; it is restated from torchvision 0.9.1's published source, step by step:

  * Resize / RandomResizedCrop on a Tensor: transforms.Resize.forward -> functional.resize -> functional_tensor.resize, which for
    'bilinear' calls torch.nn.functional.interpolate(img, size=[new_h, new_w], mode='bilinear', align_corners=False) - torch
    ITSELF, available here; 0.9.x has no antialias argument (it arrived in 0.10).  `resize` below is that call, so the
    resampling arithmetic is pinned by construction.  Resize(int): the shorter side becomes `size`, the longer one
    int(size * long / short) (functional_tensor.resize, the `isinstance(size, int)` branch).
  * CenterCrop: functional.center_crop - crop_top = int(round((h - crop_h) / 2.)), crop_left likewise.
  * RandomResizedCrop.get_params, RandomHorizontalFlip.forward, ColorJitter.forward: the parameter DRAWS, restated call for
    call on torch's global generator (tv_random_resized_crop_params / tv_train_draws below) with a seeded known-answer in
    tests/test_frames_cpu.py.  ColorJitter(0, 0, 0) changes no pixel but still draws its order (torch.randperm(4)).
  * Normalize: (x - mean) / std per channel.

What stays unpinned: video DECODING (cv2 / av / decord readers, base/base_dataset.py:465-552) - no codec exists in this image,
on either side; decoded uint8 frames are where this repository's input pipeline starts."""
import math

import torch
import torch.nn.functional as F

MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)


def to_float_chw(frames_u8):
    """base_dataset.py:519-545: uint8 [F, H, W, 3] -> float [F, 3, H, W] in [0, 1]."""
    return frames_u8.permute(0, 3, 1, 2).float() / 255


def resize(x, hw):
    return F.interpolate(x, size=hw, mode="bilinear", align_corners=False)


def normalize(x):
    return (x - MEAN) / STD


def train_clip(frames_u8, crop, flip, R):
    """RandomResizedCrop (crop box given) + RandomHorizontalFlip (decision given) + Normalize."""
    x0, y0, w, h = crop
    x = resize(to_float_chw(frames_u8)[:, :, y0:y0 + h, x0:x0 + w], (R, R))
    if flip:
        x = x.flip(-1)
    return normalize(x)


def oa_clip(frames_u8, R):
    """base_dataset_global_local.py:251-257: Resize((R, R)) + Normalize."""
    return normalize(resize(to_float_chw(frames_u8), (R, R)))


def eval_clip(frames_u8, R, center_crop=256):
    """'val' / 'test': Resize(center_crop) (shorter side) + CenterCrop + Resize(R) + Normalize."""
    x = to_float_chw(frames_u8)
    H, W = x.shape[-2:]
    h1, w1 = (int(center_crop * H / W), center_crop) if W <= H else (center_crop, int(center_crop * W / H))
    x = resize(x, (h1, w1))
    top, left = int(round((h1 - center_crop) / 2.0)), int(round((w1 - center_crop) / 2.0))
    x = x[:, :, top:top + center_crop, left:left + center_crop]
    return normalize(resize(x, (R, R)))


def tv_random_resized_crop_params(height, width, scale=(0.5, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision 0.9.1 transforms/transforms.py: RandomResizedCrop.get_params, call for call on torch's global generator
    -> (i, j, h, w) = (top, left, height, width) of the crop box."""
    area = height * width
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        log_ratio = torch.log(torch.tensor(ratio))
        aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,)).item()
            j = torch.randint(0, width - w + 1, size=(1,)).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)          # fallback to a central crop
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def tv_train_draws(height, width, scale=(0.5, 1.0)):
    """The random draws of the 'train' Compose (data_loader/transforms.py:11-16) in torchvision 0.9.1's order:
    RandomResizedCrop.get_params, RandomHorizontalFlip.forward (`torch.rand(1) < p`), ColorJitter.forward (`torch.randperm(4)`,
    drawn even when every factor is None).  -> ((i, j, h, w), flip)"""
    box = tv_random_resized_crop_params(height, width, scale)
    flip = bool(torch.rand(1) < 0.5)
    torch.randperm(4)
    return box, flip
