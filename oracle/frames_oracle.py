"""TEST INFRASTRUCTURE - CPU restatement of the reference's frame transforms.  Only tests/ may import it.

The reference composes torchvision transforms on float tensors (data_loader/transforms.py:4-31;
base_dataset_global_local.py:251-257) after frames.float() / 255 (base/base_dataset.py:519-545).  torchvision is not
installed in this image, so the pipeline cannot be RUN here: **parity unpinned** for the composition.  Each step is
restated from torchvision's tensor implementation, whose resampling IS torch.nn.functional.interpolate(mode='bilinear',
align_corners=False) - torch itself, available here - so the arithmetic below is the arithmetic torchvision executes."""
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
