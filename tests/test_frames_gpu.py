"""Device-side input pipeline: oat_frames_resize / data_loader/frames.py against the CPU restatement
of the reference's transform pipelines (oracle/frames_oracle.py: torch's own bilinear interpolation - what torchvision runs
on tensors; the composition itself is unpinned, torchvision being absent here).  Tolerance: fp32 output <= 2e-5 abs
(summation order of the four taps), bf16 output = the rounding of that."""
import random

import pytest
import torch

from oracle import frames_oracle as forc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw", [(360, 640), (480, 360), (224, 224), (97, 131)])
def test_frame_transforms_vs_oracle(hw):
    from OATrans.data_loader import frames as fr
    from OATrans.ops import hip
    H, W = hw
    g = torch.Generator().manual_seed(H * 1000 + W)
    frames = torch.randint(0, 256, (5, H, W, 3), generator=g, dtype=torch.uint8)
    dev = frames.cuda()
    # OA datasets: Resize((224, 224)) + Normalize
    got = fr.clip_from_frames(dev, "oa", 224, dtype=torch.float32).cpu()
    want = forc.oa_clip(frames, 224)
    assert (got - want).abs().max().item() < 2e-5
    got16 = fr.clip_from_frames(dev, "oa", 224).cpu()
    assert got16.dtype == torch.bfloat16 and (got16.float() - want.bfloat16().float()).abs().max().item() <= 2 ** -6
    # train: the crop box and the flip decision are drawn by the host (Python's generator), then one launch
    rng = random.Random(7)
    crop = fr.random_resized_crop_params(H, W, (0.5, 1.0), rng=rng)
    x0, y0, w, h = crop
    assert 0 <= x0 and x0 + w <= W and 0 <= y0 and y0 + h <= H and 0.5 * H * W * 0.9 <= w * h <= H * W
    for flip in (False, True):
        got = hip.frames_resize(dev, (224, 224), crop=crop, flip=flip, dtype=torch.float32).cpu()
        assert (got - forc.train_clip(frames, crop, flip, 224)).abs().max().item() < 2e-5
    # val / test: Resize(256) + CenterCrop(256) + Resize(224) + Normalize (two launches)
    if min(H, W) >= 64:
        got = fr.clip_from_frames(dev, "val", 224, dtype=torch.float32).cpu()
        assert (got - forc.eval_clip(frames, 224)).abs().max().item() < 5e-5
    # seeded reproducibility of the train path
    a = fr.clip_from_frames(dev, "train", 224, rng=random.Random(3))
    b = fr.clip_from_frames(dev, "train", 224, rng=random.Random(3))
    assert torch.equal(a, b)


def test_frames_resize_rejects_bad_crop_and_packs_into_batch_buffer():
    from OATrans.ops import hip
    frames = torch.randint(0, 256, (3, 40, 50, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(hip.OatError):
        hip.frames_resize(frames, (16, 16), crop=(30, 0, 40, 20))
    batch = torch.zeros(2, 4, 3, 16, 16, dtype=torch.bfloat16, device="cuda")      # [B, T + 1, 3, R, R]
    hip.frames_resize(frames, (16, 16), out=batch[1, :3])                          # straight into the sample's slot
    assert batch[1, :3].float().abs().sum() > 0 and not batch[0].any() and not batch[1, 3].any()


def test_token_cache_pads_like_the_tokenizer_and_tokenises_once():
    from OATrans.data_loader.frames import TokenCache
    calls = []

    def tok(texts, truncation=True, max_length=None):
        calls.append(list(texts))
        ids = [[101] + [1000 + (hash(w) % 5000) for w in t.split()][: (max_length or 99) - 2] + [102] for t in texts]
        return {"input_ids": ids, "attention_mask": [[1] * len(i) for i in ids]}
    cache = TokenCache(tok, max_length=8)
    b1 = cache(["a b c", "a", "a b c"], device="cuda")
    assert b1["input_ids"].shape == (3, 5) and b1["attention_mask"].tolist() == [[1] * 5, [1, 1, 1, 0, 0], [1] * 5]
    assert torch.equal(b1["input_ids"][0], b1["input_ids"][2]) and b1["input_ids"][1, 3:].tolist() == [0, 0]
    b2 = cache(["a", "z y x w v u t s r"])
    assert calls == [["a b c", "a"], ["z y x w v u t s r"]] and cache.hits == 2 and cache.misses == 3
    assert b2["input_ids"].shape == (2, 8)                                 # truncated to max_length
