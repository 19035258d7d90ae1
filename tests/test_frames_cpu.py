"""Input pipeline, host side: the random parameters of the reference's 'train' transform and the geometry of
its 'val' / 'test' transform (/root/reference/OATrans/data_loader/transforms.py:4-31, torchvision==0.9.1 per environment.yml:162).

torchvision is absent from this image, so the draws are pinned the only way they can be: oracle/frames_oracle.py restates
RandomResizedCrop.get_params / RandomHorizontalFlip / ColorJitter of torchvision 0.9.1 call for call on torch's global
generator, the known answers below were produced by that restatement under fixed seeds, and the product's
`torchvision_train_draws` must reproduce the oracle draw for draw.  CPU only."""
import torch

from OATrans.data_loader import frames as fr
from oracle import frames_oracle as fo


def test_train_draws_known_answers():
    torch.manual_seed(1234)
    got = [fo.tv_train_draws(360, 640) for _ in range(4)]
    # (top, left, height, width), flip - samples 3 and 4 exhaust the ten tries (a 16:9 frame admits few boxes of ratio 3/4..4/3 at
    # half its area or more) and take the central fallback: h = H, w = round(H * 4/3)
    assert got == [((0, 65, 354, 335), True), ((2, 51, 353, 404), True), ((0, 80, 360, 480), True), ((0, 80, 360, 480), True)]
    torch.manual_seed(7)
    assert [fo.tv_train_draws(224, 160, (0.9, 1.0)) for _ in range(3)] == [((5, 0, 213, 160), False)] * 3      # portrait frame: w = W, h = round(W / (3/4))
    torch.manual_seed(99)
    assert fo.tv_random_resized_crop_params(10, 400, (0.99, 1.0)) == (0, 193, 10, 13)


def test_product_draws_equal_the_restatement_draw_for_draw():
    for seed, (H, W), scale in [(0, (360, 640), (0.5, 1.0)), (1, (240, 320), (0.5, 1.0)), (2, (224, 224), (0.08, 1.0)), (3, (500, 100), (0.5, 1.0))]:
        torch.manual_seed(seed)
        ref = [fo.tv_train_draws(H, W, scale) for _ in range(25)]
        after_ref = torch.rand(1).item()
        torch.manual_seed(seed)
        mine = [fr.torchvision_train_draws(H, W, scale) for _ in range(25)]
        after_mine = torch.rand(1).item()
        for ((i, j, h, w), flip), ((x0, y0, w2, h2), flip2) in zip(ref, mine):
            assert (x0, y0, w2, h2, flip2) == (j, i, w, h, flip)
        assert after_ref == after_mine                  # the same NUMBER of draws: the generator is left where torchvision would leave it


def test_eval_geometry_follows_torchvision_resize_and_center_crop():
    # Resize(int): shorter side -> size, longer side int(size * long / short) (functional_tensor.resize); CenterCrop: int(round((h - c) / 2.))
    assert fr.resize_shorter_side(360, 640, 256) == (256, int(256 * 640 / 360))
    assert fr.resize_shorter_side(640, 360, 256) == (int(256 * 640 / 360), 256)
    assert fr.resize_shorter_side(300, 300, 256) == (256, 256)
    x = torch.arange(2 * 3 * 7 * 9, dtype=torch.float32).reshape(2, 3, 7, 9)
    # the oracle's resize IS the call torchvision 0.9.1 makes on tensors
    assert torch.equal(fo.resize(x, (5, 4)), torch.nn.functional.interpolate(x, size=[5, 4], mode="bilinear", align_corners=False))
