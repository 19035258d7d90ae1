"""Object-aware input producers on CPU: the oracle restatement AND the product's host
code (OATrans/data_loader/object_inputs.py) against outputs of the reference's own functions
(tests/golden/oa_inputs.pt + tests/golden/oa_inputs/*.npz, made by tests/golden/make_golden_inputs.py), plus
product-vs-oracle on random cases.  Integer / string results: exact; box features: bit-exact (same float32 divisions)."""
import os
import random

import numpy as np
import pytest
import torch

from OATrans.data_loader import object_inputs as prod
from oracle import oa_inputs_oracle as orc

IMPLS = [pytest.param(orc, id="oracle"), pytest.param(prod, id="product")]


@pytest.fixture(scope="module")
def g(golden_dir):
    return torch.load(os.path.join(golden_dir, "oa_inputs.pt"), map_location="cpu", weights_only=False)


@pytest.mark.parametrize("impl", IMPLS)
def test_sample_frames_vs_reference(g, impl):
    for c in g["sample_frames"]:
        rng = random.Random(c["seed"])          # random.seed(s) + module-level calls == Random(s) instance calls
        got = impl.sample_frames(c["num_frames"], c["vlen"], sample=c["sample"], fix_start=c["fix_start"], rng=rng)
        assert [int(x) for x in got] == c["idxs"], c


@pytest.mark.parametrize("impl", IMPLS)
def test_object_frame_selection_vs_reference(g, impl):
    for c in g["object_frame"]:
        for key, mode in (("global_local", "part"), ("region_part", "part"), ("region_full", "full")):
            frames, obj = impl.select_object_frame(list(c["frame_idxs"]), c["vlen"], mode)
            assert ([int(x) for x in frames], int(obj)) == (c[key][0], c[key][1]), (c, key)
            assert frames[1:] == c["frame_idxs"]                       # object frame first, clip untouched


@pytest.mark.parametrize("impl", IMPLS)
def test_detector_npz_vs_reference(g, golden_dir, impl):
    classes = impl.parse_vocab(g["vocab_lines"])
    assert classes[0] == "__background__" and len(classes) == 61 and all(c == c.lower().strip() for c in classes)
    for c in g["npz"]:
        frame = np.load(os.path.join(golden_dir, c["file"]), allow_pickle=True)
        tags, ids, feats = impl.read_bboxs_tags(frame, classes, top_k=c["top_k"], v=c["v"])
        assert tags == c["tags"], c["file"]
        assert torch.equal(torch.as_tensor(np.asarray(ids).astype(np.int64)), c["ids"]), c["file"]
        assert feats.dtype == c["feats"].dtype and torch.equal(feats, c["feats"]), c["file"]
    # the product also takes the path itself
    c = g["npz"][0]
    tags, _, _ = prod.read_bboxs_tags(os.path.join(golden_dir, c["file"]), classes, top_k=c["top_k"], v=c["v"])
    assert tags == c["tags"]


@pytest.mark.parametrize("impl", IMPLS)
def test_tag_token_masks_and_region_embeddings_vs_reference(g, impl):
    lens = g["token_lens"].numpy()
    for c in g["tag_masks"]:
        ends, total = impl.object_tags_masks(c["ids"], lens)
        assert torch.equal(ends.float(), c["ends"]) and int(total) == c["total"]
    r = g["region_embeddings"]
    assert torch.equal(impl.region_embeddings(r["memory"], r["labels"]), r["out"])


@pytest.mark.parametrize("impl", IMPLS)
def test_pack_clip(impl):
    imgs = torch.randn(7, 3, 8, 8)                       # object frame + 6 of 8 clip frames decoded
    out = impl.pack_clip(imgs, 8, 8)
    assert out.shape == (9, 3, 8, 8) and torch.equal(out[:7], imgs) and not out[7:].any()
    full = impl.pack_clip(torch.randn(9, 3, 8, 8), 8, 8)
    assert full.shape == (9, 3, 8, 8)


def test_product_matches_oracle_on_random_detector_outputs():
    rs = np.random.RandomState(3)
    classes = ["__background__"] + [f"c{i}" for i in range(200)]
    for trial in range(60):
        n = int(rs.randint(1, 40))
        top_k = int(rs.choice([1, 5, 10, 15, 20]))
        w, h = int(rs.randint(100, 2000)), int(rs.randint(100, 2000))
        xy0 = rs.uniform(0, 0.8, size=(n, 2)) * [w, h]
        bbox = np.concatenate([xy0, xy0 + rs.uniform(1, 50, size=(n, 2))], 1).astype(np.float32)
        frame = {"x": np.zeros((n, 4), np.float32), "bbox": bbox,
                 "info": {"objects_conf": rs.permutation(n).astype(np.float32) / n,       # distinct confidences: argsort ties are unspecified
                          "objects_id": rs.randint(0, int(rs.choice([3, 199])), size=n).astype(np.int64), "image_w": w, "image_h": h}}
        for v in (1, 2):
            a = orc.read_bboxs_tags(frame, classes, top_k=top_k, v=v)
            b = prod.read_bboxs_tags(frame, classes, top_k=top_k, v=v)
            assert a[0] == b[0] and np.array_equal(np.asarray(a[1]), np.asarray(b[1])) and torch.equal(a[2], b[2])
    for trial in range(200):
        T, vlen = int(rs.randint(1, 17)), int(rs.randint(40, 3000))
        seed = int(rs.randint(1 << 30))
        fa = orc.sample_frames(T, vlen, "rand", rng=random.Random(seed))
        fb = prod.sample_frames(T, vlen, "rand", rng=random.Random(seed))
        assert fa == fb
        for mode in ("part", "full"):
            assert orc.select_object_frame(fa, vlen, mode) == prod.select_object_frame(fb, vlen, mode)
    # the region-memory class draw consumes the generator like the reference's random.sample
    ids = list(rs.randint(0, 6, size=15))
    sel = prod.select_region_classes(ids, 5, rng=random.Random(9))
    bb = np.concatenate([rs.uniform(0, 0.5, (15, 2)), rs.uniform(0.5, 1.0, (15, 2))], 1)
    _, sel_o = orc.patch_all_masks_region(bb, ids, 5, rng=random.Random(9))
    assert sel == [int(s) for s in sel_o]
