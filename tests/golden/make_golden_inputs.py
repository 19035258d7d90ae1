"""Golden vectors for the object-aware INPUT producers: frame sampling, object-frame
selection, the object detector's .npz wire format (x / bbox / info) -> tags, class ids, 6-d box features, and the
tag-token masks.  Produced by running the REFERENCE's own code: whole functions / methods are lifted with `ast` from
base_dataset_global_local.py and base_dataset_region_mem.py (the modules themselves import cv2 / decord / torchvision,
absent here); the object-frame selection lives inside read_frames_cv2 between the sample_frames() call and the
frame_idxs.insert() - exactly those statements are lifted and run, nothing of cv2 is touched or faked.

    python tests/golden/make_golden_inputs.py      # needs /root/reference; writes tests/golden/oa_inputs.pt and
                                                   # the synthetic detector files tests/golden/oa_inputs/*.npz

Inputs are synthetic (seeded); only numeric / string inputs and outputs are stored."""
import ast
import math
import os
import random

import numpy as np
import torch

REF = "/root/reference/OATrans/base"
HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 20240921


def _tree(path):
    with open(path) as fh:
        return ast.parse(fh.read())


def lift(path, name, ns=None):
    """Compile one function / method of the reference file as a plain function (executes the reference's own source)."""
    for node in ast.walk(_tree(path)):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            env = {"np": np, "math": math, "random": random, "torch": torch}
            env.update(ns or {})
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), env)
            return env[name]
    raise KeyError(name)


def lift_object_frame_selection(path, sample_frames):
    """The statements of read_frames_cv2 after `frame_idxs = sample_frames(...)` up to and including
    `frame_idxs.insert(0, average_object_index)`, wrapped as f(frame_idxs, vlen, object_num) -> (frame_idxs, object_index)."""
    for node in ast.walk(_tree(path)):
        if isinstance(node, ast.FunctionDef) and node.name == "read_frames_cv2":
            body, start, stop = node.body, None, None
            for i, st in enumerate(body):
                src = ast.unparse(st)
                if start is None and src.startswith("frame_idxs = sample_frames("):
                    start = i + 1
                if src.startswith("frame_idxs.insert(0, average_object_index)"):
                    stop = i + 1
            stmts = body[start:stop]
            fn = ast.parse("def select(frame_idxs, vlen, object_num='part'):\n    pass\n    return frame_idxs, object_index").body[0]
            fn.body = stmts + [fn.body[-1]]
            mod = ast.fix_missing_locations(ast.Module(body=[fn], type_ignores=[]))
            env = {"sample_frames": sample_frames, "random": random, "np": np}
            exec(compile(mod, path, "exec"), env)
            return env["select"]
    raise KeyError("read_frames_cv2")


class _Self:
    pass


def synthetic_vocab(n=60):
    """Lines in the format of utils/objects_vocab.txt ('name,alias,...'); parsed with the reference's own expression."""
    rng = random.Random(SEED)
    syll = ["ka", "lo", "mi", "ra", "te", "su", "no", "vi", "da", "pe"]
    lines = []
    for i in range(n):
        name = "".join(rng.choice(syll) for _ in range(rng.randint(1, 3))) + str(i)
        lines.append(name.upper() + " ,alias%d\n" % i if i % 3 == 0 else name + "\n")
    return lines


def write_npz(path, n, rng, w=640, h=360, dup_classes=False, n_classes=60):
    xy0 = rng.uniform([0, 0], [w * 0.8, h * 0.8], size=(n, 2))
    wh = rng.uniform([8, 8], [w * 0.5, h * 0.5], size=(n, 2))
    bbox = np.concatenate([xy0, np.minimum(xy0 + wh, [w, h])], 1).astype(np.float32)
    conf = rng.uniform(0.05, 0.99, size=n).astype(np.float32)
    ids = rng.randint(0, 6 if dup_classes else n_classes - 1, size=n).astype(np.int64)
    info = {"objects_conf": conf, "objects_id": ids, "image_w": w, "image_h": h, "num_boxes": n}
    np.savez(path, x=rng.randn(n, 8).astype(np.float32), bbox=bbox, info=np.array(info, dtype=object))


def main():
    gl_py, rm_py = os.path.join(REF, "base_dataset_global_local.py"), os.path.join(REF, "base_dataset_region_mem.py")
    out = {}
    # ---- frame sampling (module-level sample_frames, identical text in both files)
    sample_frames = lift(gl_py, "sample_frames")
    cases = []
    for k, (T, vlen, mode, fix) in enumerate([(1, 30, "rand", None), (4, 97, "rand", None), (8, 300, "rand", None), (8, 17, "rand", None),
                                              (4, 64, "uniform", None), (8, 123, "uniform", None), (16, 500, "uniform", None),
                                              (4, 80, "uniform", 3), (8, 6, "uniform", None)]):
        random.seed(SEED + k)
        cases.append(dict(num_frames=T, vlen=vlen, sample=mode, fix_start=fix, seed=SEED + k,
                          idxs=list(map(int, sample_frames(T, vlen, sample=mode, fix_start=fix)))))
    out["sample_frames"] = cases
    # ---- object-frame selection
    sel_gl = lift_object_frame_selection(gl_py, sample_frames)
    sel_rm = lift_object_frame_selection(rm_py, sample_frames)
    sel = []
    for k, (T, vlen) in enumerate([(1, 40), (4, 97), (8, 300), (8, 33), (2, 1000), (16, 480)]):
        random.seed(SEED + 100 + k)
        idxs = list(map(int, sample_frames(T, vlen, sample="rand")))
        f_gl, o_gl = sel_gl(list(idxs), vlen)
        f_p, o_p = sel_rm(list(idxs), vlen, "part")
        f_f, o_f = sel_rm(list(idxs), vlen, "full")
        sel.append(dict(frame_idxs=idxs, vlen=vlen, global_local=(list(map(int, f_gl)), int(o_gl)),
                        region_part=(list(map(int, f_p)), int(o_p)), region_full=(list(map(int, f_f)), int(o_f))))
    out["object_frame"] = sel
    # ---- detector .npz -> tags / ids / box features
    vocab_lines = synthetic_vocab()
    classes = ['__background__']
    for object in vocab_lines:                                  # the reference's parsing line (base_dataset_global_local.py:283-285)
        classes.append(object.split(',')[0].lower().strip())
    me = _Self()
    me.classes = classes
    rng = np.random.RandomState(SEED)
    os.makedirs(os.path.join(HERE, "oa_inputs"), exist_ok=True)
    read_gl = lift(gl_py, "read_bboxs_tags_from_disk")
    read_rm = lift(rm_py, "read_bboxs_tags_from_disk")
    npz = []
    for k, (n, top_k, v, dup) in enumerate([(20, 20, 1, False), (24, 20, 1, False), (12, 20, 1, False), (30, 10, 2, False),
                                            (30, 10, 2, True), (20, 15, 1, True), (7, 15, 1, False)]):
        rel = f"oa_inputs/det_{k}.npz"
        write_npz(os.path.join(HERE, rel), n, rng, dup_classes=dup)
        tags, ids, feats = read_gl(me, os.path.join(HERE, rel), top_k=top_k, v=v)
        npz.append(dict(file=rel, top_k=top_k, v=v, tags=tags, ids=torch.as_tensor(np.asarray(ids).astype(np.int64)), feats=feats.clone()))
    # region-memory variant addresses '<dir>/<index>.npz'
    os.makedirs(os.path.join(HERE, "oa_inputs", "clip0"), exist_ok=True)
    for idx in (0, 3):
        write_npz(os.path.join(HERE, "oa_inputs", "clip0", f"{idx}.npz"), 18, rng, dup_classes=True)
        tags, ids, feats = read_rm(me, os.path.join(HERE, "oa_inputs", "clip0"), index=idx, top_k=15, v=1)
        npz.append(dict(file=f"oa_inputs/clip0/{idx}.npz", dir="oa_inputs/clip0", index=idx, top_k=15, v=1, tags=tags,
                        ids=torch.as_tensor(np.asarray(ids).astype(np.int64)), feats=feats.clone()))
    out["npz"] = npz
    out["vocab_lines"] = vocab_lines
    # ---- tag-token masks
    tok = lift(gl_py, "object_tags_masks")
    me.object_token_lens = rng.randint(1, 4, size=len(classes)).astype(np.float64)      # np.loadtxt gives float64
    tm = []
    for ids in ([3, 1, 4, 1, 5], [0], list(range(20)), [7, 7, 7]):
        mask, total = tok(me, ids)
        tm.append(dict(ids=ids, ends=mask.clone(), total=int(total)))
    out["tag_masks"] = tm
    out["token_lens"] = torch.from_numpy(me.object_token_lens.copy())
    # ---- region-memory text embeddings
    emb = lift(rm_py, "get_region_embeddings")
    mem = torch.from_numpy(rng.randn(40, 16).astype(np.float32))
    labels = [5, 0, 39, 5, 17]
    out["region_embeddings"] = dict(memory=mem, labels=labels, out=emb(me, mem, labels).clone())
    torch.save(out, os.path.join(HERE, "oa_inputs.pt"))
    print("wrote oa_inputs.pt:", {k: len(v) if hasattr(v, "__len__") else v for k, v in out.items()})


if __name__ == "__main__":
    main()
