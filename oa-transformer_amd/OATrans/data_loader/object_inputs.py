"""Object-aware input producers: what the reference's OA datasets compute per sample between the video decoder and the
model.  Contract (names and argument meaning follow the reference's methods):

  sample_frames / select_object_frame   which frames of a video form the clip and which one is the OBJECT frame
                                        (base_dataset_global_local.py:871-907, base_dataset_region_mem.py:596-607)
  DetectorFile / read_bboxs_tags        the object detector's .npz wire format: x [n, 2048] region features,
                                        bbox [n, 4] pixel boxes, info {objects_conf, objects_id, image_w, image_h}
                                        -> tag sentence, class ids, 6-d normalised box features (:428-472)
  object_tags_masks                     cumulative tag-token ends for the tag-mask kernel (:395-405)
  pack_clip                             [object frame | T clip frames] -> zero-padded [T + 1, 3, R, R] (:669-671) -
                                        the 'native' object-clip layout of model/oa_model_global_local.py
  region_embeddings                     rows of the CLIP class-prompt memory (base_dataset_region_mem.py:298-302)
  ObjectBatch                           collate: per-sample host results -> ONE device batch; box masks are rasterised
                                        on the GPU (oat_patch_masks) from the 6-d features instead of by the
                                        reference's per-sample numpy loops (:348-356, region_mem :233-247)

The host part is array code (no per-box Python loops); everything per-batch and data-parallel runs on the device.
Decoding video files and image transforms are the input pipeline and not part of this module."""
import random

import numpy as np
import torch


def sample_frames(num_frames, vlen, sample="rand", fix_start=None, rng=random):
    """One frame index per equal interval of [0, vlen).  sample='rand' draws with `rng.choice(range(lo, hi))` per
    interval - the same consumption of Python's generator as the reference, so a seeded run picks the same frames."""
    acc = min(num_frames, vlen)
    edges = np.linspace(start=0, stop=vlen, num=acc + 1).astype(int)
    lo, hi = edges[:-1], edges[1:] - 1
    if sample == "rand":
        return [rng.choice(range(int(a), int(b))) for a, b in zip(lo, hi)]
    if fix_start is not None:
        return [int(a) + fix_start for a in lo]
    if sample == "uniform":
        return [int(v) for v in (lo + hi) // 2]
    raise NotImplementedError(sample)


def select_object_frame(frame_idxs, vlen, object_num="part"):
    """-> (frame indices to decode, object frame FIRST; index of the detector file).  'part': detector files exist for
    8 uniformly spaced frames, the nearest to the clip's mean index is used (ties: the earlier one); 'full': one file
    per frame, the mean index itself."""
    mean = int(sum(frame_idxs) / len(frame_idxs))
    if object_num == "full":
        return [mean] + list(frame_idxs), mean
    cand = np.asarray(sample_frames(8, vlen, sample="uniform"))
    dist = np.abs(cand - mean)
    nearest = int(cand[int(np.argmin(dist))])                  # first minimum, like min(key=...)
    object_index = int(np.flatnonzero(cand == nearest)[-1])     # the reference keeps the LAST position holding that index
    return [nearest] + list(frame_idxs), object_index


def parse_vocab(lines):
    """Class-name table of the detector vocabulary file (one 'name,alias,...' line per class); entry 0 = background."""
    return ["__background__"] + [line.split(",")[0].lower().strip() for line in lines]


class DetectorFile:
    """One detector output (np.load(path, allow_pickle=True), or any mapping with x / bbox / info)."""

    def __init__(self, source):
        frame = np.load(source, allow_pickle=True) if isinstance(source, (str, bytes)) or hasattr(source, "__fspath__") else source
        info = frame["info"]
        info = info.item() if hasattr(info, "item") and not isinstance(info, dict) else info
        self.bbox = np.asarray(frame["bbox"])
        self.conf = np.asarray(info["objects_conf"])
        self.ids = np.asarray(info["objects_id"])
        self.image_w, self.image_h = info["image_w"], info["image_h"]
        self._frame = frame

    @property
    def features(self):
        return np.asarray(self._frame["x"])


def read_bboxs_tags(source, classes, top_k=10, v=1):
    """-> (tag sentence ' name name ...', class ids [top_k], box features [top_k, 6] = x0 y0 x1 y1 w h in units of the
    image size).  Boxes are ranked by confidence; v=2 keeps one box per class (ascending class id) when at least top_k
    classes are present; fewer than top_k boxes: the last one is repeated."""
    det = source if isinstance(source, DetectorFile) else DetectorFile(source)
    order = np.argsort(det.conf)[::-1]
    boxes, ids = det.bbox[order], det.ids[order]
    if v == 2:
        _, first = np.unique(ids, return_index=True)
        if len(first) >= top_k:
            boxes, ids = boxes[first], ids[first]
    if boxes.shape[0] < top_k:
        take = np.minimum(np.arange(top_k), boxes.shape[0] - 1)
        boxes, ids = boxes[take], ids[take]
    boxes, ids = boxes[:top_k, :4], ids[:top_k]
    tags = "".join(" " + classes[i + 1] for i in ids.tolist())
    scale = np.array([det.image_w, det.image_h, det.image_w, det.image_h], dtype=boxes.dtype)
    xyxy = boxes / scale
    wh = (boxes[:, 2:4] - boxes[:, 0:2]) / scale[:2]
    feats = np.concatenate([xyxy[:, :2], xyxy[:, :2] + wh, wh], axis=1)
    return tags, ids, torch.from_numpy(np.ascontiguousarray(feats))


def object_tags_masks(ids, token_lens):
    """-> (cumulative tag-token end per tag [len(ids)] float, total tag tokens).  `token_lens[class id]` = number of
    tokeniser pieces of the class name (utils/objects_vocab_token_len.txt in the reference)."""
    lens = np.asarray(token_lens)[np.asarray(ids, dtype=np.int64)].astype(np.int64)
    ends = np.cumsum(lens)
    return torch.from_numpy(ends.astype(np.float32)), int(ends[-1]) if len(ends) else 0


def region_embeddings(memory, labels):
    return memory[torch.as_tensor(np.asarray(labels, dtype=np.int64))].float()


def pack_clip(imgs, num_frames, res, out=None):
    """imgs [f <= T + 1, 3, res, res] (object frame first) -> [T + 1, 3, res, res]; frames that failed to decode leave
    zero frames at the end.  `out` may be a slice of a pinned / device batch buffer."""
    if out is None:
        out = torch.zeros(num_frames + 1, 3, res, res, dtype=imgs.dtype, device=imgs.device)
    else:
        out.zero_()
    out[:imgs.shape[0]] = imgs
    return out


def select_region_classes(ids, para_num=5, rng=random):
    """The region-memory variant's draw (base_dataset_region_mem.py:235-242): para_num boxes without replacement;
    their classes are the 5 region prompts of the sample."""
    picks = rng.sample(range(0, len(ids)), para_num)
    return [int(ids[i]) for i in picks]


class ObjectBatch:
    """Collates per-sample producer results into the batch dict the OA models consume, with the box masks rasterised on
    the device.  variant: 'global_local' (all top_k boxes, one mask each) or 'region_mem' (para_num class-union masks)."""

    def __init__(self, variant="global_local", patch_rows=14):
        if variant not in ("global_local", "region_mem"):
            raise ValueError(variant)
        self.variant, self.patch_rows = variant, patch_rows

    def __call__(self, samples, device):
        """samples: dicts with 'video' [T+1,3,R,R], 'bboxs' [K,6], and - global_local: 'object_token_masks',
        'object_token_len'; region_mem: 'box_class' [K], 'sel_class' [5], 'text_region_embedding' [5, 512].
        Text fields ('text', 'pad_text') are passed through as lists for the tokenizer."""
        try:
            from ..ops import hip
        except ImportError:
            from ops import hip
        dev = torch.device(device)
        batch = {"video": torch.stack([s["video"] for s in samples]).to(dev, non_blocking=True)}
        boxes = torch.stack([torch.as_tensor(s["bboxs"]).float() for s in samples]).to(dev)
        if self.variant == "global_local":
            batch["patch_masks"] = hip.patch_masks(boxes, self.patch_rows)
            batch["object_token_masks"] = torch.stack([torch.as_tensor(s["object_token_masks"]) for s in samples]).to(dev)
            batch["object_token_len"] = torch.as_tensor([int(s["object_token_len"]) for s in samples], device=dev)
        else:
            cls = torch.stack([torch.as_tensor(np.asarray(s["box_class"]), dtype=torch.int32) for s in samples]).to(dev)
            sel = torch.stack([torch.as_tensor(np.asarray(s["sel_class"]), dtype=torch.int32) for s in samples]).to(dev)
            batch["patch_masks"] = hip.patch_masks(boxes, self.patch_rows, box_class=cls, sel_class=sel)
            batch["text_region_embedding"] = torch.stack([s["text_region_embedding"] for s in samples]).to(dev)
        for k in ("text", "pad_text"):
            if k in samples[0]:
                batch[k] = [s[k] for s in samples]
        batch["meta"] = [s.get("meta") for s in samples]
        return batch
