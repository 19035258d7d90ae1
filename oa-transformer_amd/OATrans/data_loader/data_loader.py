"""Synthetic-tensor loaders with the reference's class names and constructor kwargs
(/root/reference/OATrans/data_loader/data_loader.py:165-227).  The throughput metric is defined on
synthetic video / caption tensors and the reference's dataset code needs
cv2/decord/real files, so the hot path ships loaders that emit the SAME batch dictionary
({'video': [B,T,3,R,R], 'text': ..., 'meta': ...}) from seeded generators.

Captions are emitted pre-tokenised ({'input_ids','attention_mask'} int64) because no tokenizer
vocabulary exists offline; the trainers tokenise only when `tokenizer is not None and text is a
list of strings` (trainer_dist.py:151-152 semantics otherwise unchanged).
"""
import math

import torch


class _EpochSampler:
    """The slice of DistributedSampler the trainers touch (`set_epoch`, base_data_loader.py:120)."""

    def __init__(self):
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch


class _SyntheticLoader:
    def __init__(self, dataset_name, text_params, video_params, data_dir, object_dir=None, metadata_dir=None,
                 split='train', tsfm_params=None, tsfm_split=None, cut=None, subsample=1, sliding_window_stride=-1,
                 reader='synthetic', batch_size=1, num_workers=0, shuffle=True, object_params=None, args=None,
                 n_samples=None, **unused):
        self.dataset_name = dataset_name
        self.batch_size = batch_size
        self.split = split
        self.video_params = dict(video_params)
        self.text_params = dict(text_params or {})
        self.object_params = dict(object_params or {})
        self.args = args
        self.world_size = getattr(args, 'world_size', 1) if args is not None else 1
        self.rank = getattr(args, 'rank', 0) if args is not None else 0
        self.n_samples = n_samples or (batch_size * self.world_size * (8 if split == 'train' else 2))
        self.train_sampler = _EpochSampler()
        self.device = None                 # generate straight on the training device when set

    def __len__(self):
        return self.n_samples // (self.batch_size * self.world_size)       # drop_last=True

    def make_batch(self, seed, device=None, dtype=torch.float32):
        g = torch.Generator(device='cpu').manual_seed(seed)
        B, T = self.batch_size, self.video_params.get('num_frames', 1)
        R = self.video_params.get('input_res', 224)
        L = self.text_params.get('max_length', 32)
        video = torch.randn(B, T, 3, R, R, generator=g).to(dtype)
        ids = torch.randint(1000, 30000, (B, L), generator=g)
        ids[:, 0], ids[:, -1] = 101, 102
        batch = {'video': video, 'text': {'input_ids': ids, 'attention_mask': torch.ones(B, L, dtype=torch.int64)},
                 'meta': {'paths': [f'synthetic/{seed}/{i}' for i in range(B)], 'dataset': [self.dataset_name] * B}}
        op = self.object_params
        if op.get('input_objects') or op.get('pseudo_labels') or op.get('input_object_bboxs'):
            # object-aware fields in the formats the reference datasets emit:
            #   patch_masks        [B, O, (R/16)^2] 0/1 - boxes rasterised on the patch grid (14x14 at 224^2)
            #                      (base_dataset_global_local.py:348-356); region_mem uses O = 5 (:233-247)
            #   object_token_masks [B, O] cumulative tag-token ends, object_token_len [B]
            #   pad_text           caption + object tags, pre-tokenised
            #   text_region_embedding [B, 5, 512] (CLIP text features of 5 sampled classes)
            O, g14 = int(op.get('num_objects', 10)), R // 16        # patch grid of the frame: 14 at 224^2, 21 at 336^2
            x0 = torch.randint(0, g14 - 1, (B, O), generator=g)
            y0 = torch.randint(0, g14 - 1, (B, O), generator=g)
            x1 = x0 + 1 + torch.randint(0, g14, (B, O), generator=g) % (g14 - x0)
            y1 = y0 + 1 + torch.randint(0, g14, (B, O), generator=g) % (g14 - y0)
            xs = torch.arange(g14)[None, None, :]
            in_x = (xs >= x0[..., None]) & (xs < x1[..., None])
            in_y = (xs >= y0[..., None]) & (xs < y1[..., None])
            batch['patch_masks'] = (in_y[..., :, None] & in_x[..., None, :]).reshape(B, O, g14 * g14).float()
            # the same boxes in the wire format of the object extractor (x0, y0, x1, y1 normalised to the frame,
            # a quarter cell inside the grid lines): on a GPU the masks are produced from these by oat_patch_masks
            batch['bboxs'] = torch.stack([x0 + 0.25, y0 + 0.25, x1 - 0.25, y1 - 0.25], dim=-1).float() / g14
            ntok = torch.randint(1, 4, (B, O), generator=g)
            batch['object_token_masks'] = ntok.cumsum(dim=1)
            batch['object_token_len'] = batch['object_token_masks'][:, -1].clone()
            Lp = L + int(3 * O)
            pids = torch.randint(1000, 30000, (B, Lp), generator=g)
            pids[:, 0] = 101
            batch['pad_text'] = {'input_ids': pids, 'attention_mask': torch.ones(B, Lp, dtype=torch.int64)}
            batch['text_region_embedding'] = torch.randn(B, 5, 512, generator=g)
        if device is not None and torch.device(device).type == 'cuda' and 'bboxs' in batch:
            try:
                from OATrans.ops import hip
            except ImportError:
                from ops import hip
            batch['patch_masks'] = hip.patch_masks(batch['bboxs'].to(device), g14)      # bbox -> mask on the device
        if device is not None:
            for k, v in list(batch.items()):
                if isinstance(v, torch.Tensor):
                    batch[k] = v.to(device)
                elif isinstance(v, dict) and k != 'meta':
                    batch[k] = {kk: vv.to(device) for kk, vv in v.items()}
        return batch

    def __iter__(self):
        for i in range(len(self)):
            yield self.make_batch(1234 + self.rank + 7919 * (i + 1000 * self.train_sampler.epoch), self.device)


class TextObjectVideoDataLoader(_SyntheticLoader):
    """Single-process loader used by train.py (reference :165-195)."""


class MultiDistTextObjectVideoDataLoader(_SyntheticLoader):
    """One loader per rank, sharded by (rank, world_size) from `args` (reference :197-227)."""
