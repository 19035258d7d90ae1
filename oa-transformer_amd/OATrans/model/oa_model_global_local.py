"""FrozenInTime, global + local variant
(/root/reference/OATrans/model/oa_model_global_local.py:16-230).  Two text passes (caption, caption +
object tags) with pooling tok0 + mean(tok1:) (:217); 2B-clip video stream; local features by mask-pooling:
region_feat = vid_local_proj(patch_masks @ object patches) (:178), tags_feat = text_local_proj(tags_masks
@ pad-text tokens) (:200).  The reference builds tags_masks with a Python loop over B x O on the host
(:183-196); here it is one device kernel.  `cross_model = CrossModalityFusion()` (:143) is undefined in the
reference and never used in forward - it is not reproduced.

Object clip layouts (`video_params['object_clip']`):
  'interleaved'  the reference's: data['video'] [B, F, ...] is VIEWED as 2B clips of F/2 frames (:170) - the even
                 clips are called object images, the odd ones videos.  Only F = 2 gives what the names say.
  'native'       data['video'] = [object frame | T video frames] (F = 1 + T): the one-frame object clip and the
                 T-frame video clip go through the SAME encoder weights as two calls (gradients of the second
                 accumulate into the first's); mask-pooling and the region similarity are unchanged.  This is the
                 form config 3 of BASELINE.json needs at 8 frames ("8-frame + 10 obj"); at F = 2 both layouts are
                 the same computation (tests/test_oa_gpu.py pins native against the F = 2 goldens).
  'auto' (default)  native when F is odd, interleaved otherwise."""
import os

import torch
from torch import nn

from ..ops import hip
from ..utils.util import state_dict_data_parallel_fix
from .layers import HipLinear, ReLULinear, sim_matrix  # noqa: F401
from .oa_layers import mask_pool, mean_rows, mix
from .oa_model import BaseModel, FrozenInTime as _Plain, TEXT_BWD_FIRST, VIT_INIT
from .oa_video_transformer_global_local import SpaceTimeTransformer
from .text_transformer import DistilBertHIP


def encode_object_and_video(model, v):
    """Shared by the global+local and the region-memory models: run the video encoder over the object clip and the
    video clip of every sample.  Returns (object_emb, object_region, video_emb, video_region) where *_region are
    whatever `model.compute_video` returns second."""
    layout = model.video_params.get('object_clip', 'auto')
    F = v.size(1)
    if layout == 'auto':
        layout = 'native' if F % 2 else 'interleaved'
    if layout == 'interleaved':
        v = v.view(v.size(0) * 2, -1, v.size(2), v.size(3), v.size(4))     # oa_model_global_local.py:170
        emb, region = model.compute_video(v)
        return emb[0::2], region[0::2], emb[1::2], region[1::2]
    if layout != 'native':
        raise ValueError(f"object_clip = {layout!r}: expected 'auto', 'interleaved' or 'native'")
    if F < 2:
        raise ValueError("native object clip layout needs the object frame and at least one video frame")
    # Both clips go through the encoder as two SEGMENTS of one launch sequence (engine/video.py): alone on the GPU the object
    # clip's GEMMs are 75-300 tiles wide and leave most CUs idle (it cost 14 ms for 12.5 % of the tokens; a second stream
    # beside the video clip did not help, 70.2 vs 69.8 ms), as extra rows of the video clip's launches it is nearly free.
    # (video_params['object_clip_segments'] = False: two encoder calls, the second backward accumulating into the first's gradients)
    if model.video_params.get('object_clip_segments', True) and hasattr(model, "compute_videos"):
        (obj_emb, obj_region), (vid_emb, vid_region) = model.compute_videos([v[:, :1], v[:, 1:]])
        return obj_emb, obj_region, vid_emb, vid_region
    obj_emb, obj_region = model.compute_video(v[:, :1])
    vid_emb, vid_region = model.compute_video(v[:, 1:])
    return obj_emb, obj_region, vid_emb, vid_region


class FrozenInTime(BaseModel):
    def __init__(self, video_params, object_params, text_params, projection_dim=256, load_checkpoint=None,
                 projection='minimal', load_temporal_fix='zeros'):
        super().__init__()
        self.video_params, self.text_params, self.object_params = video_params, text_params, object_params
        self.load_temporal_fix = load_temporal_fix
        if not text_params['pretrained']:
            raise NotImplementedError("Huggingface text models require pretrained init.")
        if object_params['model'] != "":
            raise NotImplementedError("only object_params.model == '' exists in the reference (SimpleMLP / ObjectTransformer are undefined there)")
        tname = text_params['model']
        self.text_model = DistilBertHIP.from_pretrained(tname) if os.path.isdir(tname) else DistilBertHIP(text_params.get('config'))
        self.text_model.train()
        self.object_model = None
        if video_params['model'] != "SpaceTimeTransformer" or video_params.get('arch_config', 'base_patch16_224') != 'base_patch16_224':
            raise NotImplementedError(f"{video_params['model']} not implemented")
        model = SpaceTimeTransformer(num_frames=video_params.get('num_frames', 4), time_init=video_params.get('time_init', 'zeros'),
                                     attention_style=video_params.get('attention_style', 'frozen-in-time'),
                                     **video_params.get('arch_kwargs', {}))
        model.head = nn.Identity()
        model.pre_logits = nn.Identity()
        if load_checkpoint in ("", None) and os.path.exists(VIT_INIT):
            model.load_state_dict(torch.load(VIT_INIT, map_location="cpu"), strict=False)
        self.video_model = model
        self.video_model.fc = nn.Identity()
        if projection != 'minimal':
            raise NotImplementedError
        hid = self.text_model.config.hidden_size
        self.txt_proj = ReLULinear(hid, projection_dim)
        self.text_local_proj = ReLULinear(hid, projection_dim)
        self.vid_proj = nn.Sequential(HipLinear(model.embed_dim, projection_dim))
        self.vid_local_proj = nn.Sequential(HipLinear(model.embed_dim, projection_dim))
        if load_checkpoint not in ("", None):
            checkpoint = torch.load(load_checkpoint, map_location="cpu")
            sd = state_dict_data_parallel_fix(checkpoint['state_dict'], self.state_dict())
            self.load_state_dict(self._inflate_positional_embeds(sd), strict=False)

    _inflate_positional_embeds = _Plain._inflate_positional_embeds

    def set_device(self, device):
        self.device = device

    def begin_step(self):
        self.text_model.begin_step()
        self.video_model.begin_step()

    def encode_clips(self, v):
        """data['video'] -> (object_emb, object_region, video_emb, video_region) in the configured clip layout."""
        return encode_object_and_video(self, v)

    def forward(self, data, return_embeds=True):
        # everything on the text side (two DistilBERT passes, tag masks, tag pooling, text_local_proj) is independent of the
        # video side until the losses: it runs on its own HIP stream beneath the video encoder, enqueued first (as in
        # oa_model.FrozenInTime.forward); autograd replays each side's backward on the stream of its forward
        main = torch.cuda.current_stream()
        if getattr(self, "_text_stream", None) is None:
            self._text_stream = hip.side_stream("text")
        side = self._text_stream
        side.wait_stream(main)
        def text_side(t1=None, t2=None):
            text_embeddings, text_tokens = self.compute_text(data['text'], launched=t1)
            pad_text_embeddings, pad_tokens = self.compute_text(data['pad_text'], launched=t2)
            n_txt = data['text']['attention_mask'].sum(dim=1)
            tags_masks = hip.tag_masks(data['object_token_masks'].to(torch.int64), n_txt.to(torch.int64), pad_tokens.shape[1])
            return text_embeddings, text_tokens, pad_text_embeddings, pad_tokens, self.text_local_proj(mask_pool(tags_masks, pad_tokens))

        # both DistilBERT passes are enqueued now, their autograd nodes (and the small pooling / projection launches behind them)
        # after the video side: backward then issues the text side first (DistilBertHIP.launch)
        early = TEXT_BWD_FIRST and torch.is_grad_enabled()
        with torch.cuda.stream(side):
            if early:
                t1 = self.text_model.launch(input_ids=data['text']['input_ids'], attention_mask=data['text'].get('attention_mask'))
                t2 = self.text_model.launch(input_ids=data['pad_text']['input_ids'], attention_mask=data['pad_text'].get('attention_mask'))
            else:
                text_embeddings, text_tokens, pad_text_embeddings, pad_tokens, tags_feat = text_side()
        object_image_embeddings, object_region, video_embeddings, video_region = self.encode_clips(data['video'])
        region_feat = self.vid_local_proj(mask_pool(data['patch_masks'].float(), object_region))
        if early:
            with torch.cuda.stream(side):
                text_embeddings, text_tokens, pad_text_embeddings, pad_tokens, tags_feat = text_side(t1, t2)
        main.wait_stream(side)
        for t in (text_embeddings, text_tokens, pad_text_embeddings, pad_tokens, tags_feat):
            t.record_stream(main)
        return text_embeddings, pad_text_embeddings, video_embeddings, object_image_embeddings, \
            [text_tokens, pad_tokens, video_region, object_region, region_feat, tags_feat]

    def compute_text(self, text_data, launched=None):
        hidden = self.text_model(input_ids=text_data['input_ids'], attention_mask=text_data.get('attention_mask'), launched=launched).last_hidden_state
        pooled = mix(hidden[:, 0, :], mean_rows(hidden[:, 1:, :]), 1.0, 1.0)
        return self.txt_proj(pooled), hidden

    def compute_video(self, video_data):
        emb, region = self.video_model(video_data)
        return self.vid_proj(emb), region

    def compute_videos(self, clips):
        """compute_video for several clips encoded together"""
        return [(self.vid_proj(emb), region) for emb, region in self.video_model.forward_clips(clips)]
