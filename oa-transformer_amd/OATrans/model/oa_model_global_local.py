"""FrozenInTime, global + local variant
(/root/reference/OATrans/model/oa_model_global_local.py:16-230).  Two text passes (caption, caption +
object tags) with pooling tok0 + mean(tok1:) (:217); 2B-clip video stream; local features by mask-pooling:
region_feat = vid_local_proj(patch_masks @ object patches) (:178), tags_feat = text_local_proj(tags_masks
@ pad-text tokens) (:200).  The reference builds tags_masks with a Python loop over B x O on the host
(:183-196); here it is one device kernel.  `cross_model = CrossModalityFusion()` (:143) is undefined in the
reference and never used in forward - it is not reproduced."""
import os

import torch
from torch import nn

from ..ops import hip
from ..utils.util import state_dict_data_parallel_fix
from .layers import HipLinear, ReLULinear, sim_matrix  # noqa: F401
from .oa_layers import mask_pool, mean_rows, mix
from .oa_model import BaseModel, FrozenInTime as _Plain, VIT_INIT
from .oa_video_transformer_global_local import SpaceTimeTransformer
from .text_transformer import DistilBertHIP


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

    def forward(self, data, return_embeds=True):
        text_embeddings, text_tokens = self.compute_text(data['text'])
        pad_text_embeddings, pad_tokens = self.compute_text(data['pad_text'])
        v = data['video']
        v = v.view(v.size(0) * 2, -1, v.size(2), v.size(3), v.size(4))
        vision_embeddings, vision_region = self.compute_video(v)
        object_image_embeddings, object_region = vision_embeddings[0::2], vision_region[0::2]
        video_embeddings, video_region = vision_embeddings[1::2], vision_region[1::2]
        region_feat = mask_pool(data['patch_masks'].float(), object_region)
        n_txt = data['text']['attention_mask'].sum(dim=1)
        tags_masks = hip.tag_masks(data['object_token_masks'].to(torch.int64), n_txt.to(torch.int64), pad_tokens.shape[1])
        tags_feat = mask_pool(tags_masks, pad_tokens)
        region_feat = self.vid_local_proj(region_feat)
        tags_feat = self.text_local_proj(tags_feat)
        return text_embeddings, pad_text_embeddings, video_embeddings, object_image_embeddings, \
            [text_tokens, pad_tokens, video_region, object_region, region_feat, tags_feat]

    def compute_text(self, text_data):
        hidden = self.text_model(input_ids=text_data['input_ids'], attention_mask=text_data.get('attention_mask')).last_hidden_state
        pooled = mix(hidden[:, 0, :], mean_rows(hidden[:, 1:, :]), 1.0, 1.0)
        return self.txt_proj(pooled), hidden

    def compute_video(self, video_data):
        emb, region = self.video_model(video_data)
        return self.vid_proj(emb), region
