"""FrozenInTime, region-memory variant
(/root/reference/OATrans/model/oa_model_region_mem.py:19-151): the input holds F = 2*T' frames per
sample; `view(2B, T', ...)` makes even clips the "object frames" and odd clips the video (:109-115) - or, in the
'native' clip layout (oa_model_global_local.py docstring), frame 0 is the object frame and frames 1..T the video.
vid_proj is applied to the CLS and to every block-6 region token; the video embedding is
(cls + mean regions)/2; region_sim = sigmoid(text-region x object-region^T)."""
import os

import torch
from torch import nn

from ..ops import hip
from ..utils.util import state_dict_data_parallel_fix
from .layers import HipLinear, ReLULinear, sim_matrix  # noqa: F401
from .oa_layers import mean_rows, mix, region_sim
from .oa_model import BaseModel, FrozenInTime as _Plain, TEXT_BWD_FIRST, VIT_INIT
from .oa_model_global_local import encode_object_and_video
from .oa_video_transformer_region import SpaceTimeTransformer
from .text_transformer import DistilBertHIP


class FrozenInTime(BaseModel):
    def __init__(self, video_params, object_params, text_params, projection_dim=256, load_checkpoint=None,
                 projection='minimal', load_temporal_fix='zeros'):
        super().__init__()
        self.video_params, self.text_params, self.object_params = video_params, text_params, object_params
        self.load_temporal_fix = load_temporal_fix
        if not text_params['pretrained']:
            raise NotImplementedError("Huggingface text models require pretrained init.")
        tname = text_params['model']
        self.text_model = DistilBertHIP.from_pretrained(tname) if os.path.isdir(tname) else DistilBertHIP(text_params.get('config'))
        self.text_model.train()
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
        self.txt_proj = ReLULinear(self.text_model.config.hidden_size, projection_dim, xavier=True)
        self.txt_proj_2 = ReLULinear(512, projection_dim, xavier=True)
        self.vid_proj = nn.Sequential(HipLinear(model.embed_dim, projection_dim))
        nn.init.xavier_uniform_(self.vid_proj[0].weight)
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

    def forward(self, data, aug=False, return_embeds=True):
        # text side (DistilBERT pass, txt_proj_2 of the class-prompt embeddings) on its own stream beneath the video encoder
        main = torch.cuda.current_stream()
        if getattr(self, "_text_stream", None) is None:
            self._text_stream = hip.side_stream("text")
        side = self._text_stream
        side.wait_stream(main)
        early = TEXT_BWD_FIRST and torch.is_grad_enabled()       # kernels now, autograd node after the video side (DistilBertHIP.launch)
        with torch.cuda.stream(side):
            if early:
                ticket = self.text_model.launch(input_ids=data['text']['input_ids'], attention_mask=data['text'].get('attention_mask'))
            else:
                text_embeddings = self.compute_text(data['text'])
                text_region = self.txt_proj_2(data['text_region_embedding'].float())
        # clip layouts ('interleaved' = the reference's view(2B, F/2), 'native' = object frame + T-frame video):
        # oa_model_global_local.encode_object_and_video
        _, object_region, video_embeddings, video_region = encode_object_and_video(self, data['video'])
        video_embeddings = mix(video_embeddings, mean_rows(video_region), 0.5, 0.5)
        if early:
            with torch.cuda.stream(side):
                text_embeddings = self.compute_text(data['text'], launched=ticket)
                text_region = self.txt_proj_2(data['text_region_embedding'].float())
        main.wait_stream(side)
        text_embeddings.record_stream(main)
        text_region.record_stream(main)
        return text_embeddings, video_embeddings, self.compute_region_sim(object_region, text_region)

    def compute_text(self, text_data, pad=False, launched=None):
        hidden = self.text_model(input_ids=text_data['input_ids'], attention_mask=text_data.get('attention_mask'), launched=launched).last_hidden_state
        return self.txt_proj(hidden[:, 0, :].float())

    def compute_video(self, video_data):
        cls, region = self.video_model(video_data)
        return self.vid_proj(cls), self.vid_proj(region)

    def compute_videos(self, clips):
        """compute_video for several clips encoded together"""
        return [(self.vid_proj(cls), self.vid_proj(region)) for cls, region in self.video_model.forward_clips(clips)]

    def compute_region_sim(self, video_feats, text_feats):
        return region_sim(text_feats, video_feats)
