"""FrozenInTime - the contract class of the hot path
(/root/reference/OATrans/model/oa_model.py:10-133: same constructor arguments, attributes,
state_dict keys and forward convention), assembled from HIP-executed encoders."""
import os

import torch
import torch.nn.functional as F
from torch import nn

from ..ops import hip
from ..utils.util import state_dict_data_parallel_fix
from .layers import HipLinear, ReLULinear, sim_matrix  # noqa: F401  (sim_matrix re-exported like the reference)
from .text_transformer import DistilBertHIP
from .video_transformer import SpaceTimeTransformer

VIT_INIT = "pretrained/jx_vit_base_p16_224-80ecf9dd.pth"
# the text tower's kernels are enqueued first in forward, its autograd node is created last (DistilBertHIP.launch), so that
# backward issues the text tower BEFORE the video tower; 0: node and kernels together (the earlier order, for A/B runs)
TEXT_BWD_FIRST = True


class BaseModel(nn.Module):
    def __str__(self):
        n = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return super().__str__() + '\nTrainable parameters: {}'.format(n)


class FrozenInTime(BaseModel):
    def __init__(self, video_params, object_params, text_params, projection_dim=256, load_checkpoint=None,
                 projection='minimal', load_temporal_fix='zeros'):
        super().__init__()
        self.video_params = video_params
        self.text_params = text_params
        self.object_params = object_params
        self.load_temporal_fix = load_temporal_fix
        if not text_params['pretrained']:
            raise NotImplementedError("Huggingface text models require pretrained init.")
        tname = text_params['model']
        if not tname.split('/')[-1].startswith('distilbert'):
            raise NotImplementedError(f"text model {tname}: only DistilBERT is on the HIP path")
        if os.path.isdir(tname):
            self.text_model = DistilBertHIP.from_pretrained(tname)
        else:
            # offline box without the HF checkpoint: random init of the same geometry, said loudly
            print(f"### {tname} not found: DistilBERT-base geometry with random init")
            self.text_model = DistilBertHIP(text_params.get('config'))
        self.text_model.train()
        if video_params['model'] in ("SpaceTimeTransformer", "SpaceTimeObjectTransformer"):
            num_frames = video_params.get('num_frames', 4)
            time_init = video_params.get('time_init', 'zeros')
            attention_style = video_params.get('attention_style', 'frozen-in-time')
            arch_config = video_params.get('arch_config', 'base_patch16_224')
            if arch_config != 'base_patch16_224':
                raise NotImplementedError
            model = SpaceTimeTransformer(num_frames=num_frames, time_init=time_init, attention_style=attention_style,
                                         **video_params.get('arch_kwargs', {}))
            model.head = nn.Identity()
            model.pre_logits = nn.Identity()
            model.need_patch_tokens = False          # compute_video discards the patch tokens (:130)
            ftr_dim = model.embed_dim
            if load_checkpoint in ("", None) and os.path.exists(VIT_INIT):
                model.load_state_dict(torch.load(VIT_INIT, map_location="cpu"), strict=False)
            self.video_model = model
            self.video_model.fc = nn.Identity()
        elif video_params['model'] == "":
            print("no vision model available!")
        else:
            raise NotImplementedError(f"{video_params['model']} not implemented")
        if projection == 'minimal':
            self.txt_proj = ReLULinear(self.text_model.config.hidden_size, projection_dim)
            if video_params['model'] != "":
                self.vid_proj = nn.Sequential(HipLinear(ftr_dim, projection_dim))
        elif projection != '':
            self.txt_proj = nn.Identity()
            if video_params['model'] != "":
                self.vid_proj = nn.Identity()
        else:
            raise NotImplementedError
        if load_checkpoint not in ("", None):
            checkpoint = torch.load(load_checkpoint, map_location="cpu")
            state_dict = state_dict_data_parallel_fix(checkpoint['state_dict'], self.state_dict())
            self.load_state_dict(self._inflate_positional_embeds(state_dict), strict=False)

    def set_device(self, device):
        self.device = device

    def begin_step(self):
        self.text_model.begin_step()
        self.video_model.begin_step()

    def forward(self, data, aug=False, return_embeds=True):
        # the two towers are independent until the loss: the (small, launch-bound) text tower runs on its
        # own HIP stream under the video tower; autograd replays each tower's backward on its forward stream
        main = torch.cuda.current_stream()
        if getattr(self, "_text_stream", None) is None:
            self._text_stream = hip.side_stream("text")
        side = self._text_stream
        side.wait_stream(main)           # the text stream starts after what is on `main` NOW (the optimiser step)
        # host enqueue order: the text tower first.  Its ~80 launches are queued in about a millisecond and then run
        # beneath the first blocks of the video tower; queued behind the video tower's ~450 launches they started only
        # when the video forward was nearly over and added their whole length (fp32 forward: ~5 ms) to the step
        # ... and its autograd node LAST (DistilBertHIP.launch): backward then issues the text tower before the video tower
        early = TEXT_BWD_FIRST and torch.is_grad_enabled()
        with torch.cuda.stream(side):
            if early:
                ticket = self.text_model.launch(input_ids=data['text']['input_ids'], attention_mask=data['text'].get('attention_mask'))
            else:
                text_embeddings = self.compute_text(data['text'])
        video_embeddings = self.compute_video(data['video'], aug=aug)
        if early:
            with torch.cuda.stream(side):
                text_embeddings = self.compute_text(data['text'], launched=ticket)
        main.wait_stream(side)
        text_embeddings.record_stream(main)
        if return_embeds:
            return text_embeddings, video_embeddings
        return sim_matrix(text_embeddings, video_embeddings)

    def compute_text(self, text_data, pad=False, launched=None):
        hidden = self.text_model(input_ids=text_data['input_ids'],
                                 attention_mask=text_data.get('attention_mask'), launched=launched).last_hidden_state
        return self.txt_proj(hidden[:, 0, :].float())

    def compute_video(self, video_data, aug=False):
        video_embeddings, _ = self.video_model(video_data)
        return self.vid_proj(video_embeddings)

    def _inflate_positional_embeds(self, new_state_dict):
        """Load a checkpoint trained with a different frame count (reference :148-189):
        truncate, zero-pad or interpolate `video_model.temporal_embed`."""
        curr = self.state_dict()
        key = 'video_model.temporal_embed'
        if key in new_state_dict and key in curr:
            load = new_state_dict[key]
            n_load, n_curr, dim = load.shape[1], self.video_params['num_frames'], load.shape[2]
            if n_load > n_curr:
                print(f'### loaded {self.video_params["model"]} model has MORE frames than current...')
                new_state_dict[key] = load[:, :n_curr, :]
            elif n_load < n_curr:
                print(f'### loaded {self.video_params["model"]} model has FEWER frames than current...'
                      f'### filling in the extras via {self.load_temporal_fix}')
                if self.load_temporal_fix == 'zeros':
                    new = torch.zeros([load.shape[0], n_curr, dim])
                    new[:, :n_load] = load
                elif self.load_temporal_fix in ('interp', 'bilinear'):
                    mode = 'bilinear' if self.load_temporal_fix == 'bilinear' else 'nearest'
                    new = F.interpolate(load.unsqueeze(0), (n_curr, dim), mode=mode).squeeze(0)
                else:
                    raise NotImplementedError
                new_state_dict[key] = new
        key = 'video_model.pos_embed'
        if key in new_state_dict and key in curr and new_state_dict[key].shape[1] != curr[key].shape[1]:
            raise NotImplementedError('Loading models with different spatial resolution / patch number not yet implemented, sorry.')
        return new_state_dict
