"""SpaceTimeTransformer, global+local variant
(/root/reference/OATrans/model/oa_video_transformer_global_local.py:352-359): the tail returns
(1/2 norm(x)[:,0] + 1/2 mean(norm(x)[:,1:]), norm(x)[:,1:]).  Carries the unused `object_embed`."""
from torch import nn

from .oa_layers import mean_rows, mix
from .video_transformer import SpaceTimeTransformer as _Base


class SpaceTimeTransformer(_Base):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.object_embed = nn.Linear(2054, self.embed_dim)        # declared, never used in forward
        self.need_patch_tokens = True

    def forward(self, x):
        cls, patches = self.forward_features(x)
        return mix(cls, mean_rows(patches), 0.5, 0.5), patches

    def forward_clips(self, clips):
        """several clips in one launch sequence -> [(0.5 cls + 0.5 mean patches, patches), ...]"""
        return [(mix(cls, mean_rows(patches), 0.5, 0.5), patches) for cls, patches in self.forward_features_clips(clips)]
