"""SpaceTimeTransformer, region variant (/root/reference/OATrans/model/oa_video_transformer_region.py):
same blocks; the tail returns (norm(x)[:,0], region_norm(x after block 6)[:,1:]) (:364-376).  Carries the
reference's extra parameters `region_norm` and the unused `object_embed` (:250) for checkpoint parity."""
from torch import nn

from .video_transformer import SpaceTimeTransformer as _Base

REGION_LAYER = 6


class SpaceTimeTransformer(_Base):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.object_embed = nn.Linear(2054, self.embed_dim)        # declared, never used in forward
        self.region_norm = nn.LayerNorm(self.embed_dim, eps=1e-6)
        self.need_patch_tokens = False
        self.region_layer = REGION_LAYER
        if len(self.blocks) < REGION_LAYER:
            raise ValueError("the region variant taps block 6: depth >= 6 required")

    def forward(self, x):
        cls, _, region = self.forward_features(x)
        return cls, region

    def forward_clips(self, clips):
        """several clips in one launch sequence -> [(cls, region), ...]"""
        return [(cls, region) for cls, _, region in self.forward_features_clips(clips)]
