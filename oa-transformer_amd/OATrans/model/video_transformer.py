"""SpaceTimeTransformer - host-side mirror of the reference class, executed by HIP kernels.

Same constructor signature, parameter names and return convention as
/root/reference/OATrans/model/video_transformer.py:179-357 so that reference checkpoints load
(`blocks.{i}.{norm1,norm2,norm3,attn.qkv,attn.proj,timeattn.qkv,timeattn.proj,mlp.fc1,mlp.fc2}`,
`cls_token`, `pos_embed`, `temporal_embed`, `patch_embed.proj`, `norm`).  The nn.Module tree
below only *holds parameters*; forward/backward are the launch schedules of
OATrans.engine.video.VideoEngine.  There is no eager fallback: without liboatrans_hip.so or
off-GPU the forward raises.
"""
from functools import partial

import torch
from torch import nn

from ..engine.module import EngineModule
from ..engine.video import VideoEngine
from ..ops import hip


def _trunc_normal_(t, std):
    return nn.init.trunc_normal_(t, std=std, a=-2.0, b=2.0)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _VarAttention(nn.Module):
    """Parameter holder for VarAttention (reference :79-97), incl. the 'zeros' time init that sets
    qkv to 0 and proj.weight to ONE (:89-95)."""

    def __init__(self, dim, qkv_bias, initialize):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        if initialize == "zeros":
            self.qkv.weight.data.fill_(0)
            self.qkv.bias.data.fill_(0)
            self.proj.weight.data.fill_(1)
            self.proj.bias.data.fill_(0)


class _Block(nn.Module):
    def __init__(self, dim, mlp_ratio, qkv_bias, norm_layer, time_init):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = _VarAttention(dim, qkv_bias, "random")
        self.timeattn = _VarAttention(dim, qkv_bias, time_init)
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.norm3 = norm_layer(dim)


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim, num_frames):
        super().__init__()
        self.img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.num_frames = num_frames
        self.embed_dim = embed_dim
        self.num_patches = (self.img_size[1] // self.patch_size[1]) * (self.img_size[0] // self.patch_size[0]) * num_frames
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)


class _EncoderFn(torch.autograd.Function):
    """Bridges the explicit schedules into autograd.  Parameter gradients are written by the
    kernels straight into persistent `.grad` buffers, so the function returns None for them.  The first backward of
    a step overwrites them, later ones (the encoder ran more than once: object clip + video clip of the object-aware
    models) accumulate; `begin_step()` of the owning model resets the call counters."""

    @staticmethod
    def forward(ctx, module, n_clips, need_patches, region_layer, *rest):
        """rest = n_clips clip tensors (encoded together as segments of one launch sequence, engine/video.py), then the
        parameters.  Returns (cls, patches | None, region | None) per clip, flattened."""
        clips, eng = list(rest[:n_clips]), module._engine
        pd = module._param_data()
        call = 0
        if module._track_calls:              # a forward that will be differentiated keeps its own activation plan
            module._new_step_guard()
            call = module._fwd_calls
            module._fwd_calls += 1
        cls, patches, plan = eng.forward(clips, pd, need_patches, module._weights_signature(), region_layer, call=call)
        ctx.module, ctx.plan, ctx.n_clips = module, plan, n_clips
        ctx.set_materialize_grads(False)
        D = cls[0].shape[-1]
        regions = plan.regions if region_layer is not None else [None] * n_clips
        outs = []
        for k, v in enumerate(clips):        # views of plan buffers that the next step overwrites; used within the step
            B = v.shape[0]
            outs += [cls[k].clone(), patches[k].view(B, -1, D) if need_patches else None,
                     regions[k].view(B, -1, D) if region_layer is not None else None]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        module, plan, n_clips = ctx.module, ctx.plan, ctx.n_clips
        D = module.embed_dim
        flat = lambda t: None if t is None else t.reshape(-1, D).float()
        d_cls = [flat(douts[3 * k]) for k in range(n_clips)]
        d_patches = [flat(douts[3 * k + 1]) for k in range(n_clips)]
        d_region = [flat(douts[3 * k + 2]) for k in range(n_clips)]
        accumulate = module._bwd_calls > 0
        module._bwd_calls += 1
        # autograd runs a backward on the stream of its forward.  Two clips encoded on two streams: the accumulating
        # backward must not overtake the one that overwrites the gradient buffers
        cur = torch.cuda.current_stream()
        prev = module.__dict__.get("_last_bwd_stream")
        if accumulate and prev is not None and prev != cur:
            cur.wait_stream(prev)
        module.__dict__["_last_bwd_stream"] = cur
        # a block's gradients are final - and may go to the all-reduce / eager optimiser - only in the LAST backward of
        # the step (earlier clips have been accumulated by then); if a clip's output never receives a gradient the
        # ranges stay unannounced and are handled after backward (GradSync.all_reduce / AdamW.step)
        last = module._bwd_calls >= module._fwd_calls
        ready = module._announce if (module.grad_ready_hook is not None and last) else None
        module._engine.backward(plan, module._param_data(), module._grad_views(), d_cls, d_patches, d_region,
                                ready=ready, accumulate=accumulate)
        if last:
            module._fwd_calls = module._bwd_calls = 0        # step complete without begin_step(): start over
        return (None, None, None, None) + (None,) * (n_clips + module._n_params)


class SpaceTimeTransformer(EngineModule):
    _skip_prefixes = ("head.",)

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, qk_scale=None, representation_size=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., hybrid_backbone=None, norm_layer=None,
                 num_frames=8, time_init='rand', attention_style='frozen-in-time'):
        super().__init__()
        if hybrid_backbone is not None:
            raise NotImplementedError('hybrid backbone not implemented')
        if attention_style != 'frozen-in-time':
            raise NotImplementedError(attention_style)                       # reference :171-172
        if drop_rate or attn_drop_rate or drop_path_rate:
            raise NotImplementedError("dropout / stochastic depth are 0 in every shipped config")
        if qk_scale is not None or not qkv_bias:
            raise NotImplementedError("qk_scale / qkv_bias=False are unused by the shipped configs")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_frames = num_frames
        self.num_heads = num_heads
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim, num_frames)
        self.patches_per_frame = self.patch_embed.num_patches // num_frames
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patches_per_frame + 1, embed_dim))
        self.temporal_embed = nn.Parameter(torch.zeros(1, num_frames, embed_dim))
        self.blocks = nn.ModuleList([_Block(embed_dim, mlp_ratio, qkv_bias, norm_layer, time_init)
                                     for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        if representation_size:
            raise NotImplementedError("representation_size is unused on the hot path")
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        _trunc_normal_(self.pos_embed, .02)
        _trunc_normal_(self.cls_token, .02)
        if num_frames == 1:
            self.apply(self._init_weights)
        self.need_patch_tokens = True
        self.region_layer = None
        self._fwd_calls = self._bwd_calls = 0
        self._track_calls = True
        self._engine = VideoEngine(depth, embed_dim, num_heads, mlp_ratio, self.patch_embed.patch_size[0], in_chans,
                                   num_frames)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, .02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def begin_step(self):
        """A new training step: the next backward overwrites the gradient buffers."""
        self._fwd_calls = self._bwd_calls = 0

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def forward_features(self, x, aug=False):
        if not x.is_cuda:
            raise hip.OatError("SpaceTimeTransformer runs on MI355X only (no CPU path); use the oracle for CPU")
        hip.lib()
        params = [p for _, p in self._engine_params()]
        self._track_calls = torch.is_grad_enabled()
        cls, patches, region = _EncoderFn.apply(self, 1, bool(self.need_patch_tokens), self.region_layer, x, *params)
        if self.region_layer is not None:
            return cls, patches, region
        return cls, patches

    def forward_features_clips(self, clips):
        """Several clips of different frame counts (same batch size is not required) through the encoder in ONE launch
        sequence - the object-aware models' object frame + video clip.  -> list of forward_features() results."""
        if not clips[0].is_cuda:
            raise hip.OatError("SpaceTimeTransformer runs on MI355X only (no CPU path); use the oracle for CPU")
        hip.lib()
        params = [p for _, p in self._engine_params()]
        self._track_calls = torch.is_grad_enabled()
        outs = _EncoderFn.apply(self, len(clips), bool(self.need_patch_tokens), self.region_layer, *clips, *params)
        res = []
        for k in range(len(clips)):
            cls, patches, region = outs[3 * k:3 * k + 3]
            res.append((cls, patches, region) if self.region_layer is not None else (cls, patches))
        return res

    def forward(self, x, aug=False):
        x = self.forward_features(x, aug=aug)
        if isinstance(self.head, nn.Identity):
            return x                      # oa_model.py:50 sets head = Identity -> tuple passes through
        return self.head(x[0])
