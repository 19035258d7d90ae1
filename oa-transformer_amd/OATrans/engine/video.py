"""SpaceTimeTransformer forward/backward as an explicit kernel schedule over caller-owned HBM.

This is the MI355X-native execution of
  /root/reference/OATrans/model/video_transformer.py:303-351 (forward_features)
  /root/reference/OATrans/model/video_transformer.py:161-176 (SpaceTimeBlock.forward)
and of the autograd graph PyTorch would record for them.  Nothing here traces or compiles:
forward and backward are fixed launch sequences of liboatrans_hip.so entry points over activation
buffers sized once per shape ("plan"; 288 GB of HBM3E holds every saved activation of a bs-64
8-frame step, so nothing is recomputed).

HBM layout
  token rows  : patch (b,f,n) -> row (b*T+f)*N + n ; CLS(b) -> row B*T*N + b ; M = B*T*N + B
                rows, padded with zero rows to a multiple of 256 (GEMM tiles never branch on M).
                A forward may take SEVERAL clips of different frame counts (the object-aware models: one object frame +
                a T-frame clip per sample): each clip is a *segment* of the row space laid out as above, one after the
                other.  Every row-wise kernel (GEMMs, LayerNorm, quantisation, weight gradients) runs once over all
                segments - a 6304-row object clip alone leaves most CUs idle, as 12 % more rows of the video clip's
                launches it is free - and only attention, embedding and the CLS rows are handled per segment.
  residuals   : bf16 [Mp, D]          one tensor per block, its output x + space + mlp (round 4; sums in fp32, x + time and
                x + space never stored; the patch embedding x0 and the CLS lane's rows stay fp32; OAT_RES16=0 / fp8 mode:
                fp32 x, x+time, x+space, block output as in round 3)
  GEMM inputs : bf16 [Mp, D|3D|4D]    (LN outputs, qkv, attention outputs, MLP hidden)
  weights     : fp32 masters (nn.Parameters, reference state_dict names) + bf16 shadows
                W [out,in] (forward "NT" operand) and W^T [in,out] (data-gradient operand)

Concurrency (HIP streams and events, no graph compiler)
  * Every kernel of the video tower runs in order on the caller's stream and owns the whole GPU.  Measured on MI355X:
    two MFMA-bound kernels side by side only split the machine, and the HBM-bound kernels (LayerNorm, attention) are
    limited by per-CU load throughput, so confined to the CUs a GEMM leaves free they slow down by as much as the
    overlap would gain (the "slot" schedule of rounds 1-5 that ran them beside the weight gradients left the engine in round 6).
  * forward: the CLS-query attention (independent of the patch attention) runs beside it on a side stream; the text
    tower (0.7 % of the FLOPs, latency-bound) runs on its own stream beside the video tower.
"""
import os

import torch

from ..ops import hip


def _round_up(x, m):
    return (x + m - 1) // m * m


class _BlockActs:
    """Saved activations of one SpaceTimeBlock (all caller-owned, reused every step)."""

    def __init__(self, Mp, D, Hd, H, dev, res16=False):
        z16 = lambda c: torch.zeros(Mp, c, dtype=torch.bfloat16, device=dev)
        z32 = lambda c: torch.zeros(Mp, c, dtype=torch.float32, device=dev)
        self.a3, self.a1, self.a2 = z16(D), z16(D), z16(D)
        self.qkv_t, self.qkv_s = z16(3 * D), z16(3 * D)
        self.o_t, self.o_s = z16(D), z16(D)
        self.h, self.g = z16(Hd), z16(Hd)
        self.h8 = None                       # 8-bit view of h (engine.h_u8)
        # only the block output is stored (x + time and y = x + space are formed inside the LayerNorm kernels): bf16 on the
        # bf16 residual stream (default), fp32 with res16 off
        self.out = z16(D) if res16 else z32(D)
        self.lse_t, self.lse_s = z32(H), z32(H)
        self.stats = torch.zeros(6, Mp, dtype=torch.float32, device=dev)   # mean/rstd of norm3, norm1, norm2


class _Seg:
    """One clip batch [B, T] inside a plan's row space: rows [row0, row0 + B*T*N) are its patch tokens, the next B its CLS
    tokens; lane0 = its first row in the (sum of B)-row buffers of the CLS lane."""

    def __init__(self, B, T, N, row0, lane0, D, H, Kp, dev):
        self.B, self.T, self.N = B, T, N
        self.BTN = B * T * N
        self.M = self.BTN + B
        self.row0, self.cls0, self.end = row0, row0 + self.BTN, row0 + self.M
        self.lane0 = lane0
        self.cols = torch.zeros(_round_up(self.BTN, 256), Kp, dtype=torch.bfloat16, device=dev)
        self.table = torch.zeros(T * N, D, dtype=torch.float32, device=dev)
        self.cls_side = torch.zeros(B, H, 3, 64, dtype=torch.float32, device=dev)
        self.cls_done = torch.zeros(B, H, dtype=torch.int32, device=dev)       # tickets of the fused CLS-row finalize
        self.Gp = torch.zeros(T * N, D, dtype=torch.float32, device=dev)
        self.video = self.video_buf = None                # static copy of the input clip
        self.d_region_buf = None

    def rows(self, t):
        """this segment's rows (patches + CLS) of a [Mp, ...] buffer"""
        return t[self.row0:self.end]

    def lane(self, t):
        return t[self.lane0:self.lane0 + self.B]


class _Plan:
    """Activation and gradient buffers of one list of (B, T, N) clip shapes, allocated once and reused every step."""

    def __init__(self, shapes, D, Hd, H, depth, Kp, dev, res16=False):
        self.res16 = res16
        self.segs, row0, lane0 = [], 0, 0
        for (B, T, N) in shapes:
            self.segs.append(_Seg(B, T, N, row0, lane0, D, H, Kp, dev))
            row0 += self.segs[-1].M
            lane0 += B
        self.M, self.Bsum = row0, lane0
        self.Mp = _round_up(self.M, 256)
        Mp = self.Mp
        z16 = lambda c: torch.zeros(Mp, c, dtype=torch.bfloat16, device=dev)
        self.blocks = [_BlockActs(Mp, D, Hd, H, dev, res16) for _ in range(depth)]
        self.x0 = torch.zeros(Mp, D, dtype=torch.float32, device=dev)
        self.cls0 = torch.zeros(D, dtype=torch.float32, device=dev)
        self.fstats = torch.zeros(2, Mp, dtype=torch.float32, device=dev)
        self.normed = torch.zeros(Mp, D, dtype=torch.float32, device=dev)
        self.region = None                      # [Mp, D] fp32, allocated on first use (region_mem variant)
        self.rstats = None
        self.branch16 = z16(D)                  # forward: bf16 branch output awaiting its fused add + LayerNorm
        # backward temporaries.  Weight gradients read their dY from small rings so the chain can run ahead:
        #   ga[3]  : block-input gradient (written one block ahead by LN3-backward)
        #   sets[2]: d_h, dy16, d_qkv_s, dxt16, d_qkv_t of even / odd blocks
        self.G = torch.zeros(Mp, D, dtype=torch.float32, device=dev)
        self.ga = [z16(D) for _ in range(3)]
        self.sets = [dict(d_h=z16(Hd), gb=z16(D), d_qkv_s=z16(3 * D), gc=z16(D), d_qkv_t=z16(3 * D)) for _ in range(2)]
        self.d_a = z16(D)
        self.d_o = z16(D)
        self.dx2_16 = z16(D)                    # folded LayerNorms: norm2's dx of the current block (added to G by norm3's backward)
        self.side = self.x_final = None                   # set per call
        self.tape_fwd = self.tape_bwd = None              # (key, tape id, outputs, segments)
        self.wq = None                                    # queue of weight-gradient problems (grouped mode, engine._wgrad)
        self.tn_groups = {}                               # (block, group) -> (key, hip.TnGroup)
        self.fold_tabs = {}                               # block -> (key, hip.FoldGradTable)
        self.fold_pending = None
        self.dn = torch.zeros(Mp, D, dtype=torch.float32, device=dev)     # static copy of the output gradients
        self.d_region = None                              # [Mp, D] static copy of the region-token gradients
        self.lane = None                                  # fp32 buffers of the precise CLS lane (VideoEngine.forward)


class _Run:
    """What a forward leaves for its backward: the plan and the output flags."""

    def __init__(self, pl, need_patches, region_layer):
        self.pl, self.need_patches, self.region_layer = pl, need_patches, region_layer
        self.G = pl.G

    @property
    def regions(self):
        """region tokens per segment: [B*T*N, D] views"""
        return [self.pl.region[sg.row0:sg.cls0] for sg in self.pl.segs]

    @property
    def region(self):
        return self.regions[0]

    @property
    def blocks(self):
        return self.pl.blocks


class VideoEngine:
    """Owns bf16 weight shadows, per-shape plans and the launch schedules.

    `params` maps the reference's state_dict names (without the `video_model.` prefix) to fp32
    CUDA tensors; `grads` maps the same names to fp32 gradient buffers the backward WRITES
    (overwrite semantics: the reference zeroes grads every step, trainer_dist.py:156)."""

    LINEARS = ("attn.qkv", "attn.proj", "timeattn.qkv", "timeattn.proj", "mlp.fc1", "mlp.fc2")
    FOLDED = {"timeattn.qkv": "norm3", "attn.qkv": "norm1", "mlp.fc1": "norm2"}     # linear <- the LayerNorm folded into it

    def res16_active(self):
        """The residual stream (and the residual-gradient stream of backward) STORED as bf16 (the default; OAT_RES16=0 / res16 = False:
        fp32, the round-3 kernels).  What it costs in parity was measured in the CPU oracle first (scripts/dev/rounding_study3.py):
        sim-matrix error unchanged (it comes from the fp32 CLS lane), gradients 1.8e-2 -> 2.2e-2 relative L2."""
        return bool(self.res16)

    def __init__(self, depth, embed_dim, num_heads, mlp_ratio, patch_size, in_chans, num_frames):
        self.depth, self.D, self.H = depth, embed_dim, num_heads
        self.res16 = os.environ.get("OAT_RES16", "1") != "0"
        self.Hd = int(embed_dim * mlp_ratio)
        self.ps, self.C, self.num_frames = patch_size, in_chans, num_frames
        self.Kp = in_chans * patch_size * patch_size
        self.scale = (embed_dim // num_heads) ** -0.5
        if embed_dim // num_heads != 64:
            raise hip.OatError("the HIP attention kernels are built for head_dim 64")
        self.plans = {}
        self.shadow = {}
        self.shadow_versions = None
        self._cast = None
        _g = os.environ.get("OAT_BWD_NT_GRID", "0")                          # "auto": parallel.GradSync derives it from the device when it is built
        self.bwd_nt_grid = 0 if _g == "auto" else int(_g, 0)                 # gemm_nt grid during backward (0 = as in forward, 0xffff = one workgroup per tile)
        self.cls_lane = os.environ.get("OAT_CLS_LANE", "1") != "0"      # fp32 lane for the CLS rows (see _lane_ln)
        # fp8 forward (BASELINE.json config 5): the six linears of every block run on OCP e4m3 operands with per-tensor
        # delayed scaling (csrc/fp8.hip, gemm_nt_pp.hip PPF_F8), on the bf16 residual stream (oat_layernorm_fwd_r16_f8); attention,
        # LayerNorm, the CLS lane, the loss and the whole backward stay bf16 / fp32.  Set by OAT_FP8=1 or by the caller
        # (bench.py --dtype fp8).  (fp8 data gradients - measured equal in time, gradient norms 10-12 % off - left in round 6.)
        self.fp8 = os.environ.get("OAT_FP8", "0") != "0"
        self.fp8_margin = 1.0
        # the saved GELU derivative as 8-bit fixed point where the ping-pong GEMM serves the MLP pair (gemm_nt_pp.hip HU8_*)
        self.h_u8 = os.environ.get("OAT_H_U8", "1") != "0"
        # launch tapes (csrc/tape.hip): forward and backward are recorded once per plan and replayed from C
        self.use_tape = os.environ.get("OAT_TAPE", "1") != "0"
        self._f8 = None
        self._streams = None
        self._tn_ws = None
        self._tn_retired = []
        self._tn_slabs = None               # fp32 partial-tile workspace shared by every grouped weight-gradient launch
        # Fixed parts of the schedule (each was a knob while it was being measured):
        #  * the six weight gradients of a block are queued and launched together at the block's end (csrc/gemm_tn_sk.hip);
        #  * norm3 / norm1 / norm2 are FOLDED into the linear layer that follows them (timeattn.qkv / attn.qkv / mlp.fc1): the
        #    shadows are W' = W diag(gamma), the bias b' = b + W beta, LayerNorm forward writes the plain normalised row xhat,
        #    LayerNorm backward reads the saved bf16 xhat + rstd instead of the fp32 input, and (dW, dgamma, dbeta) come out of dW'
        #    by oat_ln_fold_grads: the same function and the same gradients with 77 MB less per LayerNorm backward (rowops.hip);
        #  * y = x + space is never stored (the next block's norm3 adds both branch outputs);
        #  * the CLS-row gradients of the attention backward are written by the attention kernel itself (oat_attn_*_bwd_fin);
        #  * two clips in one plan (OA models) share their space-attention and TIME-backward launches.
        # 1 (default since round 6; OAT_PRUNE_TOP=0 = every launch of the reference's graph): when the caller consumes only the
        # CLS rows of the encoder output (contract class oa_model.FrozenInTime, video_transformer.py:349-351 -> oa_model.py:129-133)
        # the patch rows of the TOP block's space-attention projection, norm2, fc1 / GELU and fc2 are never read and their output
        # gradient is exactly zero, so these launches - forward, data gradient and weight gradient - run on the B CLS rows only.
        # Bit-identical loss, gradients equal to 4e-7 (the weight-gradient sums lose only exact-zero terms; tests/test_prune_gpu.py);
        # 49.9 of 1115.9 GF per pair at 8 frames are not executed (bench.py reports the executed figure AND the full graph's).
        # Also oa_model_region_mem (CLS rows + the block-6 region tap; both clips of its plan); never oa_model_global_local
        # (it averages the final patch rows).  See _top_tail_fwd / _top_block_bwd_pruned.
        self.prune_top = os.environ.get("OAT_PRUNE_TOP", "1") != "0"
        self.fbias = {}                     # folded biases b' (fp32), per folded linear
        self._fold_bias = None
        self._fold_tmp = {}                 # accumulate mode: scratch (dW', db') of the folded linears

    # ------------------------------------------------------------------ weights / plans / streams
    def refresh_shadows(self, params, sig=None):
        """bf16 W and W^T copies of every GEMM weight; re-cast only when a master changed
        (`sig` = EngineModule._weights_signature())."""
        names = [f"blocks.{i}.{l}.weight" for i in range(self.depth) for l in self.LINEARS]
        names.append("patch_embed.proj.weight")
        fold = True                              # norm3 / norm1 / norm2 are folded into the linear that follows them
        if sig is not None and sig == self.shadow_versions:
            return
        srcs = [params[n].detach().reshape(params[n].shape[0], -1) for n in names]
        key = tuple(w.data_ptr() for w in srcs)
        if self._cast is None or self._cast.key != key:           # masters moved (first call, .to(), flatten)
            entries, fb = [], []
            for n, w in zip(names, srcs):
                self.shadow[n] = (torch.empty_like(w, dtype=torch.bfloat16),
                                  torch.empty(w.shape[1], w.shape[0], dtype=torch.bfloat16, device=w.device))
                norm = self.FOLDED.get(n.split(".", 2)[2][:-len(".weight")]) if (fold and n.startswith("blocks.")) else None
                if norm is None:
                    entries.append((w, self.shadow[n][0], self.shadow[n][1], w.shape[1], w.shape[0]))
                else:                                             # W' = W diag(gamma of the LayerNorm in front), b' = b + W beta
                    blk = ".".join(n.split(".")[:2])                  # "blocks.<i>"
                    gamma, beta = params[f"{blk}.{norm}.weight"].detach(), params[f"{blk}.{norm}.bias"].detach()
                    entries.append((w, self.shadow[n][0], self.shadow[n][1], w.shape[1], w.shape[0], gamma))
                    bname = n[:-len("weight")] + "bias"
                    self.fbias[bname] = torch.empty(w.shape[0], dtype=torch.float32, device=w.device)
                    fb.append((w, beta, params[bname].detach(), self.fbias[bname]))
            self._cast = hip.CastTable(entries)
            self._cast.key = key
            self._fold_bias = hip.FoldBiasTable(fb) if fb else None
        self._cast.run()                                          # all 73 W / W^T shadows in one launch
        if self._fold_bias is not None:
            self._fold_bias.run()                                 # the 36 folded biases in one launch
        self.shadow_versions = sig
        if self.fp8:
            self._refresh_fp8_weights()

    # ------------------------------------------------------------------ fp8 forward
    F8_LINEARS = ("timeattn.qkv", "timeattn.proj", "attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")

    def _fp8_state(self, dev):
        """Quantisation sites of block i, linear j: weight 18 i + j (e4m3), forward input 18 i + 6 + j (e4m3); sites 18 i + 12 + j are
        unused since round 6 (they held the incoming gradients of the fp8 data-gradient mode).  amax / qscale / dq are rows of one
        device tensor."""
        if self._f8 is None:
            n = 18 * self.depth
            st = torch.zeros(3, n, dtype=torch.float32, device=dev)
            self._f8 = dict(n=n, amax=st[0], qscale=st[1], dq=st[2], w8={}, table=None, primed=set(), key=None)
        return self._f8

    def _refresh_fp8_weights(self):
        """e4m3 copies of the six weight matrices of every block, from their bf16 shadows, scaled by their CURRENT amax
        (3 launches for all 72: amax, scales, quantise)."""
        names = [(i, j, f"blocks.{i}.{l}.weight") for i in range(self.depth) for j, l in enumerate(self.F8_LINEARS)]
        dev = self.shadow[names[0][2]][0].device
        f8 = self._fp8_state(dev)
        key = tuple(self.shadow[n][0].data_ptr() for _, _, n in names)
        if f8["key"] != key:
            entries = []
            for i, j, n in names:
                w16 = self.shadow[n][0]
                f8["w8"][n] = torch.empty(w16.shape, dtype=torch.uint8, device=dev)
                entries.append((w16, f8["w8"][n], 18 * i + j))
            f8["table"], f8["key"] = hip.Fp8Table(entries), key
        f8["table"].run(f8["qscale"], f8["amax"], quant=False)
        hip.fp8_update_scales(f8["amax"], f8["qscale"], f8["dq"], f8["n"], 1.0)       # activation sites: amax == 0, untouched
        f8["table"].run(f8["qscale"], f8["amax"], quant=True)

    def _f8_site(self, i, j):
        """(qscale, amax, dq) one-element views of the forward-input site of linear j of block i."""
        f8, s = self._f8, 18 * i + 6 + j
        return f8["qscale"][s:s + 1], f8["amax"][s:s + 1], f8["dq"][s:s + 1]

    def _f8_primed(self, i, j):
        """Has this GEMM operand been seen (does its delayed scale exist)?  Producers quantise in their own epilogue only
        then; the very first step quantises in a separate pass with the CURRENT amax."""
        return self.fp8 and (18 * i + 6 + j) in self._f8["primed"]

    def _ln_f8(self, pl, i, j, x, gamma, beta, y, mean, rstd, add16=None, sum32=None):
        """LayerNorm whose output feeds linear j of block i: bf16 y (kept for backward) and its e4m3 copy in pl.x8."""
        q, am, _ = self._f8_site(i, j)
        hip.layernorm_fwd_f8(x, gamma, beta, pl.M, self.D, 1e-6, y, pl.x8, q, am, mean, rstd, add16=add16, sum32=sum32)

    def _ln_f8_kw(self, pl, i, j):
        """The fp8 arguments of oat_layernorm_fwd_r16_f8 for the LayerNorm in front of linear j of block i."""
        q, am, _ = self._f8_site(i, j)
        return dict(y8=pl.x8, qscale=q, amax=am)

    def _linear_f8(self, pl, i, j, x16, K, N, epi, out, bias, out2=None, quantised=False):
        """out = x16 @ W^T + bias on fp8 operands.  quantised=True: the producer already left e4m3(x16) in pl.x8 /
        pl.x8_wide; otherwise the bf16 input is quantised here (delayed scale; first use: current scale).  The fc1 launch
        (j == 4) also leaves e4m3(gelu) for fc2 in pl.x8_wide once fc2's site is primed."""
        f8 = self._f8
        site, wsite = 18 * i + 6 + j, 18 * i + j
        x8 = pl.x8_wide if K == self.Hd else pl.x8
        M = pl.M
        q, am, dq = self._f8_site(i, j)
        if not quantised:
            if site not in f8["primed"]:
                hip.fp8_amax(x16, M, K, am)
                hip.fp8_update_scales(am, q, dq, 1, self.fp8_margin)
                f8["primed"].add(site)
            hip.fp8_quant(x16, x8, M, K, q, am)
        w8 = f8["w8"][f"blocks.{i}.{self.F8_LINEARS[j]}.weight"]
        kw = {}
        if j == 4 and self._f8_primed(i, 5):
            q5, am5, _ = self._f8_site(i, 5)
            kw = dict(out8=pl.x8_wide, q_out=q5, amax_out=am5)
        hip.gemm_nt_f8(x8, w8, M, N, K, epi, out, dq, f8["dq"][wsite:wsite + 1], out2=out2, bias=bias, **kw)
        return bool(kw)

    def _fp8_end_of_forward(self):
        """Next step's activation scales from this step's amax (weight sites saw no amax: unchanged)."""
        f8 = self._f8
        hip.fp8_update_scales(f8["amax"], f8["qscale"], f8["dq"], f8["n"], self.fp8_margin)

    def plan(self, shapes, dev, call=0):
        """One plan per list of clip shapes AND per call of a step (each forward's activations must survive to its
        backward)."""
        res16 = self.res16_active()
        key = (tuple(shapes), str(dev), call, res16)
        if key not in self.plans:
            self.plans[key] = _Plan(shapes, self.D, self.Hd, self.H, self.depth, self.Kp, dev, res16)
        return self.plans[key]

    def _get_streams(self, dev):
        """Streams are created only when used: HIP multiplexes streams onto a few hardware queues
        (GPU_MAX_HW_QUEUES, default 4) and two streams that share a queue run in enqueue order."""
        if self._streams is None:
            self._streams = dict(side=hip.side_stream("lane", dev))
        return self._streams

    # ------------------------------------------------------------------ forward
    def forward(self, video, params, need_patches=False, sig=None, region_layer=None, call=0):
        """video [B,T,C,R,R] fp32|bf16 -> (cls_normed fp32 [B,D], patches_normed fp32 [B*T*N, D] | None, run).
        region_layer=K additionally leaves region_norm(x after block K)[patch rows] in run.region(s)
        (oa_video_transformer_region.py:364-376).
        video may be a LIST of clips (same C, R; any B, T): they are encoded together as segments of one row space and
        the first two results are lists (one entry per clip)."""
        many = isinstance(video, (list, tuple))
        clips = list(video) if many else [video]
        C, R = clips[0].shape[2], clips[0].shape[3]
        g = R // self.ps
        N = g * g
        shapes = []
        for v in clips:
            if v.shape[1] > self.num_frames:
                raise ValueError(f"{v.shape[1]} frames > num_frames={self.num_frames}")     # video_transformer.py:73
            if v.shape[2] != C or v.shape[3] != R:
                raise ValueError("clips encoded together must share channels and resolution")
            shapes.append((v.shape[0], v.shape[1], N))
        self.refresh_shadows(params, sig)
        dev = clips[0].device
        st = self._get_streams(dev)
        pl = self.plan(shapes, dev, call)
        pl.side = st["side"]
        # the clips are copied into plan-owned buffers: every launch of the schedule then has static arguments and the
        # schedule can be replayed from its tape (csrc/tape.hip)
        for sg, v in zip(pl.segs, clips):
            if sg.video_buf is None or sg.video_buf.dtype != v.dtype or sg.video_buf.shape != v.shape:
                sg.video_buf = torch.empty(v.shape, dtype=v.dtype, device=dev)
                pl.tape_fwd = None
            sg.video_buf.copy_(v)
            sg.video = sg.video_buf
        if self.cls_lane and pl.lane is None:
            z = lambda c: torch.zeros(pl.Bsum, c, dtype=torch.float32, device=dev)
            pl.lane = dict(x=z(self.D), xt=z(self.D), y=z(self.D), out=z(self.D), a32=z(self.D), q32=z(self.D), o32=z(self.D),
                           br32=z(self.D), g32=z(self.Hd))
        elif not self.cls_lane:
            pl.lane = None
        if self.fp8 and getattr(pl, "x8", None) is None:
            pl.x8 = torch.zeros(pl.Mp, self.D, dtype=torch.uint8, device=dev)          # fp8 GEMM inputs (one in flight)
            pl.x8_wide = torch.zeros(pl.Mp, self.Hd, dtype=torch.uint8, device=dev)
        if self.fp8 and not pl.res16:
            raise hip.OatError("the fp8 forward runs on the bf16 residual stream (res16): OAT_RES16=0 / res16 = False is the bf16 engine's fp32-stream option")
        run = _Run(pl, need_patches, region_layer)
        # (a region tap BELOW the top block - oa_model_region_mem: block 6 - leaves the top block's patch rows just as unused; a tap
        # on the encoder output reads them)
        pl.prune_top = bool(self.prune_top and not need_patches and region_layer != self.depth and pl.res16 and not self.fp8)
        run.prune_top = pl.prune_top
        if pl.prune_top and getattr(pl, "d_o_top", None) is None:
            # dL/d(attention output) of the top block: its patch rows are zero and stay zero (only the CLS rows are ever written)
            pl.d_o_top = torch.zeros(pl.Mp, self.D, dtype=torch.bfloat16, device=dev)
        pl.fwd_modes = (pl.res16, self.fp8)      # what the saved activations MEAN: backward checks it
        if getattr(pl, "branch16s", None) is None:     # the space branch keeps its own buffer until the next block's norm3
            pl.branch16s = torch.zeros(pl.Mp, self.D, dtype=torch.bfloat16, device=dev)
        pl.h_u8 = (self.h_u8 and pl.M >= 256 and self.Hd % 256 == 0 and self.Hd <= 4096 and self.D % 128 == 0 and self.D >= 128)
        if pl.h_u8 and pl.blocks[0].h8 is None:
            for a in pl.blocks:              # the bf16 buffer's first half, viewed as [Mp, Hd] bytes
                a.h8 = a.h.view(torch.uint8).view(-1)[:pl.Mp * self.Hd].view(pl.Mp, self.Hd)

        def body():
            self._embed(pl, params, C, R)
            pend = None
            for i in range(self.depth):
                pend = self._block_fwd(pl, i, params, pend, region_layer)
            out = self._final_fwd(pl, params, need_patches, region_layer)
            if self.fp8:
                self._fp8_end_of_forward()
            return out

        key = self._tape_key(pl, params, None, need_patches, region_layer, C, R)
        out = self._taped(pl, "tape_fwd", key, body)
        if many:
            return out[0], out[1], run
        return out[0][0], out[1][0], run

    # ------------------------------------------------------------------ launch tapes
    def _tape_key(self, pl, params, grads, *flags):
        """Everything a recorded schedule depends on besides the plan itself: the streams, where parameters / gradients
        live, the option flags and the state of the fp8 sites."""
        if not self.use_tape:
            return None
        ptrs = tuple(t.data_ptr() for t in params.values())
        gptr = next(iter(grads.values())).data_ptr() if grads else 0
        f8 = (len(self._f8["primed"]), self._f8["key"]) if (self.fp8 and self._f8) else None
        return (torch.cuda.current_stream().cuda_stream, pl.side.cuda_stream, ptrs, gptr, self.fp8, f8, self.cls_lane, self.h_u8,
                self.bwd_nt_grid, pl.res16, getattr(pl, "prune_top", False), flags)

    @staticmethod
    def _announce_segment(ready, prefixes, recording):
        """End of a backward segment while the schedule runs live: close the tape segment, then let the host callback
        (gradient all-reduce / eager optimiser) issue its own work unrecorded."""
        if recording:
            hip.lib().oat_tape_mark()
            hip.lib().oat_tape_pause(1)
        try:
            ready(prefixes)
        finally:
            if recording:
                hip.lib().oat_tape_pause(0)

    def _taped(self, pl, slot, key, body, segments=None):
        """Run `body` (a launch schedule without host-side data dependence) - the first time with the tape recording,
        afterwards by replaying the tape.  segments: callback(k) run after segment k when the schedule was recorded
        with marks (the gradient announcements of backward)."""
        if key is None:
            return body()
        held = getattr(pl, slot, None)
        if held is not None and held[0] == key:
            _, tid, out, nseg = held
            if segments is None:
                hip.tape_replay(tid)
            else:
                for k in range(nseg):
                    hip.tape_replay(tid, k, k + 1)
                    segments(k)
            return out
        if held is not None:
            hip.tape_free(held[1])
            setattr(pl, slot, None)
        hip.tape_begin()
        try:
            out = body()
        except BaseException:
            hip.tape_abort()
            self._reset_tickets(pl)
            raise
        tid = hip.tape_end()
        setattr(pl, slot, (key, tid, out, hip.lib().oat_tape_segments(tid)))
        return out

    @staticmethod
    def _reset_tickets(pl):
        """After a schedule that did not run to its end (a launch raised, a recording was aborted): the device-side state that
        kernels hand from launch to launch - the CLS-row sums and tickets of the fused attention finalize, the partial sums and
        tickets of oat_ln_fold_grads - is put back to zero, the state every complete launch leaves behind.  Without this a
        half-run backward would make every later one compute wrong CLS / dgamma / dbeta gradients without any error."""
        try:
            torch.cuda.synchronize()
            for sg in pl.segs:
                sg.cls_side.zero_()
                sg.cls_done.zero_()
            for _, tab in pl.fold_tabs.values():
                tab.work.zero_()
        except Exception:                # the device itself may be gone; the original exception is the one to report
            pass

    def _embed(self, pl, params, C, R):
        D = self.D
        for sg in pl.segs:
            hip.im2col(sg.video, sg.cols, sg.B * sg.T, C, R, self.ps)
            hip.pos_table(params["pos_embed"], params["temporal_embed"], params["cls_token"], sg.table, pl.cls0, sg.T, sg.N, D)
            hip.gemm_nt(sg.cols, self.shadow["patch_embed.proj.weight"][0], sg.BTN, D, self.Kp, hip.EPI_F32, pl.x0[sg.row0:],
                        bias=params["patch_embed.proj.bias"], resid=sg.table, resid_mod=sg.T * sg.N)
            hip.broadcast_rows(pl.cls0, pl.x0[sg.cls0:], sg.B, D)

    # ---- the precise CLS lane ------------------------------------------------------------------------------------
    # The cosine-similarity matrix must match the fp32 reference within 1e-3.  Only the CLS row of a clip reaches the
    # embedding, and its OWN rounding errors (bf16 operands of its q / proj / fc1 / fc2 products, bf16 branch outputs)
    # do not average out - those of the ~1.5 k patch keys it attends do.  So the B CLS rows are computed a second time
    # by an fp32 lane on the side stream: its own fp32 residual rows -> LayerNorm -> q = oat_linear_f32(master weights)
    # -> the precise query of oat_attn_cls_fwd_dual against the main path's bf16 keys / values -> proj / fc1 / GELU /
    # fc2 in fp32.  The lane only CONSUMES the main path (q|k|v buffers) and delivers the final CLS embedding; the
    # bf16 main path - which backward differentiates - neither waits for it nor reads it, so the lane may lag behind.
    # ~10 launches of a few us per block; video embedding error 6.7e-3 -> 1-3e-3, sim-matrix error 0.7-1.7e-3 ->
    # 1-3e-4 (tests/test_model_gpu.py; rounding emulation in the CPU oracle: scripts/dev/rounding_study2.py).
    def _lane_ln(self, pl, x, add32, sum32, gamma, beta, y32):
        """side stream: [sum32 = x + add32 ;] y32 = LayerNorm(.)  on the lane's B fp32 rows"""
        with torch.cuda.stream(pl.side):
            if add32 is None:
                hip.layernorm_fwd(x, gamma, beta, pl.Bsum, self.D, 1e-6, y32=y32)
            else:
                hip.add32_layernorm_fwd(x, add32, sum32, gamma, beta, pl.Bsum, self.D, 1e-6, y32=y32)

    def _lane_linear(self, pl, A, W, bias, N, K, out32, act=0):
        with torch.cuda.stream(pl.side):
            # LIN_EXACT: the lane is what holds the 1e-3 sim-matrix bound - f32 products also when more than 64 clips share a plan
            hip.linear_f32(A, W, pl.Bsum, N, K, bias=bias, out32=out32, act=act | hip.LIN_EXACT)

    def _block_fwd(self, pl, i, params, pend, region_layer):
        """Residual adds are fused into the NEXT LayerNorm: the projection / fc2 GEMMs write their branch output as bf16 and
        the streaming LN kernel forms x + branch (in fp32), stores the new stream where one is needed and the normalised bf16
        operand in one pass.  `pend` = the previous block, whose out = x + space + mlp is still to be formed.  The LayerNorms
        are folded into the linear that follows them: the kernels write the plain normalised row (gamma = beta = None), the
        linear layers take W' = W diag(gamma) and the folded bias."""
        M, D, Hd = pl.M, self.D, self.Hd
        a, br, brs = pl.blocks[i], pl.branch16, pl.branch16s
        p = lambda s: params[f"blocks.{i}.{s}"]
        w = lambda s: self.shadow[f"blocks.{i}.{s}.weight"][0]
        st = a.stats
        lane = pl.lane
        lin_b = lambda n: self.fbias[f"blocks.{i}.{n}.bias"]
        f8 = self.fp8
        q3 = self._f8_primed(i, 0)          # fp8: LayerNorm outputs are quantised by the LayerNorm kernel itself
        if pend is None:
            x = pl.x0
            if q3:
                self._ln_f8(pl, i, 0, x, None, None, a.a3, st[0], st[1])
            else:
                hip.layernorm_fwd(x, None, None, M, D, 1e-6, y=a.a3, mean=st[0], rstd=st[1])
            if lane is not None:                         # the lane starts from the embedding's CLS rows
                hip.stream_edge(torch.cuda.current_stream(), pl.side)
                with torch.cuda.stream(pl.side):
                    for sg in pl.segs:
                        hip.copy_(sg.lane(lane["x"]), x[sg.cls0:sg.end])
                self._lane_ln(pl, lane["x"], None, None, p("norm3.weight"), p("norm3.bias"), lane["a32"])
        else:
            # out = x + space + mlp of the previous block in one pass (its y = x + space was never stored)
            if pl.res16:             # bf16 stream: out16 = bf16(x + space + mlp), a3 = LN of the unrounded sum
                hip.layernorm_fwd_r16(pend.xin, M, D, 1e-6, add_a=brs, add_b=br, sum16=pend.out, y=a.a3, mean=st[0], rstd=st[1],
                                      **(self._ln_f8_kw(pl, i, 0) if q3 else {}))
            else:
                hip.add2_layernorm_fwd(pend.xin, brs, br, pend.out, None, None, M, D, 1e-6, y=a.a3, mean=st[0], rstd=st[1])
            x = pend.out
            if lane is not None:                         # x = y + mlp of the previous block
                self._lane_ln(pl, lane["y"], lane["br32"], lane["x"], p("norm3.weight"), p("norm3.bias"), lane["a32"])
            if region_layer is not None and i == region_layer:
                self._region_tap(pl, params, x)
        # ---- time attention
        if lane is not None:
            self._lane_linear(pl, lane["a32"], p("timeattn.qkv.weight")[:D], p("timeattn.qkv.bias")[:D], D, D, lane["q32"])
        if f8:
            self._linear_f8(pl, i, 0, a.a3, D, 3 * D, hip.EPI_BF16, a.qkv_t, lin_b("timeattn.qkv"), quantised=q3)
        else:
            hip.gemm_nt(a.a3, w("timeattn.qkv"), M, 3 * D, D, hip.EPI_BF16, a.qkv_t, bias=lin_b("timeattn.qkv"))
        self._attention(pl, hip.attn_time_fwd, a.qkv_t, a.o_t, a.lse_t)
        if lane is not None:
            self._lane_linear(pl, lane["o32"], p("timeattn.proj.weight"), p("timeattn.proj.bias"), D, D, lane["br32"])
            self._lane_ln(pl, lane["x"], lane["br32"], lane["xt"], p("norm1.weight"), p("norm1.bias"), lane["a32"])
            self._lane_linear(pl, lane["a32"], p("attn.qkv.weight")[:D], p("attn.qkv.bias")[:D], D, D, lane["q32"])
        if f8:
            self._linear_f8(pl, i, 1, a.o_t, D, D, hip.EPI_BF16, br, p("timeattn.proj.bias"))
        else:
            hip.gemm_nt(a.o_t, w("timeattn.proj"), M, D, D, hip.EPI_BF16, br, bias=p("timeattn.proj.bias"))
        # xt = x + time feeds norm1 only (the space residual comes from x): it is never stored
        q1 = self._f8_primed(i, 2)
        if pl.res16:
            hip.layernorm_fwd_r16(x, M, D, 1e-6, add_a=br, y=a.a1, mean=st[2], rstd=st[3], **(self._ln_f8_kw(pl, i, 2) if q1 else {}))
        else:
            hip.add_layernorm_fwd(x, br, None, None, None, M, D, 1e-6, y=a.a1, mean=st[2], rstd=st[3])
        # ---- space attention
        if f8:
            self._linear_f8(pl, i, 2, a.a1, D, 3 * D, hip.EPI_BF16, a.qkv_s, lin_b("attn.qkv"), quantised=q1)
        else:
            hip.gemm_nt(a.a1, w("attn.qkv"), M, 3 * D, D, hip.EPI_BF16, a.qkv_s, bias=lin_b("attn.qkv"))
        top_pruned = pl.prune_top and i == self.depth - 1
        if top_pruned:
            # only the CLS query's output is consumed: the patch queries are not run.  Backward still walks them (their dO is
            # zero) and needs P = exp2(s - lse) = 0 there whatever the stale rows hold: lse = 3.4e38 (bytes 0x7f; the kernel
            # scales it by log2(e) to +inf - the form it gives padding queries itself, attn_space.hip)
            for sg in pl.segs:
                hip.fill_bytes_(a.lse_s[sg.row0:sg.cls0], 0x7f)
        self._attention(pl, hip.attn_space_fwd, a.qkv_s, a.o_s, a.lse_s, patch=not top_pruned)
        if lane is not None:
            self._lane_linear(pl, lane["o32"], p("attn.proj.weight"), p("attn.proj.bias"), D, D, lane["br32"])
            # space residual comes from x, NOT from x + time (video_transformer.py:170)
            self._lane_ln(pl, lane["x"], lane["br32"], lane["y"], p("norm2.weight"), p("norm2.bias"), lane["a32"])
            self._lane_linear(pl, lane["a32"], p("mlp.fc1.weight"), p("mlp.fc1.bias"), Hd, D, lane["g32"], act=hip.LIN_GELU)
            self._lane_linear(pl, lane["g32"], p("mlp.fc2.weight"), p("mlp.fc2.bias"), D, Hd, lane["br32"])
        a.xin = x
        if top_pruned:
            self._top_tail_fwd(pl, a, x, brs, br, p, w, lin_b)
            return a
        if f8:
            self._linear_f8(pl, i, 3, a.o_s, D, D, hip.EPI_BF16, brs, p("attn.proj.bias"))
        else:
            hip.gemm_nt(a.o_s, w("attn.proj"), M, D, D, hip.EPI_BF16, brs, bias=p("attn.proj.bias"))
        # space residual comes from x, NOT from x + time (video_transformer.py:170); y = x + space is not stored
        q2 = self._f8_primed(i, 4)
        if pl.res16:
            hip.layernorm_fwd_r16(x, M, D, 1e-6, add_a=brs, y=a.a2, mean=st[4], rstd=st[5], **(self._ln_f8_kw(pl, i, 4) if q2 else {}))
        else:
            hip.add_layernorm_fwd(x, brs, None, None, None, M, D, 1e-6, y=a.a2, mean=st[4], rstd=st[5])
        # ---- MLP
        if f8:
            gq = self._linear_f8(pl, i, 4, a.a2, D, Hd, hip.EPI_GELU_GRAD | (hip.EPI_U8 if pl.h_u8 else 0), a.h8 if pl.h_u8 else a.h,
                                 lin_b("mlp.fc1"), out2=a.g, quantised=q2)
            self._linear_f8(pl, i, 5, a.g, Hd, D, hip.EPI_BF16, br, p("mlp.fc2.bias"), quantised=gq)
        else:
            if pl.h_u8:
                hip.gemm_nt(a.a2, w("mlp.fc1"), M, Hd, D, hip.EPI_GELU_GRAD | hip.EPI_U8, a.h8, out2=a.g, bias=lin_b("mlp.fc1"))
            else:
                hip.gemm_nt(a.a2, w("mlp.fc1"), M, Hd, D, hip.EPI_GELU_GRAD, a.h, out2=a.g, bias=lin_b("mlp.fc1"))
            hip.gemm_nt(a.g, w("mlp.fc2"), M, D, Hd, hip.EPI_BF16, br, bias=p("mlp.fc2.bias"))
        return a                                                                    # out = x + brs + br, formed lazily

    def _top_tail_fwd(self, pl, a, x, brs, br, p, w, lin_b):
        """prune_top: the top block from the space-attention projection on, for the B CLS rows only (the tail block of the
        row space).  The patch rows of brs / a.a2 / a.h / a.g / br keep whatever they held: nothing reads them - the final
        LayerNorm takes the CLS rows (_final_fwd), backward runs _top_block_bwd_pruned.  The saved GELU derivative of these
        rows is plain bf16 in a.h (the 8-bit blocked form belongs to the ping-pong GEMM, which does not serve 32-row problems)."""
        D, Hd, st = self.D, self.Hd, a.stats
        for sg in pl.segs:                 # every clip's CLS rows are one tail block of its row range
            c0, Bc = sg.cls0, sg.B
            hip.gemm_nt(a.o_s[c0:], w("attn.proj"), Bc, D, D, hip.EPI_BF16, brs[c0:], bias=p("attn.proj.bias"))
            hip.layernorm_fwd_r16(x[c0:], Bc, D, 1e-6, add_a=brs[c0:], y=a.a2[c0:], mean=st[4][c0:], rstd=st[5][c0:])
            hip.gemm_nt(a.a2[c0:], w("mlp.fc1"), Bc, Hd, D, hip.EPI_GELU_GRAD, a.h[c0:], out2=a.g[c0:], bias=lin_b("mlp.fc1"))
            hip.gemm_nt(a.g[c0:], w("mlp.fc2"), Bc, D, Hd, hip.EPI_BF16, br[c0:], bias=p("mlp.fc2.bias"))

    def _final_fwd(self, pl, params, need_patches, region_layer):
        """-> ([cls rows per segment], [patch rows per segment] | [None, ...])"""
        M, D = pl.M, self.D
        last, br = pl.blocks[-1], pl.branch16
        pl.x_final = last.out
        lane = pl.lane
        g, bt = params["norm.weight"], params["norm.bias"]
        tap_last = region_layer is not None and region_layer == self.depth
        def final_ln(r0, rows):
            kw = dict(y32=pl.normed[r0:], mean=pl.fstats[0][r0:], rstd=pl.fstats[1][r0:])
            if pl.res16:
                hip.layernorm_fwd_r16(last.xin[r0:], rows, D, 1e-6, add_a=pl.branch16s[r0:], add_b=br[r0:], sum16=last.out[r0:],
                                      gamma=g, beta=bt, **kw)
            else:
                hip.add2_layernorm_fwd(last.xin[r0:], pl.branch16s[r0:], br[r0:], last.out[r0:], g, bt, rows, D, 1e-6, **kw)

        if need_patches or tap_last:
            final_ln(0, M)
            if tap_last:
                self._region_tap(pl, params, last.out)
        else:
            # contract class: only the CLS rows of the last block's output are ever consumed
            for sg in pl.segs:
                final_ln(sg.cls0, sg.B)
        cls_out = [pl.normed[sg.cls0:sg.end] for sg in pl.segs]
        if lane is not None:                              # the CLS embedding comes from the lane's fp32 rows
            self._lane_ln(pl, lane["y"], lane["br32"], lane["x"], g, bt, lane["out"])
            hip.stream_edge(pl.side, torch.cuda.current_stream())
            cls_out = [sg.lane(lane["out"]) for sg in pl.segs]
        return cls_out, [pl.normed[sg.row0:sg.cls0] if need_patches else None for sg in pl.segs]

    def _attention(self, pl, patch_kernel, qkv, out, lse, patch=True):
        """Patch attention on the caller's stream, the independent CLS-query attention (it only writes the
        CLS rows of out / lse, and the lane's precise context) concurrently on the side stream.  patch=False (pruned top
        block): the CLS query alone."""
        cur = torch.cuda.current_stream()
        hip.stream_edge(cur, pl.side)                    # qkv is complete
        with torch.cuda.stream(pl.side):
            for sg in pl.segs:                           # attention is per clip: each segment is a self-contained row range
                q, o, l = sg.rows(qkv), sg.rows(out), sg.rows(lse)
                if pl.lane is not None:
                    hip.attn_cls_fwd_dual(q, o, l, sg.lane(pl.lane["q32"]), sg.lane(pl.lane["o32"]), sg.B, sg.T, sg.N, self.H,
                                          self.D, self.scale)
                else:
                    hip.attn_cls_fwd(q, o, l, sg.B, sg.T, sg.N, self.H, self.D, self.scale)
        if not patch:
            pass
        elif patch_kernel is hip.attn_space_fwd and len(pl.segs) == 2 and pl.segs[0].N == pl.segs[1].N:
            # both clips' frames in ONE launch: the object frame alone is B x H problems, a third of the GPU
            hip.attn_space_fwd_clips([dict(qkv=sg.rows(qkv), out=sg.rows(out), lse=sg.rows(lse), B=sg.B, T=sg.T) for sg in pl.segs],
                                     pl.segs[0].N, self.H, self.D, self.scale)
        else:
            for sg in pl.segs:
                patch_kernel(sg.rows(qkv), sg.rows(out), sg.rows(lse), sg.B, sg.T, sg.N, self.H, self.D, self.scale)
        hip.stream_edge(pl.side, cur)

    def _region_tap(self, pl, params, x):
        """region_norm(x after block K)[patch rows] (oa_video_transformer_region.py:364-376)."""
        M, D = pl.M, self.D
        if pl.region is None:                      # all rows (row-wise kernel; the few CLS rows in between are never read)
            pl.region = torch.zeros(pl.Mp, D, dtype=torch.float32, device=x.device)
            pl.rstats = torch.zeros(2, pl.Mp, dtype=torch.float32, device=x.device)
        ln = hip.layernorm_fwd_r16 if x.dtype == torch.bfloat16 else hip.layernorm_fwd
        kw = dict(gamma=params["region_norm.weight"], beta=params["region_norm.bias"])
        ln(x, M=M, D=D, eps=1e-6, y32=pl.region, mean=pl.rstats[0], rstd=pl.rstats[1], **kw)

    # ------------------------------------------------------------------ backward
    def backward(self, run, params, grads, d_cls, d_patches=None, d_region=None, ready=None, accumulate=False):
        """Writes every video parameter gradient into `grads` (accumulate=True: ADDS to them - the second and later
        backward of a step in which the encoder ran more than once).  d_cls fp32 [B,D]; d_patches fp32 [B*T*N, D] or
        None (contract class oa_model.FrozenInTime discards patch outputs); d_region fp32 [B*T*N, D] =
        gradient of run.region (enters the residual stream below block `region_layer`).

        Schedule (see the module docstring): every kernel in order on the caller's stream; a block's six weight gradients
        as one grouped launch at the block's end.

        `ready(prefixes)` (optional) is called - on the caller's stream, after everything that writes them - as soon
        as all gradients of the parameters named by `prefixes` are enqueued: one call per block, top to bottom, so
        the gradient all-reduce can start while backward is still running."""
        pl = run.pl
        now = (pl.res16, self.fp8)
        if getattr(pl, "fwd_modes", now) != now:
            raise hip.OatError(f"engine options changed between a forward and its backward (res16, fp8): {pl.fwd_modes} -> {now}; "
                               "the saved activations would be misread")
        pl.acc = bool(accumulate)
        if run.region_layer is not None:
            ready = None                     # region_norm gradients arrive out of block order: reduce after backward
        # The incoming gradients go into plan-owned buffers (static launch arguments: the schedule replays from its tape).
        # One entry per segment (a bare tensor = the only segment's); None = that output received no gradient.
        as_list = lambda v: list(v) if isinstance(v, (list, tuple)) else [v]
        d_cls, d_patches, d_region = as_list(d_cls), as_list(d_patches), as_list(d_region)
        nseg = len(pl.segs)
        d_patches = d_patches if len(d_patches) == nseg else [None] * nseg
        d_region = d_region if len(d_region) == nseg else [None] * nseg
        have_patches = run.need_patches and any(t is not None for t in d_patches)
        have_region = run.region_layer is not None and any(t is not None for t in d_region)
        if have_region and pl.d_region is None:
            pl.d_region = torch.zeros(pl.Mp, self.D, dtype=torch.float32, device=run.G.device)
        for sg, dc, dp, dr in zip(pl.segs, d_cls, d_patches, d_region):
            if dc is None:
                pl.dn[sg.cls0:sg.end].zero_()
            else:
                pl.dn[sg.cls0:sg.end].copy_(dc)
            if have_patches:
                if dp is None:
                    pl.dn[sg.row0:sg.cls0].zero_()
                else:
                    pl.dn[sg.row0:sg.cls0].copy_(dp)
            if have_region:
                if dr is None:
                    pl.d_region[sg.row0:sg.cls0].zero_()
                else:
                    pl.d_region[sg.row0:sg.cls0].copy_(dr)
        d_region = pl.d_region if have_region else None
        prefixes = [(f"blocks.{i}.", "norm.") if i == self.depth - 1 else (f"blocks.{i}.",) for i in reversed(range(self.depth))]
        prefixes.append(("cls_token", "pos_embed", "temporal_embed", "patch_embed."))
        use_marks = ready is not None

        def body():
            for sg in pl.segs:
                hip.zero_(sg.cls_side)   # once per backward; every attn_cls_finalize leaves it zero for the next one
            self._final_bwd(pl, run, params, grads, have_patches, d_region)
            # the data-gradient GEMMs of a multi-rank job leave room for RCCL's kernels (parallel.GradSync sets bwd_nt_grid): a per-call
            # argument of oat_gemm_nt, not a mode of the library
            pl.bwd_grid = self.bwd_nt_grid
            for k, i in enumerate(reversed(range(self.depth))):
                pl.wq = []
                if getattr(run, "prune_top", False) and i == self.depth - 1:
                    self._top_block_bwd_pruned(pl, i, params, grads)
                else:
                    self._block_bwd(pl, i, run, params, grads, d_region)
                self._flush_wgrads(pl, i)
                pl.wq = None
                if getattr(pl, "fold_pending", None) == i:
                    self._fold_grads(pl, i, params, grads)
                    pl.fold_pending = None
                if use_marks:
                    self._announce_segment(ready, prefixes[k], recording)
            self._embed_bwd(pl, grads)
            if use_marks:
                self._announce_segment(ready, prefixes[-1], recording)

        key = self._tape_key(pl, params, grads, "bwd", run.need_patches, run.region_layer, have_patches, d_region is not None,
                             pl.acc, use_marks)
        recording = key is not None
        self._taped(pl, "tape_bwd", key, body, segments=(lambda k: ready(prefixes[k])) if use_marks else None)

    def _wgrad(self, P, Q, rows, n1, n2, w, b, acc=False, pl=None):
        """One weight gradient w (+)= P[:rows]^T Q[:rows], b (+)= colsum(P).  With a plan whose queue is open (`pl.wq`,
        grouped mode) the problem is only QUEUED: _flush_wgrads launches a block's six together (csrc/gemm_tn_sk.hip)."""
        if pl is not None and pl.wq is not None and n1 % 256 == 0 and n2 % 256 == 0:
            pl.wq.append((P, Q, rows, n1, n2, w, b, bool(acc)))
            return
        need = hip.lib().oat_gemm_tn_workspace_bytes(rows, n1, n2, 0) // 4     # exact for this (rows, shape)
        if self._tn_ws is None or self._tn_ws.numel() < need:
            if self._tn_ws is not None:
                self._tn_retired.append(self._tn_ws)       # launch tapes recorded so far still point at it
            self._tn_ws = torch.empty(need, dtype=torch.float32, device=P.device)   # one slab workspace, grown on demand
        hip.gemm_tn(P, Q, rows, n1, n2, w, bias_out=b, ws=self._tn_ws, accumulate=acc)

    def _flush_wgrads(self, pl, tag):
        """Launch the queued weight gradients of one block as ONE grouped launch + one fix-up (csrc/gemm_tn_sk.hip).
        The problems are layered by size: big ones share a layer as long as their output tiles leave every tile >= 2 splits
        over M on the CUs (ViT-B: {fc2, fc1, qkv, qkv} = 126 tiles x 2 splits), the small ones form the next layer
        ({proj, proj}: 18 tiles x 14 splits); every workgroup walks one segment of each layer.  Per block: 1 + 1 launches
        and 126 MB of fp32 partial tiles instead of 6 + 6 and 387 MB.  The group is cached per (plan, block): its tables
        hold raw pointers of plan-owned buffers and of the flat gradient buffer."""
        items, pl.wq = pl.wq, []
        if not items:
            return
        grid = torch.cuda.get_device_properties(items[0][0].device).multi_processor_count
        order = sorted(range(len(items)), key=lambda k: -(items[k][3] // 256) * (items[k][4] // 256))
        layers, cur, cur_tiles = [], [], 0
        for k in order:
            t = (items[k][3] // 256) * (items[k][4] // 256)
            if cur and cur_tiles + t > grid // 2:
                layers.append(cur)
                cur, cur_tiles = [], 0
            cur.append(k)
            cur_tiles += t
        if cur:
            layers.append(cur)
        probs = [items[k] for ks in layers for k in ks]
        key = tuple((P.data_ptr(), Q.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else 0, rows, acc)
                    for P, Q, rows, n1, n2, w, b, acc in probs)
        held = pl.tn_groups.get(tag)
        if held is None or held[0] != key:
            idx, pos = [], 0
            for ks in layers:
                idx.append(list(range(pos, pos + len(ks))))
                pos += len(ks)
            meta = [(rows, n1, n2) for _, _, rows, n1, n2, _, _, _ in probs]
            need = hip.lib().oat_tn_group_slab_bytes(hip.TnGroup.plan_layers([[meta[k] for k in ks] for ks in idx], grid)[3]) // 4
            if self._tn_slabs is None or self._tn_slabs.numel() < need:
                if self._tn_slabs is not None:
                    # The outgrown workspace stays alive and the groups built on it KEEP it (their tables hold its address and it
                    # is large enough for them; groups run one after the other on one stream, so which workspace each uses does
                    # not matter).  Until round 4 the plan's groups were dropped here and rebuilt on the new workspace: with a
                    # pruned top block - three problems, flushed first, smaller need than the six of the block below - that freed
                    # the tables of a group the tape being recorded had just captured (memory fault on replay), and left the
                    # group to be rebuilt (a host-to-device copy) inside a later hipGraph capture.
                    self._tn_retired.append(self._tn_slabs)
                self._tn_slabs = torch.empty(need, dtype=torch.float32, device=probs[0][0].device)
            held = (key, hip.TnGroup(probs, grid=grid, slabs=self._tn_slabs, layers=idx))
            pl.tn_groups[tag] = held
        held[1].run()

    def _fold_grads(self, pl, i, params, grads):
        """dW' / db' of the three folded linear layers of block i -> dW, db, dgamma, dbeta (one launch, oat_ln_fold_grads)."""
        ents = []
        for lin, norm in self.FOLDED.items():
            gw, gb = grads[f"blocks.{i}.{lin}.weight"], grads[f"blocks.{i}.{lin}.bias"]
            src_w, src_b = self._fold_tmp[lin] if pl.acc else (gw, gb)
            ents.append((src_w, src_b, params[f"blocks.{i}.{lin}.weight"], params[f"blocks.{i}.{norm}.weight"],
                         params[f"blocks.{i}.{norm}.bias"], gw, gb, grads[f"blocks.{i}.{norm}.weight"],
                         grads[f"blocks.{i}.{norm}.bias"], pl.acc))
        key = tuple(t.data_ptr() for e in ents for t in e[:9]) + (pl.acc,)
        held = pl.fold_tabs.get(i)
        if held is None or held[0] != key:
            held = (key, hip.FoldGradTable(ents))
            pl.fold_tabs[i] = held
        held[1].run()

    def _final_bwd(self, pl, run, params, grads, have_patches, d_region):
        """pl.dn holds dL/d(normed output) in the plan's row layout (patch rows are only valid with have_patches)."""
        D = self.D
        M = pl.M
        G = pl.G
        g16 = pl.ga[(self.depth - 1) % 3]
        if pl.res16:                 # the gradient stream is bf16 (g16 alone; G is written once, by block 0, for the embedding)
            if have_patches:
                hip.layernorm_bwd_r16(pl.dn, pl.x_final, pl.fstats[0], pl.fstats[1], params["norm.weight"], M, D, dx16=g16,
                                      dgamma=grads["norm.weight"], dbeta=grads["norm.bias"], accumulate=pl.acc)
            else:
                if not getattr(run, "prune_top", False):     # pruned top block: reads the CLS rows of g16 only
                    hip.zero_(g16[:M])
                for k, sg in enumerate(pl.segs):
                    c0 = sg.cls0
                    hip.layernorm_bwd_r16(pl.dn[c0:], pl.x_final[c0:], pl.fstats[0][c0:], pl.fstats[1][c0:], params["norm.weight"],
                                          sg.B, D, dx16=g16[c0:], dgamma=grads["norm.weight"], dbeta=grads["norm.bias"],
                                          accumulate=pl.acc or k > 0)
        elif have_patches:
            hip.layernorm_bwd(pl.dn, pl.x_final, pl.fstats[0], pl.fstats[1], params["norm.weight"], M, D, dx=G, dx16=g16,
                              dgamma=grads["norm.weight"], dbeta=grads["norm.bias"], accumulate=pl.acc)
        else:
            hip.zero_(G[:M])
            hip.zero_(g16[:M])
            for k, sg in enumerate(pl.segs):
                c0 = sg.cls0
                hip.layernorm_bwd(pl.dn[c0:], pl.x_final[c0:], pl.fstats[0][c0:], pl.fstats[1][c0:], params["norm.weight"], sg.B, D,
                                  dx=G[c0:], dx16=g16[c0:], dgamma=grads["norm.weight"], dbeta=grads["norm.bias"],
                                  accumulate=pl.acc or k > 0)
        if run.region_layer is not None and d_region is None and not pl.acc:
            for k in ("region_norm.weight", "region_norm.bias"):
                hip.zero_(grads[k])

    def _attn_bwd(self, pl, fin, qkv, o, lse, d_o, d_qkv, cls_query_only=False):
        """attention backward with the fused CLS-row finalize (fin = hip.attn_space_bwd_fin | hip.attn_time_bwd_fin: the last workgroup
        per (sample, head) writes the CLS row), per segment - each clip is a self-contained row range; two clips of one geometry
        share a launch"""
        clips = lambda: [dict(qkv=sg.rows(qkv), out=sg.rows(o), lse=sg.rows(lse), dout=sg.rows(d_o), dqkv=sg.rows(d_qkv),
                              cls_side=sg.cls_side, done=sg.cls_done, B=sg.B, T=sg.T) for sg in pl.segs]
        two = len(pl.segs) == 2 and pl.segs[0].N == pl.segs[1].N
        if fin is hip.attn_space_bwd_fin and (two or (cls_query_only and len(pl.segs) == 1)):
            # cls_query_only (pruned top block): dO of every patch query is zero and its lse +inf - the launch skips their exact zeros
            hip.attn_space_bwd_clips(clips(), pl.segs[0].N, self.H, self.D, self.scale, cls_query_only=cls_query_only)
            return
        pow2 = lambda t: 1 <= t <= 16 and t & (t - 1) == 0
        if fin is hip.attn_time_bwd_fin and two and pl.segs[0].T == 1 and pl.segs[1].T > 1 and pow2(pl.segs[1].T):
            # the one-frame object clip (a 32 us launch on a fraction of the GPU) rides in the video clip's launch
            hip.attn_time_bwd_clips(clips(), pl.segs[0].N, self.H, self.D, self.scale)
            return
        for sg in pl.segs:
            fin(sg.rows(qkv), sg.rows(o), sg.rows(lse), sg.rows(d_o), sg.rows(d_qkv), sg.cls_side, sg.cls_done, sg.B, sg.T, sg.N, self.H,
                self.D, self.scale)

    def _block_bwd(self, pl, i, run, params, grads, d_region):
        M = pl.M
        D, Hd = self.D, self.Hd
        G = pl.G
        a = pl.blocks[i]
        st8 = pl.sets[i % 2]
        ga, ga_next = pl.ga[i % 3], pl.ga[(i - 1) % 3]     # dL/d(block output) bf16 ; written by this block's LN3 bwd
        grid = pl.bwd_grid
        rl = run.region_layer
        if rl is not None and d_region is not None and i + 1 == rl:
            # region tokens branch off the output of block rl-1: add their gradient to the stream
            # (all M rows: the CLS rows in between carry a zero gradient and come out unchanged)
            if pl.res16:
                hip.layernorm_bwd_r16(d_region, a.out, pl.rstats[0], pl.rstats[1], params["region_norm.weight"], M, D, dx16=ga,
                                      dres16=ga, dgamma=grads["region_norm.weight"], dbeta=grads["region_norm.bias"],
                                      accumulate=pl.acc)
            else:
                hip.layernorm_bwd(d_region, a.out, pl.rstats[0], pl.rstats[1], params["region_norm.weight"],
                                  M, D, dx=G, dx16=ga, dres=G, dgamma=grads["region_norm.weight"],
                                  dbeta=grads["region_norm.bias"], accumulate=pl.acc)
        gr = lambda s: grads[f"blocks.{i}.{s}"]
        wT = lambda s: self.shadow[f"blocks.{i}.{s}.weight"][1]
        st = a.stats
        d_h, gb, d_qkv_s, gc, d_qkv_t = st8["d_h"], st8["gb"], st8["d_qkv_s"], st8["gc"], st8["d_qkv_t"]

        def ln_bwd(norm, k, xhat, dx16):
            """backward of norm3 / norm1 / norm2 (rstd = stats row 2k + 1) from the saved bf16 xhat.
            bf16 gradient stream (res16): gb = dL/dy = ga + dx2 (the space branch's dY AND the stream), gc = dx1 alone, and the block's
            outgoing gradient ga_next = gb + gc + dx3; block 0 also leaves the fp32 copy G the embedding reads.
            fp32 gradient stream: G is read by norm2 (its own dx stays bf16 in pl.dx2_16), untouched by norm1 (gc = its dx) and read +
            written once by norm3, which adds the two bf16 increments: G_out = G_in + dx2 + dx1 + dx3."""
            if pl.res16:
                if norm == "norm2":
                    hip.layernorm_bwd_xhat(pl.d_a, xhat, st[2 * k + 1], M, D, dx16=dx16, add_a=ga)
                elif norm == "norm1":
                    hip.layernorm_bwd_xhat(pl.d_a, xhat, st[2 * k + 1], M, D, dx16=dx16)
                else:
                    hip.layernorm_bwd_xhat(pl.d_a, xhat, st[2 * k + 1], M, D, dx=G if i == 0 else None, dx16=dx16, add_a=gb, add_b=gc)
            elif norm == "norm2":
                hip.layernorm_bwd_xhat(pl.d_a, xhat, st[2 * k + 1], M, D, dx16=dx16, dres=G, dxp16=pl.dx2_16)
            elif norm == "norm1":
                hip.layernorm_bwd_xhat(pl.d_a, xhat, st[2 * k + 1], M, D, dx16=dx16)
            else:
                hip.layernorm_bwd_xhat(pl.d_a, xhat, st[2 * k + 1], M, D, dx=G, dx16=dx16, dres=G, add_a=pl.dx2_16, add_b=gc)

        def wgrad_folded(P, Q, n1, n2, lin):
            """weight gradient of a linear layer that carries a folded LayerNorm: dW' (and db') go to the gradient buffers (in
            place; oat_ln_fold_grads finishes them at the block's end) or - when this backward ACCUMULATES into gradients an
            earlier backward of the step already finished - to scratch buffers the fold kernel adds from."""
            if pl.acc:
                if lin not in self._fold_tmp:
                    self._fold_tmp[lin] = (torch.empty(n1, n2, dtype=torch.float32, device=P.device),
                                           torch.empty(n1, dtype=torch.float32, device=P.device))
                tw, tb = self._fold_tmp[lin]
                self._wgrad(P, Q, M, n1, n2, tw, tb, False, pl=pl)
            else:
                self._wgrad(P, Q, M, n1, n2, gr(lin + ".weight"), gr(lin + ".bias"), pl.acc, pl=pl)

        # ---- MLP: out = y + fc2(gelu(fc1(LN2(y))))
        if pl.h_u8:
            hip.gemm_nt(ga, wT("mlp.fc2"), M, Hd, D, hip.EPI_MUL_AUX | hip.EPI_U8, d_h, aux=a.h8, grid=grid)
        else:
            hip.gemm_nt(ga, wT("mlp.fc2"), M, Hd, D, hip.EPI_MUL_AUX, d_h, aux=a.h, grid=grid)
        hip.gemm_nt(d_h, wT("mlp.fc1"), M, D, Hd, hip.EPI_BF16, pl.d_a, grid=grid)
        ln_bwd("norm2", 2, a.a2, gb)                                                                 # gb = dL/dy
        self._wgrad(ga, a.g, M, D, Hd, gr("mlp.fc2.weight"), gr("mlp.fc2.bias"), pl.acc, pl=pl)
        # ---- space attention: y = x + proj(attn(LN1(xt)))
        hip.gemm_nt(gb, wT("attn.proj"), M, D, D, hip.EPI_BF16, pl.d_o, grid=grid)
        self._attn_bwd(pl, hip.attn_space_bwd_fin, a.qkv_s, a.o_s, a.lse_s, pl.d_o, d_qkv_s)
        wgrad_folded(d_h, a.a2, Hd, D, "mlp.fc1")
        hip.gemm_nt(d_qkv_s, wT("attn.qkv"), M, D, 3 * D, hip.EPI_BF16, pl.d_a, grid=grid)
        ln_bwd("norm1", 1, a.a1, gc)                                                                 # gc = dL/dxt alone (feeds the time branch)
        wgrad_folded(d_qkv_s, a.a1, 3 * D, D, "attn.qkv")
        # ---- time attention: xt = x + proj(attn(LN3(x)))
        hip.gemm_nt(gc, wT("timeattn.proj"), M, D, D, hip.EPI_BF16, pl.d_o, grid=grid)
        self._attn_bwd(pl, hip.attn_time_bwd_fin, a.qkv_t, a.o_t, a.lse_t, pl.d_o, d_qkv_t)
        self._wgrad(gb, a.o_s, M, D, D, gr("attn.proj.weight"), gr("attn.proj.bias"), pl.acc, pl=pl)
        self._wgrad(gc, a.o_t, M, D, D, gr("timeattn.proj.weight"), gr("timeattn.proj.bias"), pl.acc, pl=pl)
        hip.gemm_nt(d_qkv_t, wT("timeattn.qkv"), M, D, 3 * D, hip.EPI_BF16, pl.d_a, grid=grid)
        ln_bwd("norm3", 0, a.a3, ga_next)                                                            # ga_next = dL/dx
        wgrad_folded(d_qkv_t, a.a3, 3 * D, D, "timeattn.qkv")
        pl.fold_pending = i            # dW' -> dW, dgamma, dbeta after the block's weight gradients have run (_flush_wgrads)

    def _top_block_bwd_pruned(self, pl, i, params, grads):
        """_block_bwd of the top block after a prune_top forward (bf16 streams, folded LayerNorms, in-order schedule).  The
        incoming gradient ga is non-zero on the B CLS rows only, so fc2 / fc1 / norm2 / the space projection - data and weight
        gradients - take those rows alone; from the space attention downwards (its keys and values belong to every row) the
        block runs as usual.  gb (= dL/dy) exists on the CLS rows only: norm3's backward adds it there in a second, B-row launch."""
        M, D, Hd = pl.M, self.D, self.Hd
        G, a, st8 = pl.G, pl.blocks[i], pl.sets[i % 2]
        ga, ga_next = pl.ga[i % 3], pl.ga[(i - 1) % 3]
        gr = lambda s: grads[f"blocks.{i}.{s}"]
        wT = lambda s: self.shadow[f"blocks.{i}.{s}.weight"][1]
        st = a.stats
        d_h, gb, d_qkv_s, gc, d_qkv_t = st8["d_h"], st8["gb"], st8["d_qkv_s"], st8["gc"], st8["d_qkv_t"]
        segs = pl.segs

        def wgrad(P, Q, rows, n1, n2, lin, folded, queue, more=False):
            """as wgrad_folded of _block_bwd; queue=False: a B-row problem, launched on its own (gemm_tn); more: a later
            clip's rows of the same weight - added to what the first clip's launch wrote"""
            w_, b_, acc = gr(lin + ".weight"), gr(lin + ".bias"), pl.acc
            if folded and pl.acc:
                if lin not in self._fold_tmp:
                    self._fold_tmp[lin] = (torch.empty(n1, n2, dtype=torch.float32, device=P.device),
                                           torch.empty(n1, dtype=torch.float32, device=P.device))
                (w_, b_), acc = self._fold_tmp[lin], False
            self._wgrad(P, Q, rows, n1, n2, w_, b_, acc or more, pl=pl if queue else None)

        # ---- MLP, CLS rows (per clip: its tail block)
        for k, sg in enumerate(segs):
            c0, Bc = sg.cls0, sg.B
            R = lambda t: t[c0:]
            hip.gemm_nt(R(ga), wT("mlp.fc2"), Bc, Hd, D, hip.EPI_MUL_AUX, R(d_h), aux=R(a.h))
            hip.gemm_nt(R(d_h), wT("mlp.fc1"), Bc, D, Hd, hip.EPI_BF16, R(pl.d_a))
            hip.layernorm_bwd_xhat(R(pl.d_a), R(a.a2), st[5][c0:], Bc, D, dx16=R(gb), add_a=R(ga))           # gb = ga + dx2
            wgrad(R(ga), R(a.g), Bc, D, Hd, "mlp.fc2", False, False, more=k > 0)
        # ---- space attention: the projection on the CLS rows, the attention itself on every row (dO of the patch queries = 0)
        d_o = pl.d_o_top
        for sg in segs:
            hip.gemm_nt(gb[sg.cls0:], wT("attn.proj"), sg.B, D, D, hip.EPI_BF16, d_o[sg.cls0:])
        self._attn_bwd(pl, hip.attn_space_bwd_fin, a.qkv_s, a.o_s, a.lse_s, d_o, d_qkv_s, cls_query_only=True)
        for k, sg in enumerate(segs):
            c0, Bc = sg.cls0, sg.B
            wgrad(d_h[c0:], a.a2[c0:], Bc, Hd, D, "mlp.fc1", True, False, more=k > 0)
            wgrad(gb[c0:], a.o_s[c0:], Bc, D, D, "attn.proj", False, False, more=k > 0)
        hip.gemm_nt(d_qkv_s, wT("attn.qkv"), M, D, 3 * D, hip.EPI_BF16, pl.d_a, grid=pl.bwd_grid)
        hip.layernorm_bwd_xhat(pl.d_a, a.a1, st[3], M, D, dx16=gc)                                            # gc = dx1
        wgrad(d_qkv_s, a.a1, M, 3 * D, D, "attn.qkv", True, True)
        # ---- time attention: as in _block_bwd
        hip.gemm_nt(gc, wT("timeattn.proj"), M, D, D, hip.EPI_BF16, pl.d_o, grid=pl.bwd_grid)
        self._attn_bwd(pl, hip.attn_time_bwd_fin, a.qkv_t, a.o_t, a.lse_t, pl.d_o, d_qkv_t)
        wgrad(gc, a.o_t, M, D, D, "timeattn.proj", False, True)
        hip.gemm_nt(d_qkv_t, wT("timeattn.qkv"), M, D, 3 * D, hip.EPI_BF16, pl.d_a, grid=pl.bwd_grid)
        # ga_next = gb + gc + dx3: gb is zero on the patch rows (not stored there), present on the CLS rows
        for sg in segs:
            r0, c0, Bc = sg.row0, sg.cls0, sg.B
            hip.layernorm_bwd_xhat(pl.d_a[r0:], a.a3[r0:], st[1][r0:], c0 - r0, D, dx=G[r0:] if i == 0 else None, dx16=ga_next[r0:],
                                   add_b=gc[r0:])
            hip.layernorm_bwd_xhat(pl.d_a[c0:], a.a3[c0:], st[1][c0:], Bc, D, dx=G[c0:] if i == 0 else None, dx16=ga_next[c0:],
                                   add_a=gb[c0:], add_b=gc[c0:])
        wgrad(d_qkv_t, a.a3, M, 3 * D, D, "timeattn.qkv", True, True)
        pl.fold_pending = i

    def _embed_bwd(self, pl, grads):
        """x0[patch] = cols @ Wp^T + b + pos[1+n] + temporal[f] ; x0[cls] = cls + pos[0]"""
        D = self.D
        G = pl.G
        gw = grads["patch_embed.proj.weight"]
        gt = grads["temporal_embed"].view(-1, D)
        gcls = grads["cls_token"].view(1, D)
        g_in = pl.ga[(-1) % 3]                                # dL/dx0 as bf16 (block 0's LayerNorm-3 backward)
        for k, sg in enumerate(pl.segs):                      # per clip: its own im2col columns, frame count, CLS rows
            B, T, N = sg.B, sg.T, sg.N
            acc = pl.acc or k > 0
            self._wgrad(g_in[sg.row0:], sg.cols, sg.BTN, D, self.Kp, gw.view(D, self.Kp), grads["patch_embed.proj.bias"], acc)
            hip.periodic_rowsum(G[sg.row0:], B, T * N, D, sg.Gp)
            gpos = grads["pos_embed"].view(N + 1, D)
            hip.periodic_rowsum(sg.Gp, T, N, D, gpos[1:], accumulate=acc)
            if T < gt.shape[0] and not acc:
                hip.zero_(gt[T:])
            hip.grouped_rowsum(sg.Gp, T, N, D, gt[:T], accumulate=acc)
            hip.grouped_rowsum(G[sg.cls0:], 1, B, D, gcls, accumulate=acc)
        hip.copy_(gpos[:1], gcls)        # pos_embed[0] only ever meets the CLS token: its gradient IS cls_token's
