"""SpaceTimeTransformer forward/backward as an explicit kernel schedule over caller-owned HBM.

This is the MI355X-native execution of
  /root/reference/OATrans/model/video_transformer.py:303-351 (forward_features)
  /root/reference/OATrans/model/video_transformer.py:161-176 (SpaceTimeBlock.forward)
and of the autograd graph PyTorch would record for them.  Nothing here traces or compiles:
forward and backward are fixed launch sequences of liboatrans_hip.so entry points on the
current HIP stream, over activation buffers sized once per (B, T) "plan" (288 GB of HBM3E
holds every saved activation of a bs-64 8-frame step, so nothing is recomputed).

HBM layout
  token rows  : patch (b,f,n) -> row (b*T+f)*N + n ; CLS(b) -> row B*T*N + b ; M = B*T*N + B
                rows, padded with zero rows to a multiple of 256 (GEMM tiles never branch on M)
  residuals   : fp32 [Mp, D]          (x, x+time, x+space, block output)
  GEMM inputs : bf16 [Mp, D|3D|4D]    (LN outputs, qkv, attention outputs, MLP hidden)
  weights     : fp32 masters (nn.Parameters, reference state_dict names) + bf16 shadows
                W [out,in] (forward "NT" operand) and W^T [in,out] (data-gradient operand)
"""
import torch

from ..ops import hip


def _round_up(x, m):
    return (x + m - 1) // m * m


class _BlockActs:
    """Saved activations of one SpaceTimeBlock (all caller-owned, reused every step)."""

    def __init__(self, Mp, D, Hd, H, dev):
        z16 = lambda c: torch.zeros(Mp, c, dtype=torch.bfloat16, device=dev)
        z32 = lambda c: torch.zeros(Mp, c, dtype=torch.float32, device=dev)
        self.a3, self.a1, self.a2 = z16(D), z16(D), z16(D)
        self.qkv_t, self.qkv_s = z16(3 * D), z16(3 * D)
        self.o_t, self.o_s = z16(D), z16(D)
        self.h, self.g = z16(Hd), z16(Hd)
        self.xt, self.y, self.out = z32(D), z32(D), z32(D)
        self.lse_t, self.lse_s = z32(H), z32(H)
        self.stats = torch.zeros(6, Mp, dtype=torch.float32, device=dev)   # mean/rstd of norm3, norm1, norm2


class _Plan:
    def __init__(self, B, T, N, D, Hd, H, depth, Kp, dev):
        self.B, self.T, self.N = B, T, N
        self.M = B * T * N + B
        self.Mp = _round_up(self.M, 256)
        Mp = self.Mp
        self.blocks = [_BlockActs(Mp, D, Hd, H, dev) for _ in range(depth)]
        self.x0 = torch.zeros(Mp, D, dtype=torch.float32, device=dev)
        self.cols = torch.zeros(_round_up(B * T * N, 256), Kp, dtype=torch.bfloat16, device=dev)
        self.table = torch.zeros(T * N, D, dtype=torch.float32, device=dev)
        self.cls0 = torch.zeros(D, dtype=torch.float32, device=dev)
        self.fstats = torch.zeros(2, Mp, dtype=torch.float32, device=dev)
        self.normed = torch.zeros(Mp, D, dtype=torch.float32, device=dev)
        self.region = None                      # [B*T*N, D] fp32, allocated on first use (region_mem variant)
        self.rstats = None
        # backward temporaries.  Weight gradients (gemm_tn) run on a SIDE stream concurrently with the
        # data-gradient chain, so every bf16 dY that a weight gradient reads lives in a small ring:
        #   ga[3]  : block-input gradient (written one block ahead by LN3-backward)
        #   sets[2]: d_h, dy16, d_qkv_s, dxt16, d_qkv_t of even / odd blocks
        z16 = lambda c: torch.zeros(Mp, c, dtype=torch.bfloat16, device=dev)
        self.G = torch.zeros(Mp, D, dtype=torch.float32, device=dev)
        self.ga = [z16(D) for _ in range(3)]
        self.sets = [dict(d_h=z16(Hd), gb=z16(D), d_qkv_s=z16(3 * D), gc=z16(D), d_qkv_t=z16(3 * D)) for _ in range(2)]
        self.d_a = z16(D)
        self.d_o = z16(D)
        self.branch16 = z16(D)           # forward: bf16 branch output awaiting its fused add + LayerNorm
        self.tn_ws = torch.empty(hip.lib().oat_gemm_tn_workspace_bytes(0, 3 * D, Hd) // 4 // 8, dtype=torch.float32, device=dev)
        self.cls_side = torch.zeros(B, H, 3, 64, dtype=torch.float32, device=dev)
        self.Gp = torch.zeros(T * N, D, dtype=torch.float32, device=dev)


class VideoEngine:
    """Owns bf16 weight shadows, per-shape plans and the launch schedules.

    `params` maps the reference's state_dict names (without the `video_model.` prefix) to fp32
    CUDA tensors; `grads` maps the same names to fp32 gradient buffers the backward WRITES
    (overwrite semantics: the reference zeroes grads every step, trainer_dist.py:156)."""

    LINEARS = ("attn.qkv", "attn.proj", "timeattn.qkv", "timeattn.proj", "mlp.fc1", "mlp.fc2")

    def __init__(self, depth, embed_dim, num_heads, mlp_ratio, patch_size, in_chans, num_frames):
        self.depth, self.D, self.H = depth, embed_dim, num_heads
        self.Hd = int(embed_dim * mlp_ratio)
        self.ps, self.C, self.num_frames = patch_size, in_chans, num_frames
        self.Kp = in_chans * patch_size * patch_size
        self.scale = (embed_dim // num_heads) ** -0.5
        if embed_dim // num_heads != 64:
            raise hip.OatError("the HIP attention kernels are built for head_dim 64")
        self.plans = {}
        self.shadow = {}
        self.shadow_versions = None
        self.side = None                 # HIP stream for weight gradients (created lazily on the device)

    # ------------------------------------------------------------------ weights
    def refresh_shadows(self, params, sig=None):
        """bf16 W and W^T copies of every GEMM weight; re-cast only when a master changed
        (`sig` = EngineModule._weights_signature())."""
        names = [f"blocks.{i}.{l}.weight" for i in range(self.depth) for l in self.LINEARS]
        names.append("patch_embed.proj.weight")
        versions = sig
        if sig is not None and versions == self.shadow_versions:
            return
        for n in names:
            w = params[n].detach()
            w2 = w.reshape(w.shape[0], -1)
            if n not in self.shadow:
                self.shadow[n] = (torch.empty_like(w2, dtype=torch.bfloat16),
                                  torch.empty(w2.shape[1], w2.shape[0], dtype=torch.bfloat16, device=w.device))
            hip.cast_bf16(w2, self.shadow[n][0], self.shadow[n][1])
        self.shadow_versions = versions

    def plan(self, B, T, N, dev):
        key = (B, T, N, str(dev))
        if key not in self.plans:
            self.plans[key] = _Plan(B, T, N, self.D, self.Hd, self.H, self.depth, self.Kp, dev)
        return self.plans[key]

    # ------------------------------------------------------------------ forward
    def forward(self, video, params, need_patches=False, sig=None, region_layer=None):
        """video [B,T,C,R,R] fp32|bf16 -> (cls_normed fp32 [B,D], patches_normed fp32 [B*T*N, D] | None, plan).
        region_layer=K additionally leaves region_norm(x after block K)[patch rows] in plan.region
        (oa_video_transformer_region.py:364-376)."""
        B, T, C, R, _ = video.shape
        if T > self.num_frames:
            raise ValueError(f"{T} frames > num_frames={self.num_frames}")     # video_transformer.py:73
        g = R // self.ps
        N = g * g
        D, Hd, H = self.D, self.Hd, self.H
        self.refresh_shadows(params, sig)
        pl = self.plan(B, T, N, video.device)
        M, BTN = pl.M, B * T * N
        video = video.contiguous()
        hip.im2col(video, pl.cols, B * T, C, R, self.ps)
        hip.pos_table(params["pos_embed"], params["temporal_embed"], params["cls_token"], pl.table, pl.cls0, T, N, D)
        hip.gemm_nt(pl.cols, self.shadow["patch_embed.proj.weight"][0], BTN, D, self.Kp, hip.EPI_F32, pl.x0,
                    bias=params["patch_embed.proj.bias"], resid=pl.table, resid_mod=T * N)
        hip.broadcast_rows(pl.cls0, pl.x0[BTN:], B, D)
        # Residual adds are fused into the NEXT LayerNorm (oat_add_layernorm_fwd): the three projection /
        # fc2 GEMMs of a block write their branch output as bf16 and the streaming LN kernel forms
        # x + branch, stores the new fp32 stream and the normalised bf16 operand in one pass.
        br = pl.branch16
        x = pl.x0                      # stream entering block i (fully materialised)
        pend = None                    # (y, m16) of the previous block: out = y + m16 not yet formed
        for i, a in enumerate(pl.blocks):
            p = lambda s: params[f"blocks.{i}.{s}"]
            w = lambda s: self.shadow[f"blocks.{i}.{s}.weight"][0]
            st = a.stats
            if pend is None:
                hip.layernorm_fwd(x, p("norm3.weight"), p("norm3.bias"), M, D, 1e-6, y=a.a3, mean=st[0], rstd=st[1])
            else:
                prev = pl.blocks[i - 1]
                hip.add_layernorm_fwd(prev.y, br, prev.out, p("norm3.weight"), p("norm3.bias"), M, D, 1e-6, y=a.a3,
                                      mean=st[0], rstd=st[1])
                x = prev.out
                if region_layer is not None and i == region_layer:
                    self._region_tap(pl, params, x, BTN, D, video.device)
            hip.gemm_nt(a.a3, w("timeattn.qkv"), M, 3 * D, D, hip.EPI_BF16, a.qkv_t, bias=p("timeattn.qkv.bias"))
            self._attention(hip.attn_time_fwd, a.qkv_t, a.o_t, a.lse_t, B, T, N)
            hip.gemm_nt(a.o_t, w("timeattn.proj"), M, D, D, hip.EPI_BF16, br, bias=p("timeattn.proj.bias"))
            hip.add_layernorm_fwd(x, br, a.xt, p("norm1.weight"), p("norm1.bias"), M, D, 1e-6, y=a.a1, mean=st[2],
                                  rstd=st[3])                                       # xt = x + time
            hip.gemm_nt(a.a1, w("attn.qkv"), M, 3 * D, D, hip.EPI_BF16, a.qkv_s, bias=p("attn.qkv.bias"))
            self._attention(hip.attn_space_fwd, a.qkv_s, a.o_s, a.lse_s, B, T, N)
            hip.gemm_nt(a.o_s, w("attn.proj"), M, D, D, hip.EPI_BF16, br, bias=p("attn.proj.bias"))
            # space residual comes from x, NOT from x + time (video_transformer.py:170)
            hip.add_layernorm_fwd(x, br, a.y, p("norm2.weight"), p("norm2.bias"), M, D, 1e-6, y=a.a2, mean=st[4],
                                  rstd=st[5])                                       # y = x + space
            hip.gemm_nt(a.a2, w("mlp.fc1"), M, Hd, D, hip.EPI_GELU_DUAL, a.h, out2=a.g, bias=p("mlp.fc1.bias"))
            hip.gemm_nt(a.g, w("mlp.fc2"), M, D, Hd, hip.EPI_BF16, br, bias=p("mlp.fc2.bias"))
            pend = a                                                                # out = y + br, formed lazily
        last = pl.blocks[-1]
        pl.region_layer = region_layer
        pl.x_final = last.out
        pl.need_patches = need_patches
        tap_last = region_layer is not None and region_layer == self.depth
        if need_patches or tap_last:
            hip.add_layernorm_fwd(last.y, br, last.out, params["norm.weight"], params["norm.bias"], M, D, 1e-6,
                                  y32=pl.normed, mean=pl.fstats[0], rstd=pl.fstats[1])
            if tap_last:
                self._region_tap(pl, params, last.out, BTN, D, video.device)
            return pl.normed[BTN:M], (pl.normed[:BTN] if need_patches else None), pl
        # contract class: only the CLS rows of the last block's output are ever consumed
        hip.add_layernorm_fwd(last.y[BTN:], br[BTN:], last.out[BTN:], params["norm.weight"], params["norm.bias"], B, D,
                              1e-6, y32=pl.normed[BTN:], mean=pl.fstats[0][BTN:], rstd=pl.fstats[1][BTN:])
        return pl.normed[BTN:M], None, pl

    def _attention(self, patch_kernel, qkv, out, lse, B, T, N):
        """Patch attention on the current stream, the independent CLS-query attention (it only writes the
        CLS rows of out / lse) concurrently on the side stream."""
        main = torch.cuda.current_stream()
        if self.side is None:
            self.side = torch.cuda.Stream()
        ev = torch.cuda.Event()
        ev.record(main)                                  # qkv is complete
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            hip.attn_cls_fwd(qkv, out, lse, B, T, N, self.H, self.D, self.scale)
            done = torch.cuda.Event()
            done.record(self.side)
        patch_kernel(qkv, out, lse, B, T, N, self.H, self.D, self.scale)
        main.wait_event(done)

    def _region_tap(self, pl, params, x, BTN, D, dev):
        """region_norm(x after block K)[patch rows] (oa_video_transformer_region.py:364-376)."""
        if pl.region is None:
            pl.region = torch.zeros(BTN, D, dtype=torch.float32, device=dev)
            pl.rstats = torch.zeros(2, BTN, dtype=torch.float32, device=dev)
        hip.layernorm_fwd(x, params["region_norm.weight"], params["region_norm.bias"], BTN, D, 1e-6, y32=pl.region,
                          mean=pl.rstats[0], rstd=pl.rstats[1])

    # ------------------------------------------------------------------ backward
    def backward(self, pl, params, grads, d_cls, d_patches=None, d_region=None):
        """Writes every video parameter gradient into `grads`.  d_cls fp32 [B,D]; d_patches fp32
        [B*T*N, D] or None (contract class oa_model.FrozenInTime discards patch outputs); d_region fp32
        [B*T*N, D] = gradient of plan.region (enters the residual stream below block `region_layer`).

        Two HIP streams: the data-gradient chain (dgrad GEMMs, LayerNorm / attention backward) stays on the
        current stream; every weight gradient (gemm_tn + fused bias sums) goes to a side stream as soon as
        its dY exists.  The chain never depends on a weight gradient, so the MFMA-bound wgrad GEMMs fill
        the HBM-bound stretches (LN backward, attention backward, fp32 epilogues) of the chain."""
        B, T, N = pl.B, pl.T, pl.N
        D, Hd, H = self.D, self.Hd, self.H
        M, BTN = pl.M, B * T * N
        G = pl.G
        main = torch.cuda.current_stream()
        if self.side is None:
            self.side = torch.cuda.Stream()
        side = self.side
        side.wait_stream(main)               # side may not start before this step's forward is done
        done = {}                            # block index -> event: its wgrads finished on the side stream

        def wgrad(P, Q, n1, n2, w, b, rows=M):
            ev = torch.cuda.Event()
            ev.record(main)                  # P was produced by everything enqueued on main so far
            side.wait_event(ev)
            with torch.cuda.stream(side):
                hip.gemm_tn(P, Q, rows, n1, n2, w, bias_out=b, ws=pl.tn_ws)

        top = self.depth - 1
        g16 = pl.ga[top % 3]
        # final norm
        if pl.need_patches and d_patches is not None:
            dn = torch.cat([d_patches, d_cls], dim=0).contiguous()
            hip.layernorm_bwd(dn, pl.x_final, pl.fstats[0], pl.fstats[1], params["norm.weight"], M, D, dx=G, dx16=g16,
                              dgamma=grads["norm.weight"], dbeta=grads["norm.bias"])
        else:
            G[:BTN].zero_()
            g16[:BTN].zero_()
            hip.layernorm_bwd(d_cls.contiguous(), pl.x_final[BTN:], pl.fstats[0][BTN:], pl.fstats[1][BTN:],
                              params["norm.weight"], B, D, dx=G[BTN:], dx16=g16[BTN:], dgamma=grads["norm.weight"],
                              dbeta=grads["norm.bias"])
        rl = getattr(pl, "region_layer", None)
        if rl is not None and d_region is None:
            for k in ("region_norm.weight", "region_norm.bias"):
                grads[k].zero_()
        for i in reversed(range(self.depth)):
            a = pl.blocks[i]
            st8 = pl.sets[i % 2]
            ga = pl.ga[i % 3]                              # dL/d(block output), bf16
            ga_next = pl.ga[(i - 1) % 3]                   # written by this block's LN3 backward
            if i + 2 in done:
                main.wait_event(done[i + 2])               # ring slot i%2 (and ga[(i-1)%3]) is free again
            if rl is not None and d_region is not None and i + 1 == rl:
                # region tokens branch off the output of block rl-1: add their gradient to the stream
                hip.layernorm_bwd(d_region.contiguous(), a.out, pl.rstats[0], pl.rstats[1], params["region_norm.weight"],
                                  BTN, D, dx=G, dx16=ga, dres=G, dgamma=grads["region_norm.weight"],
                                  dbeta=grads["region_norm.bias"])
            x = pl.blocks[i - 1].out if i > 0 else pl.x0
            p = lambda s: params[f"blocks.{i}.{s}"]
            gr = lambda s: grads[f"blocks.{i}.{s}"]
            wT = lambda s: self.shadow[f"blocks.{i}.{s}.weight"][1]
            st = a.stats
            d_h, gb, d_qkv_s, gc, d_qkv_t = st8["d_h"], st8["gb"], st8["d_qkv_s"], st8["gc"], st8["d_qkv_t"]
            # ---- MLP: out = y + fc2(gelu(fc1(LN2(y))))
            wgrad(ga, a.g, D, Hd, gr("mlp.fc2.weight"), gr("mlp.fc2.bias"))
            hip.gemm_nt(ga, wT("mlp.fc2"), M, Hd, D, hip.EPI_DGELU, d_h, aux=a.h)
            wgrad(d_h, a.a2, Hd, D, gr("mlp.fc1.weight"), gr("mlp.fc1.bias"))
            hip.gemm_nt(d_h, wT("mlp.fc1"), M, D, Hd, hip.EPI_BF16, pl.d_a)
            hip.layernorm_bwd(pl.d_a, a.y, st[4], st[5], p("norm2.weight"), M, D, dx=G, dx16=gb, dres=G,
                              dgamma=gr("norm2.weight"), dbeta=gr("norm2.bias"))            # G = dL/dy
            # ---- space attention: y = x + proj(attn(LN1(xt)))
            wgrad(gb, a.o_s, D, D, gr("attn.proj.weight"), gr("attn.proj.bias"))
            hip.gemm_nt(gb, wT("attn.proj"), M, D, D, hip.EPI_BF16, pl.d_o)
            pl.cls_side.zero_()
            hip.attn_space_bwd(a.qkv_s, a.o_s, a.lse_s, pl.d_o, d_qkv_s, pl.cls_side, B, T, N, H, D, self.scale)
            hip.attn_cls_finalize(pl.cls_side, d_qkv_s, B, T, N, H, D)
            wgrad(d_qkv_s, a.a1, 3 * D, D, gr("attn.qkv.weight"), gr("attn.qkv.bias"))
            hip.gemm_nt(d_qkv_s, wT("attn.qkv"), M, D, 3 * D, hip.EPI_BF16, pl.d_a)
            # G <- dL/dy + dL/dxt (both reach x directly); gc <- dL/dxt alone (feeds the time branch)
            hip.layernorm_bwd(pl.d_a, a.xt, st[2], st[3], p("norm1.weight"), M, D, dx=G, dx16=gc, dres=G,
                              dx16_excl_res=True, dgamma=gr("norm1.weight"), dbeta=gr("norm1.bias"))
            # ---- time attention: xt = x + proj(attn(LN3(x)))
            wgrad(gc, a.o_t, D, D, gr("timeattn.proj.weight"), gr("timeattn.proj.bias"))
            hip.gemm_nt(gc, wT("timeattn.proj"), M, D, D, hip.EPI_BF16, pl.d_o)
            pl.cls_side.zero_()
            hip.attn_time_bwd(a.qkv_t, a.o_t, a.lse_t, pl.d_o, d_qkv_t, pl.cls_side, B, T, N, H, D, self.scale)
            hip.attn_cls_finalize(pl.cls_side, d_qkv_t, B, T, N, H, D)
            wgrad(d_qkv_t, a.a3, 3 * D, D, gr("timeattn.qkv.weight"), gr("timeattn.qkv.bias"))
            hip.gemm_nt(d_qkv_t, wT("timeattn.qkv"), M, D, 3 * D, hip.EPI_BF16, pl.d_a)
            hip.layernorm_bwd(pl.d_a, x, st[0], st[1], p("norm3.weight"), M, D, dx=G, dx16=ga_next, dres=G,
                              dgamma=gr("norm3.weight"), dbeta=gr("norm3.bias"))            # G = dL/dx
            ev = torch.cuda.Event()
            with torch.cuda.stream(side):
                ev.record(side)
            done[i] = ev
        # ---- token embedding: x0[patch] = cols @ Wp^T + b + pos[1+n] + temporal[f]; x0[cls] = cls + pos[0]
        g16 = pl.ga[(-1) % 3]
        gw = grads["patch_embed.proj.weight"]
        wgrad(g16, pl.cols, D, self.Kp, gw.view(D, self.Kp), grads["patch_embed.proj.bias"], rows=BTN)
        hip.periodic_rowsum(G, B, T * N, D, pl.Gp)
        gpos = grads["pos_embed"].view(N + 1, D)
        hip.periodic_rowsum(pl.Gp, T, N, D, gpos[1:])
        gt = grads["temporal_embed"].view(-1, D)
        if T < gt.shape[0]:
            gt[T:].zero_()
        hip.grouped_rowsum(pl.Gp, T, N, D, gt[:T])
        hip.grouped_rowsum(G[BTN:], 1, B, D, grads["cls_token"].view(1, D))
        gpos[:1].copy_(grads["cls_token"].view(1, D))
        main.wait_stream(side)               # every weight gradient is complete before the caller continues
