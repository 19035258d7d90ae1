"""DistilBERT forward/backward as an explicit HIP launch schedule.

The reference calls HF transformers' DistilBertModel (third party; call sites
/root/reference/OATrans/model/oa_model.py:27,113-121).  Its published algorithm - learned
word + position embeddings -> LayerNorm(1e-12) -> n x post-LN [MHSA with key mask, FFN(GELU)] -
is executed here with the same GEMM / LayerNorm kernels as the video encoder plus the masked
attention and embedding kernels of csrc/text.hip.  In training mode (`drop` given to forward) HF's three dropout sites
are active - after the embedding LayerNorm, on the attention probabilities, after ffn.lin2 - with counter-based
Philox masks (csrc/rng.h) that backward regenerates; eval mode (the parity configuration of the goldens) has none.

Rows are (b, l) -> b * L + l.  FORWARD runs in fp32 end to end (oat_linear_f32 on the fp32 master weights, the precise
path of oat_attn_text_fwd_dual): the tower is 0.7 % of the step's FLOPs, runs on its own stream beside the video
tower, and with bf16 operands its ~6e-3 embedding error alone would use up the 1e-3 sim-matrix bound.  Every layer
also leaves the bf16 copies (LayerNorm outputs, q|k|v, context, GELU and its derivative) that BACKWARD - bf16 MFMA
GEMMs as in engine/video.py - reads, so backward differentiates one self-consistent bf16 function.
"""
import os

import torch

from ..ops import hip


def _round_up(x, m):
    return (x + m - 1) // m * m


class _LayerActs:
    def __init__(self, Mp, D, Hd, H, dev):
        z16 = lambda c: torch.zeros(Mp, c, dtype=torch.bfloat16, device=dev)
        z32 = lambda c: torch.zeros(Mp, c, dtype=torch.float32, device=dev)
        self.qkv, self.ctx = z16(3 * D), z16(D)
        self.qkv32, self.ctx32, self.g32 = z32(3 * D), z32(D), z32(Hd)      # precise forward chain
        self.lse = z32(H)
        self.s, self.x1, self.f, self.x2 = z32(D), z32(D), z32(D), z32(D)
        self.x1_16, self.x2_16 = z16(D), z16(D)
        self.h, self.g = z16(Hd), z16(Hd)
        self.stats = torch.zeros(4, Mp, dtype=torch.float32, device=dev)
        # backward: the incoming gradients of this layer's four linears stay until the END of the tower's backward, when
        # all 36 weight gradients run as ONE grouped launch (csrc/gemm_tn_sk.hip): 14 MB per layer at B = 32, L = 32
        self.dy_lin2, self.d_h, self.dy_out, self.d_qkv = z16(D), z16(Hd), z16(D), z16(3 * D)


class _TextPlan:
    def __init__(self, B, L, D, Hd, H, n_layers, dev):
        self.B, self.L, self.M = B, L, B * L
        self.Mp = Mp = _round_up(self.M, 256)
        self.layers = [_LayerActs(Mp, D, Hd, H, dev) for _ in range(n_layers)]
        self.emb = torch.zeros(Mp, D, dtype=torch.float32, device=dev)
        self.x0 = torch.zeros(Mp, D, dtype=torch.float32, device=dev)
        self.x0_16 = torch.zeros(Mp, D, dtype=torch.bfloat16, device=dev)
        self.estats = torch.zeros(2, Mp, dtype=torch.float32, device=dev)
        self.G = torch.zeros(Mp, D, dtype=torch.float32, device=dev)
        self.g16 = torch.zeros(Mp, D, dtype=torch.bfloat16, device=dev)
        self.d_h = torch.zeros(Mp, Hd, dtype=torch.bfloat16, device=dev)
        self.d_ctx = torch.zeros(Mp, D, dtype=torch.bfloat16, device=dev)
        self.d_qkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device=dev)
        self.delta = torch.zeros(Mp, H, dtype=torch.float32, device=dev)
        self.rng = torch.zeros(2, dtype=torch.int64, device=dev)       # {seed, offset} this forward drew its masks with
        self.drop = None                                               # (p_hidden, p_attention) of the last forward
        # static copies of the inputs (ids, mask, incoming gradient): the schedules replay from launch tapes
        self.ids = torch.zeros(B, L, dtype=torch.int64, device=dev)
        self.mask = torch.zeros(B, L, dtype=torch.int64, device=dev)
        self.tape_fwd = self.tape_bwd = None
        self.tn_group = None                                           # (key, hip.TnGroup) of the tower's weight gradients


class TextEngine:
    """`params` / `grads` map HF DistilBertModel state_dict names (no `text_model.` prefix)."""

    def __init__(self, n_layers, dim, n_heads, hidden_dim):
        self.n_layers, self.D, self.H, self.Hd = n_layers, dim, n_heads, hidden_dim
        if dim // n_heads != 64:
            raise hip.OatError("the HIP attention kernels are built for head_dim 64")
        self.scale = 64 ** -0.5
        self.plans, self.shadow, self.versions = {}, {}, None
        self.use_tape = os.environ.get("OAT_TAPE", "1") != "0"
        self.qkv_one_launch = True      # q / k / v linears of a layer in one launch (hip.linear_f32_qkv); False: three (bit-identical, tests)
        self.group_wgrads = True        # the 36 weight gradients of a backward pass as one grouped launch at its end; False: 36 gemm_tn + tn_reduce pairs

    def refresh_shadows(self, params, sig=None):
        names = []
        for i in range(self.n_layers):
            b = f"transformer.layer.{i}."
            names += [b + f"attention.{l}.weight" for l in ("q_lin", "k_lin", "v_lin", "out_lin")]
            names += [b + f"attention.{l}.bias" for l in ("q_lin", "k_lin", "v_lin")]
            names += [b + "ffn.lin1.weight", b + "ffn.lin2.weight"]
        versions = sig
        if sig is not None and versions == self.versions:
            return
        dev = params[names[0]].device
        D = self.D
        srcs = [params[n].detach() for n in names if n.endswith(".weight")]
        key = tuple(w.data_ptr() for w in srcs)
        if getattr(self, "_cast", None) is None or self._cast.key != key:
            entries = []

            def shadow(k, rows, cols):
                self.shadow[k] = (torch.empty(rows, cols, dtype=torch.bfloat16, device=dev),
                                  torch.empty(cols, rows, dtype=torch.bfloat16, device=dev))
                return self.shadow[k]

            for i in range(self.n_layers):
                b = f"transformer.layer.{i}."
                w, wT = shadow(b + "qkv", 3 * D, D)             # q|k|v rows concatenated by the cast itself
                for j, l in enumerate(("q_lin", "k_lin", "v_lin")):
                    entries.append((params[b + f"attention.{l}.weight"].detach(), w[j * D:(j + 1) * D],
                                    wT[:, j * D:(j + 1) * D], D, 3 * D))
                for l in ("attention.out_lin", "ffn.lin1", "ffn.lin2"):
                    src = params[b + l + ".weight"].detach()
                    w, wT = shadow(b + l, src.shape[0], src.shape[1])
                    entries.append((src, w, wT, src.shape[1], src.shape[0]))
            self._cast = hip.CastTable(entries)
        self._cast.run()
        for i in range(self.n_layers):
            b = f"transformer.layer.{i}."
            self.shadow[b + "qkv.bias"] = torch.cat(
                [params[b + f"attention.{l}.bias"].detach() for l in ("q_lin", "k_lin", "v_lin")], 0)
        self.versions = versions

    def plan(self, B, L, dev, slot=0):
        key = (B, L, str(dev), slot)         # one plan per call within a step (caption / caption+tags passes)
        if key not in self.plans:
            self.plans[key] = _TextPlan(B, L, self.D, self.Hd, self.H, self.n_layers, dev)
        return self.plans[key]

    @staticmethod
    def site(layer, kind):
        """Mask site numbers: 0 = embeddings; layer i: 1 + 2i = attention probabilities, 2 + 2i = ffn output."""
        return 0 if kind == "emb" else 1 + 2 * layer + (0 if kind == "attn" else 1)

    def forward(self, input_ids, attention_mask, params, sig=None, slot=0, drop=None):
        """-> (last_hidden fp32 view [B, L, D], plan).  drop = (p_hidden, p_attention, rng_state) switches the training-
        mode dropout on: rng_state (hip.new_rng_state) is ticked and copied into the plan, so this call's backward sees
        the masks of this call whatever runs in between."""
        B, L = input_ids.shape
        D, Hd, H = self.D, self.Hd, self.H
        self.refresh_shadows(params, sig)
        pl = self.plan(B, L, input_ids.device, slot)
        M = pl.M
        pl.ids.copy_(input_ids)
        pl.mask.copy_(attention_mask)
        ph = pa = 0.0
        pl.drop = None
        state = None
        if drop is not None and (drop[0] > 0 or drop[1] > 0):
            ph, pa, state = drop
            pl.drop = (ph, pa)
        key = self._tape_key(pl, params, None, ph, pa, state.data_ptr() if state is not None else 0)
        x = self._taped(pl, "tape_fwd", key, lambda: self._forward_body(pl, params, ph, pa, state))
        return x[:M].view(B, L, D), pl

    def _tape_key(self, pl, params, grads, *flags):
        if not self.use_tape:
            return None
        ptrs = tuple(t.data_ptr() for t in params.values())
        gptr = next(iter(grads.values())).data_ptr() if grads else 0
        return (torch.cuda.current_stream().cuda_stream, ptrs, gptr, self.group_wgrads, self.qkv_one_launch, flags)

    @staticmethod
    def _taped(pl, slot, key, body):
        """Run the launch schedule `body`, recording its tape the first time and replaying it afterwards (csrc/tape.hip)."""
        if key is None:
            return body()
        held = getattr(pl, slot)
        if held is not None and held[0] == key:
            hip.tape_replay(held[1])
            return held[2]
        if held is not None:
            hip.tape_free(held[1])
            setattr(pl, slot, None)
        hip.tape_begin()
        try:
            out = body()
        except BaseException:
            hip.tape_abort()
            raise
        setattr(pl, slot, (key, hip.tape_end(), out))
        return out

    def _forward_body(self, pl, params, ph, pa, state):
        B, L, M = pl.B, pl.L, pl.M
        D, Hd, H = self.D, self.Hd, self.H
        hip.embed_fwd(pl.ids, params["embeddings.word_embeddings.weight"],
                      params["embeddings.position_embeddings.weight"], pl.emb, M, L, D)
        hip.layernorm_fwd(pl.emb, params["embeddings.LayerNorm.weight"], params["embeddings.LayerNorm.bias"], M, D,
                          1e-12, y=pl.x0_16, y32=pl.x0, mean=pl.estats[0], rstd=pl.estats[1])
        if state is not None:
            hip.rng_tick(state)
            hip.copy_(pl.rng, state)
            if ph > 0:
                hip.dropout(pl.x0, M, D, ph, pl.rng, self.site(0, "emb"), out32=pl.x0, out16=pl.x0_16)
        x = pl.x0
        for i, a in enumerate(pl.layers):
            b = f"transformer.layer.{i}."
            p = lambda s: params[b + s]
            if self.qkv_one_launch and M > 64 and D % 128 == 0:      # q_lin | k_lin | v_lin: three parameters, one launch
                hip.linear_f32_qkv(x, p("attention.q_lin.weight"), p("attention.k_lin.weight"), p("attention.v_lin.weight"), M, D, D,
                                   bq=p("attention.q_lin.bias"), bk=p("attention.k_lin.bias"), bv=p("attention.v_lin.bias"),
                                   out32=a.qkv32, out16=a.qkv)
            else:
                for j, l in enumerate(("q_lin", "k_lin", "v_lin")):
                    hip.linear_f32(x, p(f"attention.{l}.weight"), M, D, D, bias=p(f"attention.{l}.bias"),
                                   out32=a.qkv32[:, j * D:(j + 1) * D], out16=a.qkv[:, j * D:(j + 1) * D])
            hip.attn_text_fwd_dual(a.qkv, a.qkv32, pl.mask, a.ctx, a.ctx32, a.lse, B, L, H, D, self.scale,
                                   drop_p=pa, rng=pl.rng if pa > 0 else None, site=self.site(i, "attn"))
            hip.linear_f32(a.ctx32, p("attention.out_lin.weight"), M, D, D, bias=p("attention.out_lin.bias"),
                           out32=a.s, resid=x)
            hip.layernorm_fwd(a.s, p("sa_layer_norm.weight"), p("sa_layer_norm.bias"), M, D, 1e-12, y=a.x1_16,
                              y32=a.x1, mean=a.stats[0], rstd=a.stats[1])
            hip.linear_f32(a.x1, p("ffn.lin1.weight"), M, Hd, D, bias=p("ffn.lin1.bias"), out32=a.g32, out16=a.g,
                           out16b=a.h, act=hip.LIN_GELU)                 # a.g = gelu(h), a.h = gelu'(h) for backward
            if ph > 0:                                                   # f = x1 + dropout(lin2(.))
                hip.linear_f32(a.g32, p("ffn.lin2.weight"), M, D, Hd, bias=p("ffn.lin2.bias"), out32=a.f)
                hip.dropout(a.f, M, D, ph, pl.rng, self.site(i, "ffn"), resid=a.x1, out32=a.f)
            else:
                hip.linear_f32(a.g32, p("ffn.lin2.weight"), M, D, Hd, bias=p("ffn.lin2.bias"), out32=a.f, resid=a.x1)
            hip.layernorm_fwd(a.f, p("output_layer_norm.weight"), p("output_layer_norm.bias"), M, D, 1e-12,
                              y=a.x2_16, y32=a.x2, mean=a.stats[2], rstd=a.stats[3])
            x = a.x2
        return x

    def backward(self, pl, params, grads, d_hidden, accumulate=False):
        """d_hidden: fp32 [B, L, D] gradient of last_hidden_state.  Writes (or accumulates) every
        text parameter gradient into `grads`."""
        M, D = pl.M, self.D
        pl.G[:M].copy_(d_hidden.reshape(M, D))
        ph, pa = pl.drop if pl.drop is not None else (0.0, 0.0)
        key = self._tape_key(pl, params, grads, "bwd", bool(accumulate), ph, pa)
        self._taped(pl, "tape_bwd", key, lambda: self._backward_body(pl, params, grads, bool(accumulate), ph, pa))

    def _backward_body(self, pl, params, grads, acc, ph, pa):
        B, L, M = pl.B, pl.L, pl.M
        D, Hd, H = self.D, self.Hd, self.H
        G = pl.G
        grouped = self.group_wgrads and D % 256 == 0 and Hd % 256 == 0
        wq = []                                   # (P, Q, M, N1, N2, dW, db, accumulate): launched together after the last layer

        def wgrad(P, Q, n1, n2, w, bias):
            if grouped:
                wq.append((P, Q, M, n1, n2, w, bias, acc))
            else:
                hip.gemm_tn(P, Q, M, n1, n2, w, accumulate=acc, bias_out=bias)

        for i in reversed(range(self.n_layers)):
            a = pl.layers[i]
            x16 = pl.layers[i - 1].x2_16 if i > 0 else pl.x0_16
            b = f"transformer.layer.{i}."
            p = lambda s: params[b + s]
            gr = lambda s: grads[b + s]
            wT = lambda s: self.shadow[b + s][1]
            dy_lin2, d_h, dy_out, d_qkv = (a.dy_lin2, a.d_h, a.dy_out, a.d_qkv) if grouped else (pl.g16, pl.d_h, pl.g16, pl.d_qkv)
            # x2 = LN(f), f = x1 + lin2(gelu(lin1(x1)))
            hip.layernorm_bwd(G, a.f, a.stats[2], a.stats[3], p("output_layer_norm.weight"), M, D, dx=G, dx16=dy_lin2,
                              dgamma=gr("output_layer_norm.weight"), dbeta=gr("output_layer_norm.bias"),
                              accumulate=acc)                                              # G = dL/df
            if ph > 0:                       # lin2's output met the mask; the residual path (G) did not
                hip.dropout(G, M, D, ph, pl.rng, self.site(i, "ffn"), out16=dy_lin2)
            wgrad(dy_lin2, a.g, D, Hd, gr("ffn.lin2.weight"), gr("ffn.lin2.bias"))
            hip.gemm_nt(dy_lin2, wT("ffn.lin2"), M, Hd, D, hip.EPI_MUL_AUX, d_h, aux=a.h)
            wgrad(d_h, a.x1_16, Hd, D, gr("ffn.lin1.weight"), gr("ffn.lin1.bias"))
            hip.gemm_nt(d_h, wT("ffn.lin1"), M, D, Hd, hip.EPI_F32, G, resid=G)             # G = dL/dx1
            # x1 = LN(s), s = x + out_lin(attn(qkv(x)))
            hip.layernorm_bwd(G, a.s, a.stats[0], a.stats[1], p("sa_layer_norm.weight"), M, D, dx=G, dx16=dy_out,
                              dgamma=gr("sa_layer_norm.weight"), dbeta=gr("sa_layer_norm.bias"), accumulate=acc)
            wgrad(dy_out, a.ctx, D, D, gr("attention.out_lin.weight"), gr("attention.out_lin.bias"))
            hip.gemm_nt(dy_out, wT("attention.out_lin"), M, D, D, hip.EPI_BF16, pl.d_ctx)
            hip.attn_text_bwd(a.qkv, pl.mask, a.ctx, a.lse, pl.delta, pl.d_ctx, d_qkv, B, L, H, D, self.scale,
                              drop_p=pa, rng=pl.rng if pa > 0 else None, site=self.site(i, "attn"))
            for k, l in enumerate(("q_lin", "k_lin", "v_lin")):      # q | k | v gradients are separate tensors: three problems
                wgrad(d_qkv[:, k * D:(k + 1) * D], x16, D, D, gr(f"attention.{l}.weight"), gr(f"attention.{l}.bias"))
            hip.gemm_nt(d_qkv, wT("qkv"), M, D, 3 * D, hip.EPI_F32, G, resid=G)              # G = dL/dx
        if wq:
            # all 36 weight gradients: 648 output tiles of 256 x 256 over M = B L rows, whole tiles per workgroup (stream
            # mode: nothing is split, no partial tiles) - one launch instead of 36 gemm_tn + 36 tn_reduce
            key = tuple((P.data_ptr(), Q.data_ptr(), w.data_ptr(), bias.data_ptr(), acc) for P, Q, _, _, _, w, bias, acc in wq)
            if pl.tn_group is None or pl.tn_group[0] != key:
                pl.tn_group = (key, hip.TnGroup(wq, splits=0))
            pl.tn_group[1].run()
        # embeddings: x0 = dropout(LN(word[ids] + pos[l]))
        if ph > 0:
            hip.dropout(G, M, D, ph, pl.rng, self.site(0, "emb"), out32=G)
        hip.layernorm_bwd(G, pl.emb, pl.estats[0], pl.estats[1], params["embeddings.LayerNorm.weight"], M, D, dx=G,
                          dgamma=grads["embeddings.LayerNorm.weight"], dbeta=grads["embeddings.LayerNorm.bias"],
                          accumulate=acc)
        gword = grads["embeddings.word_embeddings.weight"]
        if not acc:
            hip.zero_(gword)
        hip.embed_bwd(pl.ids, G, gword, M, D)
        gpos = grads["embeddings.position_embeddings.weight"]
        if not acc:
            hip.zero_(gpos[L:])
        hip.periodic_rowsum(G, B, L, D, gpos[:L], accumulate=acc)
