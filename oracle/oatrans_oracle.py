"""CPU fp32 ORACLE for the OA-Transformer training hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain fp32 PyTorch ops on the CPU, the arithmetic of the
reference's hot path.  It exists so that parity tests can
compare the HIP engine with *something that travels to the GPU box* (the
reference itself cannot).  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import it; the product package never does.

Pinning: the reference ships no tests / golden vectors, so this
oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, produced in the build
container by tests/golden/make_golden.py (imports /root/reference with shims)
and committed as tests/golden/*.pt.  tests/test_oracle_golden.py checks the
oracle against those fixtures to <= 1e-5.

Everything is functional: parameters come in as a flat dict keyed by the
reference's state_dict names, so one dict feeds the reference (load_state_dict),
this oracle and the HIP engine.

Reference lines restated (all under /root/reference/OATrans/):
  model/video_transformer.py:28-32    attn()
  model/video_transformer.py:35-51    Mlp
  model/video_transformer.py:54-76    VideoPatchEmbed
  model/video_transformer.py:99-135   VarAttention.forward (divided space/time + CLS)
  model/video_transformer.py:161-176  SpaceTimeBlock.forward (residual wiring)
  model/video_transformer.py:303-351  SpaceTimeTransformer.forward_features
  model/oa_model.py:97-133            FrozenInTime.forward / compute_text / compute_video
  model/oa_model.py:192-200           sim_matrix
  model/loss.py:13-25                 NormSoftmaxLoss.forward
  trainer/trainer_dist.py:29-45       AllGather_multi (fwd gather, bwd local slice)
  HF transformers DistilBertModel (third party, pinned 4.6.0 in environment.yml:115;
  not vendored under /root/reference) - published algorithm: learned word+position
  embeddings -> LayerNorm(1e-12) -> 6 x post-LN [MHSA(additive mask) , FFN(GELU)].
"""
import math
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- video encoder
def _lin(x, p, name):
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def _ln(x, p, name, eps):
    return F.layer_norm(x, (x.shape[-1],), p[name + ".weight"], p[name + ".bias"], eps)


def _softmax_av(q, k, v):
    """video_transformer.py:28-32: no mask, q already scaled."""
    return torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v


def divided_attention(x, p, name, mode, T, N, H):
    """VarAttention.forward (video_transformer.py:99-135).

    x: [B, 1+T*N, D] with token 0 = CLS and frame-major patch order.
    mode 'space': patch (f,n) attends {CLS} + the N patches of frame f.
    mode 'time' : patch (f,n) attends {CLS} + the T patches at position n.
    The CLS query attends ALL 1+T*N keys in both modes.
    """
    B, S, D = x.shape
    d = D // H
    qkv = _lin(x, p, name + ".qkv").reshape(B, S, 3, H, d).permute(2, 0, 3, 1, 4)  # [3,B,H,S,d]
    q, k, v = qkv[0] * (d ** -0.5), qkv[1], qkv[2]          # scale applied to q first (:105)
    cls_out = _softmax_av(q[:, :, :1], k, v)                 # [B,H,1,d]   (:110)
    qp = q[:, :, 1:].reshape(B, H, T, N, d)
    kp = k[:, :, 1:].reshape(B, H, T, N, d)
    vp = v[:, :, 1:].reshape(B, H, T, N, d)
    ck = k[:, :, :1].unsqueeze(2)                            # [B,H,1,1,d]
    cv = v[:, :, :1].unsqueeze(2)
    if mode == "space":
        kk = torch.cat([ck.expand(B, H, T, 1, d), kp], dim=3)            # [B,H,T,N+1,d]
        vv = torch.cat([cv.expand(B, H, T, 1, d), vp], dim=3)
        out = _softmax_av(qp, kk, vv)                                     # [B,H,T,N,d]
    elif mode == "time":
        qt, kt, vt = (t.transpose(2, 3) for t in (qp, kp, vp))            # [B,H,N,T,d]
        kk = torch.cat([ck.expand(B, H, N, 1, d), kt], dim=3)            # [B,H,N,T+1,d]
        vv = torch.cat([cv.expand(B, H, N, 1, d), vt], dim=3)
        out = _softmax_av(qt, kk, vv).transpose(2, 3)                     # [B,H,T,N,d]
    else:
        raise ValueError(mode)
    out = torch.cat([cls_out, out.reshape(B, H, T * N, d)], dim=2)       # [B,H,S,d]
    out = out.permute(0, 2, 1, 3).reshape(B, S, D)
    return _lin(out, p, name + ".proj")


def space_time_block(x, p, i, T, N, H, pre="video_model."):
    """SpaceTimeBlock.forward (video_transformer.py:161-176).  NB the space
    residual is taken from x, not from x + time_output (:170)."""
    b = f"{pre}blocks.{i}."
    t_out = divided_attention(_ln(x, p, b + "norm3", 1e-6), p, b + "timeattn", "time", T, N, H)
    s_out = divided_attention(_ln(x + t_out, p, b + "norm1", 1e-6), p, b + "attn", "space", T, N, H)
    y = x + s_out
    h = F.gelu(_lin(_ln(y, p, b + "norm2", 1e-6), p, b + "mlp.fc1"))      # exact erf GELU (:37)
    return y + _lin(h, p, b + "mlp.fc2")


def video_tokens(video, p, pre="video_model."):
    """patch-embed + CLS + positional tables (video_transformer.py:71-76,303-325).
    video: [B,T,3,R,R] -> [B, 1+T*N, D]"""
    B, T, C, R, _ = video.shape
    w = p[pre + "patch_embed.proj.weight"]
    D, _, ps, _ = w.shape
    g = R // ps
    N = g * g
    # 16x16/stride-16 conv == per-patch linear over (c, i, j)
    patches = video.reshape(B * T, C, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * T * N, C * ps * ps)
    tok = patches @ w.reshape(D, -1).t() + p[pre + "patch_embed.proj.bias"]
    tok = tok.reshape(B, T * N, D)
    pos = p[pre + "pos_embed"]                      # [1, N+1, D]
    tem = p[pre + "temporal_embed"]                 # [1, Tmax, D]
    table = pos[:, 1:].unsqueeze(1) + tem[:, :T].unsqueeze(2)            # [1,T,N,D]
    tok = tok + table.reshape(1, T * N, D)
    cls = (p[pre + "cls_token"] + pos[:, :1]).expand(B, 1, D)
    return torch.cat([cls, tok], dim=1), T, N


def video_encoder(video, p, num_heads=12, depth=None, pre="video_model.", return_blocks=False):
    """SpaceTimeTransformer.forward_features -> (cls [B,D], patches [B,T*N,D])
    (video_transformer.py:303-351; head/pre_logits are Identity, oa_model.py:50-51)."""
    x, T, N = video_tokens(video, p, pre)
    if depth is None:
        depth = 1 + max(int(k[len(pre) + 7:].split(".")[0]) for k in p if k.startswith(pre + "blocks."))
    per_block = []
    for i in range(depth):
        x = space_time_block(x, p, i, T, N, num_heads, pre)
        if return_blocks:
            per_block.append(x)
    x = _ln(x, p, pre + "norm", 1e-6)
    if return_blocks:
        return x[:, 0], x[:, 1:], per_block
    return x[:, 0], x[:, 1:]


# --------------------------------------------------------------------------- text encoder
def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123).
    counter: uint64 array [..., 4] of 32-bit words, key: [..., 2].  numpy, vectorised.  Pinned by Random123's
    known-answer vectors in tests/test_oracle_cpu.py; the HIP generator (csrc/rng.h) is checked against this one."""
    import numpy as np
    c = np.asarray(counter, dtype=np.uint64) & np.uint64(0xffffffff)
    k = np.asarray(key, dtype=np.uint64) & np.uint64(0xffffffff)
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    m32 = np.uint64(0xffffffff)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & m32, p0 & m32
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return np.stack([c0, c1, c2, c3], axis=-1)


def dropout_multipliers(n, p_drop, seed, offset, site):
    """The dropout multipliers (0 or 1/(1-p)) of elements [0, n) of a mask site, as the HIP kernels draw them
    (csrc/rng.h): element idx takes word idx % 4 of philox(counter = (idx//4 lo, idx//4 hi, site, offset lo),
    key = (seed lo, seed hi)) and is dropped when the word < p * 2^32."""
    import numpy as np
    q = np.arange((n + 3) // 4, dtype=np.uint64)
    ctr = np.stack([q & np.uint64(0xffffffff), q >> np.uint64(32), np.full_like(q, site), np.full_like(q, offset & 0xffffffff)], axis=-1)
    key = np.broadcast_to(np.array([seed & 0xffffffff, (seed >> 32) & 0xffffffff], dtype=np.uint64), (len(q), 2))
    words = philox4x32_10(ctr, key).reshape(-1)[:n]
    thresh = min(int(p_drop * 4294967296.0), 4294967295)
    return torch.from_numpy(np.where(words < np.uint64(thresh), 0.0, 1.0 / (1.0 - np.float32(p_drop))).astype(np.float32))


def distilbert_dropout_masks(B, L, D, n_heads, n_layers, p_hidden, p_attn, seed, offset):
    """Every training-mode mask of one DistilBERT forward, keyed as distilbert(dropout=...) expects; site numbering of
    engine/text.py: 0 = embeddings, 1 + 2i = attention probabilities of layer i, 2 + 2i = its ffn output."""
    m = {"emb": dropout_multipliers(B * L * D, p_hidden, seed, offset, 0).view(B, L, D)}
    for i in range(n_layers):
        m["attn", i] = dropout_multipliers(B * n_heads * L * L, p_attn, seed, offset, 1 + 2 * i).view(B, n_heads, L, L)
        m["ffn", i] = dropout_multipliers(B * L * D, p_hidden, seed, offset, 2 + 2 * i).view(B, L, D)
    return m


def distilbert(input_ids, attention_mask, p, n_heads=12, pre="text_model.", dropout=None):
    """HF DistilBertModel.forward(...).last_hidden_state.  dropout=None: eval mode.  dropout = dict of multiplier
    tensors {"emb": [B,L,D], ("attn", i): [B,H,L,L], ("ffn", i): [B,L,D]}: training mode with THESE masks at HF's
    three nn.Dropout sites (Embeddings.forward after LayerNorm; MultiHeadSelfAttention on the softmax weights;
    FFN after lin2 - transformers/models/distilbert/modeling_distilbert.py).
    Third-party code (transformers); restated from its published algorithm and
    validated numerically against transformers 5.15 eager attention in
    tests/golden/make_golden.py."""
    B, L = input_ids.shape
    x = p[pre + "embeddings.word_embeddings.weight"][input_ids] \
        + p[pre + "embeddings.position_embeddings.weight"][:L].unsqueeze(0)
    x = _ln(x, p, pre + "embeddings.LayerNorm", 1e-12)
    if dropout is not None:
        x = x * dropout["emb"]
    D = x.shape[-1]
    d = D // n_heads
    n_layers = 1 + max(int(k[len(pre) + 18:].split(".")[0]) for k in p if k.startswith(pre + "transformer.layer."))
    neg = torch.finfo(x.dtype).min
    keep = attention_mask.to(torch.bool)[:, None, None, :]                 # [B,1,1,L]
    for i in range(n_layers):
        b = f"{pre}transformer.layer.{i}."
        q = _lin(x, p, b + "attention.q_lin").reshape(B, L, n_heads, d).transpose(1, 2)
        k = _lin(x, p, b + "attention.k_lin").reshape(B, L, n_heads, d).transpose(1, 2)
        v = _lin(x, p, b + "attention.v_lin").reshape(B, L, n_heads, d).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(d)
        s = s.masked_fill(~keep, neg)
        w = torch.softmax(s, dim=-1)
        if dropout is not None:
            w = w * dropout["attn", i]
        ctx = (w @ v).transpose(1, 2).reshape(B, L, D)
        x = _ln(_lin(ctx, p, b + "attention.out_lin") + x, p, b + "sa_layer_norm", 1e-12)
        f = _lin(F.gelu(_lin(x, p, b + "ffn.lin1")), p, b + "ffn.lin2")
        if dropout is not None:
            f = f * dropout["ffn", i]
        x = _ln(f + x, p, b + "output_layer_norm", 1e-12)
    return x


# --------------------------------------------------------------------------- model assembly / loss
def frozen_forward(p, video, input_ids, attention_mask, num_heads=12, text_heads=12):
    """oa_model.FrozenInTime.forward (oa_model.py:97-133):
    text = txt_proj(ReLU -> Linear)(DistilBERT token 0); video = vid_proj(Linear)(CLS)."""
    t = distilbert(input_ids, attention_mask, p, n_heads=text_heads)[:, 0]
    t = F.linear(F.relu(t), p["txt_proj.1.weight"], p["txt_proj.1.bias"])
    cls, _ = video_encoder(video, p, num_heads=num_heads)
    v = F.linear(cls, p["vid_proj.0.weight"], p["vid_proj.0.bias"])
    return t, v


def sim_matrix(a, b, eps=1e-8):
    """oa_model.py:192-200: rows L2-normalised with the norm clamped to >= eps."""
    an = a / a.norm(dim=1, keepdim=True).clamp_min(eps)
    bn = b / b.norm(dim=1, keepdim=True).clamp_min(eps)
    return an @ bn.t()


def norm_softmax_loss(x, temperature=0.05):
    """loss.py:13-25: -(mean diag log_softmax(x/t, rows) + mean diag log_softmax(x^T/t, rows))."""
    i = torch.log_softmax(x / temperature, dim=1)
    j = torch.log_softmax(x.t() / temperature, dim=1)
    n = x.shape[0]
    return -(torch.diagonal(i).sum() / n) - (torch.diagonal(j).sum() / n)


def allgather_multi_sim(local_list, rank):
    """Single-process model of AllGather_multi (trainer_dist.py:29-45) for rank
    `rank`: forward = concat of every rank's tensor in rank order; backward lets
    gradient reach ONLY this rank's slice (other slices are constants here)."""
    parts = [t if r == rank else t.detach() for r, t in enumerate(local_list)]
    return torch.cat(parts, dim=0)


def train_step_loss(p, video, input_ids, attention_mask, num_heads=12, temperature=0.05, text_heads=12):
    """One rank, world_size 1: trainer_dist.py:158-162."""
    t, v = frozen_forward(p, video, input_ids, attention_mask, num_heads, text_heads)
    sim = sim_matrix(t, v)
    return norm_softmax_loss(sim, temperature), sim, t, v


# --------------------------------------------------------------------------- OA extras
def mask_pool(masks, feats):
    """einsum('b o l, b l c -> b o c') (oa_model_global_local.py:178,200)."""
    return masks @ feats


def region_sim(text_regions, object_regions):
    """sigmoid(einsum('b k f, b n f -> b k n')) (oa_model_region_mem.py:150-151)."""
    return torch.sigmoid(text_regions @ object_regions.transpose(1, 2))


def gl_tail(x_normed):
    """oa_video_transformer_global_local.py:356-359: (1/2 CLS + 1/2 mean patches, patches)."""
    return 0.5 * x_normed[:, 0] + 0.5 * x_normed[:, 1:].mean(dim=1), x_normed[:, 1:]


# --------------------------------------------------------------------------- OA variants
def _encoder_stream(video, p, num_heads, pre, tap=None):
    x, T, N = video_tokens(video, p, pre)
    depth = 1 + max(int(k[len(pre) + 7:].split(".")[0]) for k in p if k.startswith(pre + "blocks."))
    tapped = None
    for i in range(depth):
        x = space_time_block(x, p, i, T, N, num_heads, pre)
        if tap is not None and i + 1 == tap:
            tapped = x
    return x, tapped


def video_encoder_region(video, p, num_heads=12, pre="video_model.", region_layer=6):
    """oa_video_transformer_region.py:364-376: (norm(x)[:,0], region_norm(x after block 6)[:,1:])."""
    x, tapped = _encoder_stream(video, p, num_heads, pre, tap=region_layer)
    return _ln(x, p, pre + "norm", 1e-6)[:, 0], _ln(tapped, p, pre + "region_norm", 1e-6)[:, 1:]


def video_encoder_gl(video, p, num_heads=12, pre="video_model."):
    """oa_video_transformer_global_local.py:356-359."""
    x, _ = _encoder_stream(video, p, num_heads, pre)
    return gl_tail(_ln(x, p, pre + "norm", 1e-6))


def _object_and_video_clips(video, encode, object_clip):
    """The clip layouts of the object-aware models.  'interleaved' is the reference's own
    (oa_model_global_local.py:170 / oa_model_region_mem.py:109): [B, F] frames viewed as 2B clips of F/2 frames, even
    clips = object images, odd clips = videos.  'native' (BASELINE.json config 3, "8-frame + 10 obj"; no reference
    line - the reference's view() cannot express it) takes frame 0 as a one-frame object clip and frames 1..T as the
    video clip and encodes both with the same weights; everything downstream is unchanged.  At F = 2 the two
    layouts are the same computation (tests/test_oracle_oa_golden.py).  `encode(clips) -> (emb, region)`."""
    if object_clip == "interleaved":
        B = video.shape[0]
        emb, region = encode(video.reshape(B * 2, -1, *video.shape[2:]))
        return emb[0::2], region[0::2], emb[1::2], region[1::2]
    assert object_clip == "native", object_clip
    obj_emb, obj_region = encode(video[:, :1])
    vid_emb, vid_region = encode(video[:, 1:])
    return obj_emb, obj_region, vid_emb, vid_region


def _relu_lin(x, p, name):
    return F.linear(F.relu(x), p[name + ".1.weight"], p[name + ".1.bias"])


def region_mem_forward(p, video, input_ids, attention_mask, text_region_embedding, num_heads=12, text_heads=12,
                       object_clip="interleaved"):
    """oa_model_region_mem.FrozenInTime.forward (oa_model_region_mem.py:105-123,141-151)."""
    t = _relu_lin(distilbert(input_ids, attention_mask, p, n_heads=text_heads)[:, 0], p, "txt_proj")
    proj = lambda z: F.linear(z, p["vid_proj.0.weight"], p["vid_proj.0.bias"])

    def encode(v):
        cls, region = video_encoder_region(v, p, num_heads)
        return proj(cls), proj(region)
    _, obj_region, vid_emb, vid_region = _object_and_video_clips(video, encode, object_clip)
    treg = _relu_lin(text_region_embedding, p, "txt_proj_2")
    video_emb = (vid_emb + vid_region.mean(dim=1)) / 2
    return t, video_emb, region_sim(treg, obj_region)


def region_mem_loss(t, v, rsim, patch_mask, temperature=0.05):
    """trainer_region_mem.py:151-167: NCE + 0.1 * BCE_sum / rows."""
    loss = norm_softmax_loss(sim_matrix(t, v), temperature)
    rs = rsim.reshape(-1, rsim.shape[-1])
    pm = patch_mask.reshape(-1, patch_mask.shape[-1]).float()
    return loss + 0.1 * F.binary_cross_entropy(rs, pm, reduction="sum") / rs.shape[0]


def tag_masks(object_token_masks, n_txt, L):
    """oa_model_global_local.py:183-196 (Python double loop in the reference): tag k of sample j covers
    pad-text positions [n_txt-1+end_{k-1}, n_txt-1+end_k)."""
    ends = object_token_masks.long()
    starts = torch.cat([torch.zeros_like(ends[:, :1]), ends[:, :-1]], dim=1)
    pos = torch.arange(L)[None, None, :]
    base = (n_txt.long() - 1)[:, None, None]
    return ((pos >= base + starts[:, :, None]) & (pos < base + ends[:, :, None])).float()


def gl_forward(p, video, text, pad_text, patch_masks, object_token_masks, num_heads=12, text_heads=12,
               object_clip="interleaved"):
    """oa_model_global_local.FrozenInTime.forward (oa_model_global_local.py:149-208, :210-221)."""
    def compute_text(ids, mask):
        h = distilbert(ids, mask, p, n_heads=text_heads)
        return _relu_lin(h[:, 0] + h[:, 1:].mean(dim=1), p, "txt_proj"), h
    t, ttok = compute_text(*text)
    pt, ptok = compute_text(*pad_text)

    def encode(v):
        emb, region = video_encoder_gl(v, p, num_heads)
        return F.linear(emb, p["vid_proj.0.weight"], p["vid_proj.0.bias"]), region
    obj_emb, obj_region, vid_emb, vid_region = _object_and_video_clips(video, encode, object_clip)
    region_feat = mask_pool(patch_masks.float(), obj_region)
    tm = tag_masks(object_token_masks, text[1].sum(dim=1), ptok.shape[1])
    tags_feat = mask_pool(tm, ptok)
    region_feat = F.linear(region_feat, p["vid_local_proj.0.weight"], p["vid_local_proj.0.bias"])
    tags_feat = _relu_lin(tags_feat, p, "text_local_proj")
    return t, pt, vid_emb, obj_emb, region_feat, tags_feat


def gl_loss(t, pt, v, region_feat, tags_feat, temperature=0.05):
    """trainer_global_local.py:187-211."""
    return (norm_softmax_loss(sim_matrix(t, v), temperature) + norm_softmax_loss(sim_matrix(pt, v), temperature)
            + norm_softmax_loss(sim_matrix(region_feat.mean(dim=1), tags_feat.mean(dim=1)), temperature))


def patch_masks_from_bbox(bboxs, patch_rows=14, box_class=None, sel_class=None):
    """bbox -> patch-grid masks, float32 [O, patch_rows**2].

    bboxs [NB, >=4] = (x0, y0, x1, y1) normalised to the frame.  Restates
      /root/reference/OATrans/base/base_dataset_global_local.py:348-356  (one mask per box: box_class is None)
      /root/reference/OATrans/base/base_dataset_region_mem.py:233-247    (mask j = union of the boxes whose class is
                                                                         sel_class[j]; the random choice of the 5
                                                                         classes stays with the caller)
    including numpy's slice semantics (int() truncation of the start, ceil of the stop, clamping to the grid)."""
    import math
    b = bboxs[:, :4].to(torch.float32) * patch_rows
    nb = b.shape[0]
    out_n = nb if box_class is None else len(sel_class)
    m = torch.zeros(out_n, patch_rows, patch_rows)
    for o in range(out_n):
        for i in range(nb):
            if box_class is not None and int(box_class[i]) != int(sel_class[o]):
                continue
            if box_class is None and i != o:
                continue
            x0, y0, x1, y1 = (float(v) for v in b[i])
            m[o, int(y0):math.ceil(y1), int(x0):math.ceil(x1)] = 1
    return m.reshape(out_n, patch_rows * patch_rows)
