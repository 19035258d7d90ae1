"""Full hot-path parity on a real MI355X: text encoder, projections, sim_matrix, loss, optimiser
and the contract class FrozenInTime against golden vectors captured from the reference.

Stated tolerances: cosine-similarity matrix max-abs error <= 1e-3 (north_star); embeddings
relative L2 <= 1e-2; parameter gradients relative L2 <= 5e-2 with cosine >= 0.998 (bf16 GEMM
operands, fp32 accumulation)."""
import os

import pytest
import torch

from OATrans.utils import seeded_init as si

pytestmark = pytest.mark.gpu
SEED = 20240917
SMALL_VIDEO = dict(embed_dim=128, depth=2, mlp_ratio=4, num_frames=3, patches_per_frame=9, patch=16)
SMALL_TEXT = dict(dim=128, n_layers=2, hidden_dim=512, vocab=1000, max_pos=64)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def cosine(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-20)).item()


def _golden(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), map_location="cpu", weights_only=False)


def check_grads(named_params, ref_grads, prefix="", tol=5e-2, cos_min=0.998):
    worst = ("", 0.0)
    for k, ref in ref_grads.items():
        if not k.startswith(prefix):
            continue
        p = named_params[k[len(prefix):]]
        assert p.grad is not None, k
        if ref.norm() < 1e-6:                  # analytically zero (k_lin.bias): both sides round-off
            assert p.grad.norm() < 1e-3 * max(1.0, ref.numel() ** 0.5), k
            continue
        e, c = rel(p.grad, ref), cosine(p.grad, ref)
        if e > worst[1]:
            worst = (k, e)
        assert e < tol and c > cos_min, (k, e, c)
    return worst


def test_small_chain_vs_reference_golden(golden_dir):
    from OATrans.model.layers import HipLinear, ReLULinear, sim_matrix
    from OATrans.model.loss import NormSoftmaxLoss
    from OATrans.model.text_transformer import DistilBertHIP
    from OATrans.model.video_transformer import SpaceTimeTransformer
    g = _golden(golden_dir, "small_chain.pt")
    sd = si.frozen_state_dict(SEED, SMALL_VIDEO, SMALL_TEXT, proj_dim=64)
    txt = DistilBertHIP(dict(vocab_size=1000, max_position_embeddings=64, n_layers=2, n_heads=2, dim=128, hidden_dim=512))
    txt.eval()                      # the golden is an eval-mode run (dropout off)
    r = txt.load_state_dict({k[len("text_model."):]: v for k, v in sd.items() if k.startswith("text_model.")})
    vid = SpaceTimeTransformer(img_size=48, patch_size=16, embed_dim=128, depth=2, num_heads=2, num_frames=3, time_init="rand")
    vid.head = torch.nn.Identity()
    vid.load_state_dict({k[len("video_model."):]: v for k, v in sd.items() if k.startswith("video_model.")}, strict=False)
    vid.need_patch_tokens = False
    txt_proj, vid_proj = ReLULinear(128, 64), HipLinear(128, 64)
    txt_proj[1].load_state_dict({"weight": sd["txt_proj.1.weight"], "bias": sd["txt_proj.1.bias"]})
    vid_proj.load_state_dict({"weight": sd["vid_proj.0.weight"], "bias": sd["vid_proj.0.bias"]})
    txt, vid, txt_proj, vid_proj = txt.cuda(), vid.cuda(), txt_proj.cuda(), vid_proj.cuda()
    video = si.seeded_tensor(SEED, "in.video", (4, 3, 3, 48, 48)).cuda()
    ids = si.seeded_ints(SEED, "in.ids", (4, 7), 1, 1000).cuda()
    mask = g["mask"].cuda()
    txt.begin_step()
    hidden = txt(input_ids=ids, attention_mask=mask).last_hidden_state
    keep = mask.bool().cpu()
    assert rel(hidden.cpu()[keep], g["text_hidden"][keep]) < 1e-2        # padded positions: don't-care
    t = txt_proj(hidden[:, 0])
    cls, _ = vid(video)
    v = vid_proj(cls)
    assert rel(t, g["text"]) < 1e-2 and rel(v, g["video"]) < 1e-2, (rel(t, g["text"]), rel(v, g["video"]))
    sim = sim_matrix(t, v)
    # 64-d embeddings of the toy geometry average bf16 noise over 4x fewer dims than the real
    # 256-d ones, so the cos-sim bound here is 4e-3; the 1e-3 bar is asserted at ViT-B geometry below
    assert (sim.cpu() - g["sim"]).abs().max() <= 4e-3
    loss = NormSoftmaxLoss()(sim)
    assert abs(loss.item() - g["loss"].item()) < 5e-2 * max(1.0, abs(g["loss"].item()))
    loss.backward()
    torch.cuda.synchronize()
    w1 = check_grads(dict(vid.named_parameters()), g["grads"], "video_model.")
    w2 = check_grads(dict(txt.named_parameters()), g["grads"], "text_model.", tol=1.5e-1, cos_min=0.99)   # toy dims (128-d, 7 tokens): noisier
    check_grads({"1.weight": txt_proj[1].weight, "1.bias": txt_proj[1].bias}, g["grads"], "txt_proj.")
    check_grads({"0.weight": vid_proj.weight, "0.bias": vid_proj.bias}, g["grads"], "vid_proj.")
    print("worst grad errors", w1, w2)


@pytest.mark.parametrize("name,n", [("sq8", 8), ("sq1", 1), ("sq33", 33)])
def test_loss_cases_vs_reference_golden(golden_dir, name, n):
    from OATrans.model.layers import sim_matrix
    from OATrans.model.loss import NormSoftmaxLoss
    from OATrans.ops import hip
    g = _golden(golden_dir, "loss_cases.pt")[name]
    a = si.seeded_tensor(SEED, f"loss.a.{name}", (n, 16))
    b = si.seeded_tensor(SEED, f"loss.b.{name}", (n, 16))
    if name == "sq8":
        a[2] = 0.0
    a, b = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    sim = sim_matrix(a, b)
    loss = NormSoftmaxLoss()(sim)
    loss.backward()
    assert torch.allclose(sim.cpu(), g["sim"], atol=1e-5)
    assert torch.allclose(loss.cpu(), g["loss"], atol=1e-4, rtol=1e-5)
    assert torch.allclose(a.grad.cpu(), g["ga"], atol=1e-4, rtol=1e-3)
    assert torch.allclose(b.grad.cpu(), g["gb"], atol=1e-4, rtol=1e-3)
    # fused trainer path gives the same numbers
    l2, s2, dt, dv = hip.infonce(a.detach(), b.detach(), want_sim=True)
    assert torch.allclose(s2.cpu(), g["sim"], atol=1e-5) and torch.allclose(l2.cpu()[0], g["loss"], atol=1e-4)
    assert torch.allclose(dt.cpu(), g["ga"], atol=1e-4, rtol=1e-3) and torch.allclose(dv.cpu(), g["gb"], atol=1e-4, rtol=1e-3)


def test_infonce_local_rows_match_allgather_semantics():
    """AllGather_multi.backward keeps only the local slice (trainer_dist.py:41-45): the fused kernel's
    (r0, nloc) gradients equal the corresponding rows of the full gradient."""
    from OATrans.ops import hip
    t = torch.randn(24, 32, device="cuda")
    v = torch.randn(24, 32, device="cuda")
    _, _, dt, dv = hip.infonce(t, v)
    for r0, nloc in ((0, 8), (8, 8), (16, 8)):
        _, _, dtl, dvl = hip.infonce(t, v, r0=r0, nloc=nloc)
        assert torch.equal(dtl, dt[r0:r0 + nloc]) and torch.equal(dvl, dv[r0:r0 + nloc])


@pytest.mark.parametrize("hf_style", [True, False])
def test_adamw_matches_published_algorithms(hf_style):
    from OATrans.ops import hip
    n = 10007
    p0 = torch.randn(n, device="cuda")
    p = p0.clone()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    lr, b1, b2, eps, wd = 2e-4, 0.9, 0.999, 1e-6, 0.01
    pr, mr, vr = p0.double().clone(), torch.zeros(n, dtype=torch.double, device="cuda"), torch.zeros(n, dtype=torch.double, device="cuda")
    opt = None if hf_style else torch.optim.AdamW([torch.nn.Parameter(p0.clone())], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    for step in range(1, 4):
        g = torch.randn(n, device="cuda")
        hip.adamw(p, g, m, v, lr, b1, b2, eps, wd, step, hf_style=hf_style)
        if hf_style:                       # transformers.AdamW (4.x): eps outside the bias correction, decay after
            mr = b1 * mr + (1 - b1) * g.double()
            vr = b2 * vr + (1 - b2) * g.double() ** 2
            pr = pr - lr * (1 - b2 ** step) ** 0.5 / (1 - b1 ** step) * mr / (vr.sqrt() + eps)
            pr = pr - lr * wd * pr
        else:
            opt.param_groups[0]["params"][0].grad = g.clone()
            opt.step()
            pr = opt.param_groups[0]["params"][0].detach().double()
    assert torch.allclose(p.double(), pr, atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("prune_top", [False, True])
@pytest.mark.parametrize("frames", [1, 4, 8])
def test_frozen_in_time_vitb_vs_reference_golden(golden_dir, frames, prune_top):
    """prune_top: the same goldens with the top block's unused patch rows skipped (VideoEngine.prune_top, opt-in).

    The contract class at ViT-B/16 + DistilBERT-base geometry against the outputs of the reference's own
    oa_model.FrozenInTime (tests/golden/full_T{1,4,8}.pt): 1 frame = BASELINE config 1's geometry, 4 frames = config 2's,
    8 frames = the headline shape of configs 3 / 4.  Stated tolerance (north_star): sim matrix <= 1e-3 max-abs."""
    from OATrans import model as module_arch
    g = _golden(golden_dir, f"full_T{frames}.pt")
    T, B, L = g["T"], g["B"], g["L"]
    m = module_arch.FrozenInTime(
        video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=T, pretrained=True, time_init="rand"),
        object_params=dict(model="", input_objects=False),
        text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text"),
        projection="minimal", load_checkpoint="")
    m.text_model.eval()          # parity runs in eval mode (the goldens' DistilBERT has dropout off)
    r = m.load_state_dict(si.frozen_state_dict(SEED, dict(num_frames=T), {}), strict=False)
    assert not r.unexpected_keys and not r.missing_keys, r
    m = m.cuda()
    m.video_model._engine.prune_top = prune_top
    video = si.seeded_tensor(SEED, f"full.video.{T}", (B, T, 3, 224, 224)).cuda()
    ids = si.seeded_ints(SEED, f"full.ids.{T}", (B, L), 1000, 30000)
    ids[:, 0] = 101
    m.begin_step()
    t, v = m({"video": video, "text": {"input_ids": ids.cuda(), "attention_mask": g["mask"].cuda()}})
    sim = module_arch.sim_matrix(t, v)
    loss = module_arch.NormSoftmaxLoss()(sim)
    loss.backward()
    torch.cuda.synchronize()
    sim_err = (sim.cpu() - g["sim"]).abs().max().item()
    print("text rel", rel(t, g["text"]), "video rel", rel(v, g["video"]), "sim err", sim_err, "loss", loss.item(), g["loss"].item())
    assert rel(t, g["text"]) < 1e-2 and rel(v, g["video"]) < 1e-2
    assert sim_err <= 1e-3
    assert abs(loss.item() - g["loss"].item()) < 2e-2
    params = dict(m.named_parameters())
    bad = []
    for k, pr in g["grad_probe"].items():
        gr = params[k].grad
        assert gr is not None, k
        if pr["norm"] < 1e-6:
            continue
        nerr = abs(gr.norm().item() - pr["norm"].item()) / pr["norm"].item()
        scale = pr["norm"].item() / gr.numel() ** 0.5
        perr = ((gr.flatten()[pr["idx"].cuda()].cpu() - pr["val"]).abs() / scale).max().item()
        # Eight sampled entries per tensor, in units of the tensor RMS: a SANITY bound (a wrong kernel is off by tens of RMS), not the
        # parity statement - the goldens keep only these samples of the reference's gradients.  The per-tensor rel-L2 / cosine bounds
        # are asserted on EVERY element against the oracle's autograd (test_four_frame_geometry_every_gradient_vs_oracle_autograd, test_headline_geometry_every_gradient_vs_oracle_autograd; the oracle
        # is pinned to the reference by tests/test_oracle_golden.py).  Why not 1.0 any more (round 6): pos_embed's RMS is dominated by
        # its CLS row (14 x the patch rows; the sampled row-0 entry is -2.3 RMS inside a row of RMS 14), so a 3 % rel-L2 error puts that
        # one entry anywhere within +- 1.3 RMS: 0.22 with round 5's GELU arithmetic, 1.21 with round 6's, at the same rel-L2 (2.95e-2
        # / 3.01e-2 for pos_embed, scripts/dev/grad_rel_probe.py).
        if nerr > 5e-2 or perr > 2.0:
            bad.append((k, nerr, perr))
    assert not bad, bad[:10]


_HEADLINE_ORACLE = {}


def _headline_oracle_grads(T=8):
    """Loss and every parameter gradient of the T-frame frozen model at B = 2 from autograd of the CPU oracle (computed once per session)."""
    if T not in _HEADLINE_ORACLE:
        from oracle import oatrans_oracle as orc
        B, L = 2, 12
        sd = si.frozen_state_dict(SEED, dict(num_frames=T), {})
        video = si.seeded_tensor(SEED, f"full.video.{T}", (B, T, 3, 224, 224))
        ids = si.seeded_ints(SEED, f"full.ids.{T}", (B, L), 1000, 30000)
        ids[:, 0] = 101
        mask = torch.ones(B, L, dtype=torch.int64)
        mask[1, 9:] = 0
        p = {k: (w.clone().requires_grad_(True) if w.is_floating_point() else w) for k, w in sd.items()}
        oloss, _, _, _ = orc.train_step_loss(p, video, ids, mask)
        oloss.backward()
        _HEADLINE_ORACLE[T] = dict(sd=sd, video=video, ids=ids, mask=mask, loss=oloss.item(),
                                   grads={k: w.grad for k, w in p.items() if w.is_floating_point() and w.grad is not None})
    return _HEADLINE_ORACLE[T]


def _headline_gradient_errors(prune_top=False, res16=None, h_u8=None, frames=8):
    """One forward + backward of the HIP model at the headline geometry (B = 2; `frames` frames) against the oracle's autograd.
    Returns (loss error, [(name, rel-L2, cosine)], all-parameter rel-L2, all-parameter cosine, the same two over the video tower alone)."""
    from OATrans import model as module_arch
    o = _headline_oracle_grads(frames)
    m = module_arch.FrozenInTime(
        video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=frames, pretrained=True, time_init="rand"),
        object_params=dict(model="", input_objects=False),
        text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text"),
        projection="minimal", load_checkpoint="")
    m.text_model.eval()
    r = m.load_state_dict(o["sd"], strict=False)
    assert not r.unexpected_keys and not r.missing_keys, r
    m = m.cuda()
    eng = m.video_model._engine
    eng.prune_top = prune_top
    if res16 is not None:
        eng.res16 = res16
    if h_u8 is not None:
        eng.h_u8 = h_u8
    m.begin_step()
    t, v = m({"video": o["video"].cuda(), "text": {"input_ids": o["ids"].cuda(), "attention_mask": o["mask"].cuda()}})
    loss = module_arch.NormSoftmaxLoss()(module_arch.sim_matrix(t, v))
    loss.backward()
    torch.cuda.synchronize()
    assert eng.plans and all(pl.prune_top == prune_top for pl in eng.plans.values())
    if res16 is not None:
        assert all(pl.res16 == res16 for pl in eng.plans.values())
    if h_u8 is not None:
        assert all(pl.h_u8 == h_u8 for pl in eng.plans.values())
    per, acc = [], {"all": [0.0] * 5, "video": [0.0] * 5}
    for k, prm in m.named_parameters():
        ref = o["grads"].get(k)
        if ref is None or ref.norm().item() < 1e-6:      # e.g. the k_lin biases: softmax is invariant to them, the gradient is rounding noise
            continue
        assert prm.grad is not None, k
        mine = prm.grad.float().cpu()
        e = ((mine - ref).norm() / ref.norm()).item()
        c = torch.nn.functional.cosine_similarity(mine.flatten(), ref.flatten(), dim=0).item()
        per.append((k, e, c))
        terms = [(mine - ref).pow(2).sum().item(), ref.pow(2).sum().item(), (mine * ref).sum().item(), mine.pow(2).sum().item(), ref.pow(2).sum().item()]
        for key in ("all",) + (("video",) if k.startswith("video_model.") else ()):
            acc[key] = [x + y for x, y in zip(acc[key], terms)]
    rel = lambda a: (a[0] / a[1]) ** 0.5
    cos = lambda a: a[2] / (a[3] * a[4]) ** 0.5
    return abs(loss.item() - o["loss"]), per, rel(acc["all"]), cos(acc["all"]), rel(acc["video"]), cos(acc["video"])


@pytest.mark.parametrize("prune_top", [False, True])
def test_four_frame_geometry_every_gradient_vs_oracle_autograd(prune_top):
    """The same all-element check at 4 frames (config 2's geometry; half the rows to average the bf16 noise over): per tensor
    rel-L2 <= 5e-2 and cosine >= 0.998, all parameters rel-L2 <= 3.5e-2 (measured 3.1e-2; 2.2e-2 at 8 frames)."""
    dloss, per, all_rel, all_cos, _, _ = _headline_gradient_errors(prune_top=prune_top, frames=4)
    assert dloss < 2e-2
    worst_rel, worst_cos = max(per, key=lambda z: z[1]), min(per, key=lambda z: z[2])
    print(f"4 frames: worst rel-L2 {worst_rel[:2]}, worst cosine {(worst_cos[0], worst_cos[2])}, all parameters rel-L2 {all_rel:.3e} cosine {all_cos:.6f}")
    bad = [z for z in per if z[1] > 5e-2 or z[2] < 0.998]
    assert not bad, bad[:10]
    assert all_rel <= 3.5e-2 and all_cos >= 0.9993, (all_rel, all_cos)


@pytest.mark.parametrize("prune_top", [False, True])
def test_headline_geometry_every_gradient_vs_oracle_autograd(prune_top):
    """prune_top: the same check with the top block's unused patch rows skipped (engine/video.py: VideoEngine.prune_top) -
    the reference computes them and discards them (video_transformer.py:349-351, oa_model.py:129-133), so every gradient must
    still match the oracle's autograd, which runs the full graph.

    Every parameter gradient of the 8-frame frozen model (ViT-B/16 + DistilBERT-base, the headline geometry) at B = 2
    against autograd of the CPU oracle on the same seeded inputs: per-tensor relative L2 and cosine, and the same two figures
    over all parameters at once.  Stated tolerance (bf16 operands, bf16 residual / gradient stream, 8-bit GELU derivative
    against an fp32 reference): per tensor rel-L2 <= 5e-2 and cosine >= 0.998, all parameters rel-L2 <= 3e-2 and
    cosine >= 0.9995 (measured values are printed; scripts/dev/rounding_study3.py predicts 2.2e-2 / 0.9998)."""
    dloss, per, all_rel, all_cos, _, _ = _headline_gradient_errors(prune_top=prune_top)
    assert dloss < 2e-2
    worst_rel, worst_cos = max(per, key=lambda z: z[1]), min(per, key=lambda z: z[2])
    print(f"worst rel-L2 {worst_rel[:2]}, worst cosine {(worst_cos[0], worst_cos[2])}, all parameters rel-L2 {all_rel:.3e} cosine {all_cos:.6f}")
    bad = [z for z in per if z[1] > 5e-2 or z[2] < 0.998]
    assert not bad, bad[:10]
    assert all_rel <= 3e-2 and all_cos >= 0.9995, (all_rel, all_cos)


def test_the_two_storage_departures_are_what_the_gradient_error_is_made_of():
    """The engine departs from the fp32 reference in two STORAGE choices a reader of the reference would not expect:
    the residual stream / residual-gradient stream of the video tower are kept as bf16 (`OAT_RES16=0` / `VideoEngine.res16 = False`: fp32),
    and the saved GELU derivative is 8-bit fixed point (`OAT_H_U8=0` / `VideoEngine.h_u8 = False`: bf16).  With both switched off the same
    all-gradient check must come out closer to the oracle's autograd - the departures, not something else, are what the default's error
    above the plain bf16-operand pipeline is made of - and each one alone must sit between the two."""
    runs = {}
    for name, (r16, u8) in {"default": (True, True), "fp32 stream": (False, True), "bf16 derivative": (True, False), "both off": (False, False)}.items():
        dloss, per, all_rel, all_cos, vid_rel, vid_cos = _headline_gradient_errors(res16=r16, h_u8=u8)
        assert dloss < 2e-2
        runs[name] = (vid_rel, vid_cos, max(z[1] for z in per if z[0].startswith("video_model.")))
        print(f"{name:16s}: video tower rel-L2 {vid_rel:.3e} cosine {vid_cos:.6f} worst tensor {runs[name][2]:.3e}; all parameters {all_rel:.3e}")
    d, off = runs["default"], runs["both off"]
    assert off[0] < 0.85 * d[0], runs                                     # measurably closer with both departures off
    assert off[0] <= runs["fp32 stream"][0] * 1.03 and off[0] <= runs["bf16 derivative"][0] * 1.03, runs
    assert runs["fp32 stream"][0] <= d[0] * 1.03 and runs["bf16 derivative"][0] <= d[0] * 1.03, runs
    assert off[0] <= 3e-2 and off[1] >= 0.9995                            # and still a bf16-operand pipeline: not zero


def test_ragged_shapes_take_optimiser_steps():
    """Edge shapes the reference accepts: a single pair (B = 1: the reference itself trips an in-place-on-view error
    there), odd batches, one frame, one-token captions, changing shapes between steps."""
    import argparse
    from OATrans import model as module_arch
    from OATrans.optim import AdamW
    from OATrans.parallel import HipDataParallel
    from OATrans.trainer.step import hot_step
    torch.manual_seed(0)
    m = module_arch.FrozenInTime(
        video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=3, pretrained=True,
                          time_init="rand", arch_kwargs=dict(depth=2)),
        object_params=dict(model="", input_objects=False),
        text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text", config=dict(n_layers=1)),
        projection="minimal", load_checkpoint="").cuda()
    m.text_model.eval()          # parity runs in eval mode (the goldens' DistilBERT has dropout off)
    m.set_device(torch.device("cuda"))
    for sub in (m.video_model, m.text_model):
        sub.flatten_parameters()
    dp = HipDataParallel(m)
    opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-5)
    sa = argparse.Namespace(world_size=1, rank=0, local_rank=0)
    for B, T, L in ((1, 3, 5), (3, 2, 9), (5, 1, 32), (2, 3, 1)):
        g = torch.Generator().manual_seed(B)
        data = {"video": torch.randn(B, T, 3, 224, 224, generator=g).cuda(),
                "text": {"input_ids": torch.randint(1000, 30000, (B, L), generator=g).cuda(),
                         "attention_mask": torch.ones(B, L, dtype=torch.int64).cuda()}}
        losses = [hot_step(dp, module_arch.NormSoftmaxLoss(), opt, data, sa).item() for _ in range(2)]
        assert all(l == l and abs(l) < 1e3 for l in losses), (B, T, L, losses)
        if B > 1:
            assert losses[1] < losses[0], (B, T, L, losses)
    assert all(torch.isfinite(p).all().item() for p in m.parameters())


def test_eager_adamw_equals_step_after_backward():
    """AdamW.attach: ranges announced during backward are updated under it on a side stream; the parameters after
    each step must equal the plain backward(); step() sequence (trainer_dist.py:163-166).  Identical gradients give
    bit-identical updates (same kernel, same operands), so the only tolerance is the fp32-atomics noise of the
    CLS-row gradients between two runs of backward."""
    import argparse
    import copy
    from OATrans import model as module_arch
    from OATrans.optim import AdamW
    from OATrans.parallel import HipDataParallel
    from OATrans.trainer.step import hot_step
    torch.manual_seed(0)
    base = module_arch.FrozenInTime(
        video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=2, pretrained=True,
                          time_init="rand", arch_kwargs=dict(depth=3)),
        object_params=dict(model="", input_objects=False),
        text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text", config=dict(n_layers=2)),
        projection="minimal", load_checkpoint="")
    base.text_model.eval()          # parity runs in eval mode (the goldens' DistilBERT has dropout off)
    sa = argparse.Namespace(world_size=1, rank=0, local_rank=0)
    g = torch.Generator().manual_seed(5)
    B, T, L = 4, 2, 12
    data = {"video": torch.randn(B, T, 3, 224, 224, generator=g).cuda(),
            "text": {"input_ids": torch.randint(1000, 30000, (B, L), generator=g).cuda(),
                     "attention_mask": torch.ones(B, L, dtype=torch.int64).cuda()}}
    results = []
    for eager in (False, True):
        m = copy.deepcopy(base).cuda()
        m.set_device(torch.device("cuda"))
        for sub in (m.video_model, m.text_model):
            sub.flatten_parameters()
        dp = HipDataParallel(m)
        opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01)
        if eager:
            opt.attach(m)
        losses = [hot_step(dp, module_arch.NormSoftmaxLoss(), opt, data, sa).item() for _ in range(4)]
        torch.cuda.synchronize()
        # the first step builds the optimiser state in step(); from the second on every block and the text tower are eager
        assert opt.eager_launches == (3 * (3 + 1 + 1) if eager else 0), opt.eager_launches
        assert all(st['step'] == 4 for st in opt.state.values())
        results.append((losses, {n: p.detach().clone() for n, p in m.named_parameters()}))
    (l0, p0), (l1, p1) = results
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 5e-3, (l0, l1)      # losses of 3-4.5: relative 1e-3
    assert l0[-1] < l0[0]
    for n in p0:
        # 4 steps of at most lr each; an element whose gradient sign flips with the atomics noise moves by <= 2 lr per step
        d = (p0[n] - p1[n]).abs()
        assert d.max().item() <= 8e-4 + 1e-6, (n, d.max().item())
        assert d.mean().item() < 2e-5, (n, d.mean().item())


def _small_frozen(seed=0, frames=2, depth=2, n_layers=1):
    from OATrans import model as module_arch
    torch.manual_seed(seed)
    m = module_arch.FrozenInTime(
        video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=frames, pretrained=True,
                          time_init="rand", arch_kwargs=dict(depth=depth)),
        object_params=dict(model="", input_objects=False),
        text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text", config=dict(n_layers=n_layers)),
        projection="minimal", load_checkpoint="").cuda()
    m.text_model.eval()          # parity runs in eval mode (the goldens' DistilBERT has dropout off)
    m.set_device(torch.device("cuda"))
    for sub in (m.video_model, m.text_model):
        sub.flatten_parameters()
    return m


def _batch(B=4, T=2, L=12, seed=5):
    g = torch.Generator().manual_seed(seed)
    return {"video": torch.randn(B, T, 3, 224, 224, generator=g).cuda(),
            "text": {"input_ids": torch.randint(1000, 30000, (B, L), generator=g).cuda(),
                     "attention_mask": torch.ones(B, L, dtype=torch.int64).cuda()}}


def test_projection_heads_follow_torch_adamw_over_several_steps():
    """txt_proj / vid_proj are ordinary autograd leaves (their gradients ACCUMULATE unless zero_grad clears them, unlike
    the engine parameters whose flat gradient buffers every backward overwrites).  Three training steps: the
    projection weights must follow a torch.optim.AdamW (transformers-4.6 formula == torch's at weight_decay 0 up to
    eps placement) fed with the per-step gradients - not the running sum of all past gradients."""
    import argparse
    from OATrans import model as module_arch
    from OATrans.optim import AdamW
    from OATrans.parallel import HipDataParallel
    from OATrans.trainer.step import hot_step
    m = _small_frozen()
    loose = {n: p for n, p in m.named_parameters() if n.startswith(("txt_proj", "vid_proj"))}
    assert len(loose) == 4
    ref = {n: p.detach().double().clone() for n, p in loose.items()}
    mom = {n: (torch.zeros_like(r), torch.zeros_like(r)) for n, r in ref.items()}
    dp = HipDataParallel(m)
    lr, b1, b2, eps = 1e-4, 0.9, 0.999, 1e-6
    opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=lr)
    sa = argparse.Namespace(world_size=1, rank=0, local_rank=0)
    data = _batch()
    grads_seen = []
    for step in range(1, 4):
        hot_step(dp, module_arch.NormSoftmaxLoss(), opt, data, sa)
        torch.cuda.synchronize()
        grads_seen.append({n: p.grad.detach().double().clone() for n, p in loose.items()})
        for n in ref:                                        # transformers.AdamW: eps outside the bias correction
            g = grads_seen[-1][n]
            mm, vv = mom[n]
            mm.mul_(b1).add_(g, alpha=1 - b1)
            vv.mul_(b2).addcmul_(g, g, value=1 - b2)
            ref[n] -= lr * (1 - b2 ** step) ** 0.5 / (1 - b1 ** step) * mm / (vv.sqrt() + eps)
    for n, p in loose.items():
        assert torch.allclose(p.detach().double(), ref[n], atol=5e-7, rtol=1e-5), (n, (p.detach().double() - ref[n]).abs().max())
    # and the gradients of consecutive steps are per-step gradients: a running sum would grow ~linearly
    n0 = "vid_proj.0.weight"
    norms = [g[n0].norm().item() for g in grads_seen]
    assert norms[2] < 1.6 * norms[0], norms


def test_optimizer_resume_continues_the_trajectory():
    """save -> load -> step must equal uninterrupted stepping: AdamW.load_state_dict carries the loaded moments into
    the flat buffers of the fused launches (they used to be rebuilt as zeros with a large step count)."""
    import argparse
    import copy
    from OATrans import model as module_arch
    from OATrans.optim import AdamW
    from OATrans.parallel import HipDataParallel
    from OATrans.trainer.step import hot_step
    sa = argparse.Namespace(world_size=1, rank=0, local_rank=0)
    data = _batch()
    loss_fn = module_arch.NormSoftmaxLoss()

    def fresh():
        m = _small_frozen(seed=1)
        return m, HipDataParallel(m), AdamW([p for p in m.parameters() if p.requires_grad], lr=2e-4)

    m1, dp1, opt1 = fresh()
    for _ in range(2):
        hot_step(dp1, loss_fn, opt1, data, sa)
    torch.cuda.synchronize()
    sd_model = copy.deepcopy(m1.state_dict())
    sd_opt = copy.deepcopy(opt1.state_dict())
    hot_step(dp1, loss_fn, opt1, data, sa)                   # third step, uninterrupted
    m2, dp2, opt2 = fresh()
    m2.load_state_dict(sd_model)
    opt2.load_state_dict(sd_opt)
    hot_step(dp2, loss_fn, opt2, data, sa)                   # third step after a resume
    torch.cuda.synchronize()
    assert all(st["step"] == 3 for st in opt2.state.values())
    for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        # same weights, same moments, same data: only the fp32-atomics noise of the CLS-row gradients differs
        d = (p1 - p2).abs().max().item()
        assert d <= 4e-4, (n, d)
        assert (p1 - p2).abs().mean().item() < 1e-5, n
    # a zero-moment restart at step 3 would move every weight by ~lr * sign(g): far outside the bound above
    exp_avg = next(iter(opt2.state.values()))["exp_avg"]
    assert exp_avg.abs().max().item() > 0


def test_optimizer_load_state_dict_in_capture_mode():
    """AdamW.load_state_dict after enable_capture(): the loaded moments go INTO the existing flat buffers (captured
    graphs hold their addresses) and the device-side step / lr scalars are re-synchronised, so the next step carries
    the loaded step count's bias corrections - it used to raise 'buffers moved after enable_capture'."""
    import copy
    from OATrans.optim import AdamW
    torch.manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(257, device="cuda")), torch.nn.Parameter(torch.randn(33, 5, device="cuda"))]
    gs = [[torch.randn_like(p) for p in ps] for _ in range(4)]

    def run(opt, k):
        for p, g in zip(ps_of[opt], gs[k]):
            p.grad = g.clone() if p.grad is None else p.grad.copy_(g)
        opt.step()

    ps_of = {}
    a_params = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    a = AdamW(a_params, lr=1e-2)
    ps_of[a] = a_params
    for k in range(3):
        run(a, k)
    sd = copy.deepcopy(a.state_dict())
    saved = [p.detach().clone() for p in a_params]
    run(a, 3)                                            # fourth step, uninterrupted, host scalars
    b_params = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    b = AdamW(b_params, lr=1e-2)
    ps_of[b] = b_params
    run(b, 0)                                            # builds the runs
    b.enable_capture()
    m_ptr = b._runs[0]['m'].data_ptr()
    with torch.no_grad():
        for p, s in zip(b_params, saved):
            p.copy_(s)
    b.load_state_dict(sd)
    assert b._runs[0]['m'].data_ptr() == m_ptr           # same flat buffers: captured graphs stay valid
    assert int(b._dev[0]['step'].item()) == 3
    run(b, 3)                                            # fourth step through the device-scalar path
    torch.cuda.synchronize()
    assert all(int(st["step"]) == 4 for st in b.state.values())
    for pa, pb in zip(a_params, b_params):
        assert torch.allclose(pa, pb, atol=2e-6, rtol=1e-5), (pa - pb).abs().max()


@pytest.mark.parametrize("B", [32, 64])
def test_headline_batch_sim_matrix_vs_oracle_rows(B):
    """The benchmarked shapes themselves - bs 32 (configs 2 / 3) and bs 64 per GPU (config 4), 8 frames, ViT-B/16 +
    DistilBERT-base - against the CPU oracle: the HIP path embeds all pairs in one forward; the oracle (minutes per full batch on a CPU) embeds all 32 captions and
    a SUBSET of 3 videos (samples are independent: no batch statistics anywhere in the model), and the corresponding
    COLUMNS of the 32 x 32 sim matrix must agree within the stated 1e-3."""
    from OATrans import model as module_arch
    from oracle import oatrans_oracle as orc
    T, L = 8, 32
    sd = si.frozen_state_dict(SEED, dict(num_frames=T), {})
    m = module_arch.FrozenInTime(
        video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=T, pretrained=True, time_init="rand"),
        object_params=dict(model="", input_objects=False),
        text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text"),
        projection="minimal", load_checkpoint="")
    m.text_model.eval()          # parity runs in eval mode (the goldens' DistilBERT has dropout off)
    r = m.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and not r.missing_keys, r
    m = m.cuda()
    video = si.seeded_tensor(SEED, f"bs{B}.video", (B, T, 3, 224, 224))
    ids = si.seeded_ints(SEED, f"bs{B}.ids", (B, L), 1000, 30000)
    ids[:, 0] = 101
    mask = torch.ones(B, L, dtype=torch.int64)
    mask[5, 20:] = 0
    mask[17, 9:] = 0
    m.begin_step()
    with torch.no_grad():
        t, v = m({"video": video.cuda(), "text": {"input_ids": ids.cuda(), "attention_mask": mask.cuda()}})
        sim = module_arch.sim_matrix(t, v).cpu()
    cols = [0, 13, B - 1]
    torch.set_num_threads(min(16, torch.get_num_threads()))
    import torch.nn.functional as F
    with torch.no_grad():
        ot = orc.distilbert(ids, mask, sd)[:, 0]
        ot = F.linear(F.relu(ot), sd["txt_proj.1.weight"], sd["txt_proj.1.bias"])
        ocls, _ = orc.video_encoder(video[cols], sd)
        ov = F.linear(ocls, sd["vid_proj.0.weight"], sd["vid_proj.0.bias"])
        osim = orc.sim_matrix(ot, ov)
    err = (sim[:, cols] - osim).abs().max().item()
    print(f"bs{B} 8f sim-matrix max abs err on 3 video columns:", err, "text rel", rel(t, ot), "video rel", rel(v[cols], ov))
    assert err <= 1e-3, err
    assert rel(t, ot) < 1e-2 and rel(v[cols], ov) < 1e-2


def test_graphed_step_matches_eager_steps():
    """trainer/graph_step.py: the training step captured into one hipGraph (ctypes launches, side streams, autograd
    backward, AdamW with device-resident step state) and replayed must walk the same trajectory as eager stepping:
    same losses step by step (up to the fp32-atomics noise of the CLS-row gradients), same parameters, a changing
    learning rate honoured, a second input signature captured separately."""
    import argparse
    import copy
    from OATrans import model as module_arch
    from OATrans.optim import AdamW
    from OATrans.parallel import HipDataParallel
    from OATrans.trainer.graph_step import GraphedStep
    from OATrans.trainer.step import hot_step
    sa = argparse.Namespace(world_size=1, rank=0, local_rank=0)
    loss_fn = module_arch.NormSoftmaxLoss()
    base = _small_frozen(seed=2, depth=2)
    batches = [_batch(seed=10 + i) for i in range(7)]
    other = _batch(B=2, T=2, L=7, seed=99)
    results = []
    for graphed in (False, True):
        m = copy.deepcopy(base)
        for sub in (m.video_model, m.text_model):
            sub.flatten_parameters()
        dp = HipDataParallel(m)
        opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4)
        stepper = GraphedStep(hot_step, dp, loss_fn, opt, sa, warmup=2) if graphed else \
            (lambda d: hot_step(dp, loss_fn, opt, d, sa))
        losses = []
        for i, b in enumerate(batches):
            if i == 5:
                for g in opt.param_groups:
                    g["lr"] = 3e-5                       # the trainer's per-epoch schedule (trainer_dist.py:117-122)
            losses.append(stepper(b).item())
        for _ in range(4):                               # a second input signature: 2 eager steps, then its own graph
            losses.append(stepper(other).item())
        torch.cuda.synchronize()
        if graphed:
            assert stepper.replays == 5 + 2 and len(stepper._graphs) == 2
        assert all(st["step"] == 11 for st in opt.state.values())
        results.append((losses, {n: p.detach().clone() for n, p in m.named_parameters()}))
    (l0, p0), (l1, p1) = results
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 5e-3, (l0, l1)
    for n in p0:
        d = (p0[n] - p1[n]).abs()
        assert d.max().item() <= 11 * 2e-4 + 1e-6, (n, d.max().item())
        assert d.mean().item() < 4e-5, (n, d.mean().item())      # atomics-order noise through 11 Adam steps (2.5e-5 seen)
