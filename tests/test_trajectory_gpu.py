"""Multi-step training-trajectory parity: K optimiser steps of the HIP hot path (`trainer/step.py: hot_step` - the function
bench.py times) against K steps of the CPU oracle (`oracle.train_step_loss` + a plain-torch transformers-4.6 AdamW),
same seeded weights, a fresh seeded batch every step.  What the single-step tests cannot show: that the error of the bf16
pipeline (bf16 MFMA operands, bf16 residual / gradient stream, 8-bit GELU derivative) is unbiased over steps, i.e. that the
HIP run TRACKS the fp32 reference's trajectory instead of drifting away from it.

Reference loop: /root/reference/OATrans/trainer/trainer_dist.py:156-168 (zero_grad, forward, all-gather, loss, backward, step);
optimiser: transformers.AdamW as the reference configs resolve it (train_dist_multi.py:66, lr 2e-4, betas (0.9, 0.999),
eps 1e-6, no weight decay, bias correction on).  DistilBERT in .eval() on both sides (torch's dropout stream cannot be
reproduced; tests/test_text_dropout_gpu.py pins training-mode dropout with shared Philox masks).

Stated bounds (measured values are printed and quoted beside the bounds below; DESIGN section 2 has them too):
  * per-step loss: |loss_hip - loss_oracle| <= loss_tol * max(1, |loss_oracle|) at every step (1.5e-2 on the small geometry,
    8e-3 at ViT-B/16: about three times the measured deviations);
  * cumulative update of all parameters after K steps, delta = theta_K - theta_0 (Adam's first steps are sign descent, so
    elements whose gradient is smaller than its bf16 error take a step of the same size in either direction - the bound is
    on direction and size of the whole update, not on elements): cosine(delta_hip, delta_oracle) >= 0.998 (small) / 0.99 (ViT-B/16)
    and | ||delta_hip|| / ||delta_oracle|| - 1 | <= 0.01;
  * the default storage choices are not systematically worse than the run with both switched off (fp32 residual stream,
    bf16 derivative): |mean signed loss deviation| of the default <= that of the off-run + 3e-3.
Parameters whose gradient is analytically zero (the key biases: softmax is invariant to them) are driven by rounding noise
through Adam's normalisation on both sides and are left out of the update comparison."""
import argparse

import pytest
import torch

from OATrans.utils import seeded_init as si

pytestmark = pytest.mark.gpu
SEED = 20240917
SMALL_VIDEO = dict(embed_dim=128, depth=2, mlp_ratio=4, num_frames=3, patches_per_frame=9, patch=16)
SMALL_TEXT = dict(dim=128, n_layers=2, hidden_dim=512, vocab=1000, max_pos=64)
B1, B2, EPS = 0.9, 0.999, 1e-6


def _batches(tag, steps, B, T, R, L, vocab_lo, vocab_hi):
    out = []
    for k in range(steps):
        video = si.seeded_tensor(SEED, f"traj.{tag}.video.{k}", (B, T, 3, R, R))
        ids = si.seeded_ints(SEED, f"traj.{tag}.ids.{k}", (B, L), vocab_lo, vocab_hi)
        mask = torch.ones(B, L, dtype=torch.int64)
        mask[B - 1, L - 3:] = 0                                  # one ragged caption
        out.append((video, ids, mask))
    return out


def _oracle_run(sd, batches, lr, heads, text_heads):
    """K steps of oracle.train_step_loss + transformers-4.6 AdamW in plain torch (fp32, CPU).  -> (losses, final params)"""
    from oracle import oatrans_oracle as orc
    torch.set_num_threads(min(16, torch.get_num_threads()))      # torch's CPU kernels collapse when every SMT thread of a GPU box is used
    p = {k: (w.clone().requires_grad_(True) if w.is_floating_point() else w) for k, w in sd.items()}
    mom = {k: (torch.zeros_like(w), torch.zeros_like(w)) for k, w in p.items() if w.is_floating_point()}
    losses = []
    for t, (video, ids, mask) in enumerate(batches, start=1):
        for w in p.values():
            if w.is_floating_point():
                w.grad = None
        loss, _, _, _ = orc.train_step_loss(p, video, ids, mask, num_heads=heads, text_heads=text_heads)
        loss.backward()
        losses.append(loss.item())
        with torch.no_grad():
            for k, w in p.items():
                if not w.is_floating_point() or w.grad is None:
                    continue
                m, v = mom[k]
                m.mul_(B1).add_(w.grad, alpha=1 - B1)
                v.mul_(B2).addcmul_(w.grad, w.grad, value=1 - B2)
                w.addcdiv_(m, v.sqrt().add_(EPS), value=-lr * (1 - B2 ** t) ** 0.5 / (1 - B1 ** t))      # eps outside the bias correction
    return losses, {k: w.detach() for k, w in p.items() if w.is_floating_point()}


def _hip_run(make_model, sd, batches, lr, res16=None, h_u8=None):
    from OATrans import model as module_arch
    from OATrans.optim import AdamW
    from OATrans.parallel import HipDataParallel
    from OATrans.trainer.step import hot_step
    m = make_model()
    m.text_model.eval()
    r = m.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and not r.missing_keys, r
    m = m.cuda()
    m.set_device(torch.device("cuda"))
    for sub in (m.video_model, m.text_model):
        sub.flatten_parameters()
    eng = m.video_model._engine
    if res16 is not None:
        eng.res16 = res16
    if h_u8 is not None:
        eng.h_u8 = h_u8
    dp = HipDataParallel(m)
    opt = AdamW([q for q in m.parameters() if q.requires_grad], lr=lr)
    sa = argparse.Namespace(world_size=1, rank=0, local_rank=0)
    loss_fn = module_arch.NormSoftmaxLoss()
    losses = []
    for video, ids, mask in batches:
        data = {"video": video.cuda(), "text": {"input_ids": ids.cuda(), "attention_mask": mask.cuda()}}
        losses.append(hot_step(dp, loss_fn, opt, data, sa).item())
    torch.cuda.synchronize()
    if res16 is not None:
        assert all(pl.res16 == res16 for pl in eng.plans.values())
    return losses, {k: w.detach().float().cpu() for k, w in m.named_parameters()}


def _update_stats(sd, hip_p, orc_p):
    """cosine and norm ratio of the cumulative updates over every parameter with a non-degenerate gradient"""
    dot = nh = no = 0.0
    for k, w0 in sd.items():
        if not w0.is_floating_point() or k not in hip_p or k not in orc_p or k.endswith("k_lin.bias"):
            continue
        dh, do = (hip_p[k] - w0).double().flatten(), (orc_p[k] - w0).double().flatten()
        if k.endswith("qkv.bias"):                               # [q | k | v]: the key third has an analytically zero gradient
            n = dh.numel() // 3
            dh, do = torch.cat([dh[:n], dh[2 * n:]]), torch.cat([do[:n], do[2 * n:]])
        dot += (dh * do).sum().item()
        nh += (dh * dh).sum().item()
        no += (do * do).sum().item()
    return dot / (nh * no) ** 0.5, (nh / no) ** 0.5


def _check(tag, lo, lh, sd, hp, op, loss_tol, cos_min, norm_tol):
    dev = [a - b for a, b in zip(lh, lo)]
    worst = max(abs(d) / max(1.0, abs(b)) for d, b in zip(dev, lo))
    cos, ratio = _update_stats(sd, hp, op)
    print(f"[{tag}] oracle loss {lo[0]:.4f} -> {lo[-1]:.4f}, hip {lh[0]:.4f} -> {lh[-1]:.4f}; worst per-step loss deviation {worst:.2e}, "
          f"mean signed {sum(dev) / len(dev):+.2e}; update cosine {cos:.4f}, norm ratio {ratio:.4f}")
    assert worst <= loss_tol, (tag, worst, dev)
    assert cos >= cos_min and abs(ratio - 1) <= norm_tol, (tag, cos, ratio)
    return sum(dev) / len(dev)


def test_twenty_adamw_steps_small_chain_geometry_track_the_oracle():
    """20 steps at lr 2e-4 (the reference configs' value) on the small_chain geometry (3 frames of 48^2, 128-d, 2 + 2 blocks):
    default engine options and the run with the bf16 residual stream switched off."""
    from OATrans import model as module_arch
    steps, lr = 20, 2e-4
    sd = si.frozen_state_dict(SEED, SMALL_VIDEO, SMALL_TEXT, proj_dim=64)
    batches = _batches("small", steps, 4, 3, 48, 7, 1, 1000)

    def make():
        return module_arch.FrozenInTime(
            video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=3, pretrained=True, time_init="rand",
                              arch_kwargs=dict(img_size=48, patch_size=16, embed_dim=128, depth=2, num_heads=2)),
            object_params=dict(model="", input_objects=False),
            text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text",
                             config=dict(vocab_size=1000, max_position_embeddings=64, n_layers=2, n_heads=2, dim=128, hidden_dim=512)),
            projection_dim=64, projection="minimal", load_checkpoint="")

    lo, op = _oracle_run(sd, batches, lr, heads=2, text_heads=2)
    lh, hp = _hip_run(make, sd, batches, lr)
    lf, fp = _hip_run(make, sd, batches, lr, res16=False, h_u8=False)
    # measured (round 6): worst per-step loss deviation 5.1e-3 / 5.6e-3, mean signed -1.0e-3 / -3.6e-4, update cosine 0.9996 / 0.9996,
    # norm ratio 0.9998 / 0.9998 (default / fp32 stream)
    b_def = _check("small, default", lo, lh, sd, hp, op, loss_tol=1.5e-2, cos_min=0.998, norm_tol=0.01)
    b_off = _check("small, fp32 stream", lo, lf, sd, fp, op, loss_tol=1.5e-2, cos_min=0.998, norm_tol=0.01)
    assert abs(b_def) <= abs(b_off) + 3e-3, (b_def, b_off)


def test_five_adamw_steps_headline_geometry_track_the_oracle():
    """5 steps at lr 2e-4 at the headline geometry (ViT-B/16, 8 frames of 224^2, DistilBERT-base) at B = 2: the default options
    (bf16 streams, 8-bit GELU derivative, pruned top block) and the run with both storage departures off."""
    from OATrans import model as module_arch
    steps, lr, T = 5, 2e-4, 8
    sd = si.frozen_state_dict(SEED, dict(num_frames=T), {})
    batches = _batches("full", steps, 2, T, 224, 12, 1000, 30000)

    def make():
        return module_arch.FrozenInTime(
            video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=T, pretrained=True, time_init="rand"),
            object_params=dict(model="", input_objects=False),
            text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text"),
            projection="minimal", load_checkpoint="")

    lo, op = _oracle_run(sd, batches, lr, heads=12, text_heads=12)
    lh, hp = _hip_run(make, sd, batches, lr)
    lf, fp = _hip_run(make, sd, batches, lr, res16=False, h_u8=False)
    # measured (round 6): worst per-step loss deviation 2.2e-3 / 1.6e-3, mean signed +1.3e-4 / +1.3e-3, update cosine 0.9971 / 0.9990,
    # norm ratio 1.0001 / 0.9999 (default / both departures off)
    b_def = _check("ViT-B/16 8f, default", lo, lh, sd, hp, op, loss_tol=8e-3, cos_min=0.99, norm_tol=0.01)
    b_off = _check("ViT-B/16 8f, both departures off", lo, lf, sd, fp, op, loss_tol=8e-3, cos_min=0.99, norm_tol=0.01)
    assert abs(b_def) <= abs(b_off) + 3e-3, (b_def, b_off)
