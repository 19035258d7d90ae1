"""fp8 forward path (BASELINE.json config 5) on a real MI355X.
Kernel level: the e4m3 conversion equals torch.float8_e4m3fn bit for bit, oat_gemm_nt_f8 equals an fp32 matmul of the
SAME quantised operands up to its bf16 output rounding, and stays within 4 % rel-L2 of the unquantised product.
Model level: the contract class with fp8 forward linears AND fp8 data-gradient GEMMs against the reference golden; stated (looser) tolerance:
embeddings rel-L2 <= 5e-2, sim matrix <= 6e-3 max-abs, loss <= 1e-2 (bf16 path: 1e-2 / 1e-3 / 2e-2).  Measured: video embedding
1.5-3.7e-2, sim 1.3-3.6e-3, loss 0.1-0.2 %.  The term that breaks the bf16 path's 1e-3: the text tower and the CLS rows of the
video tower stay fp32 (CLS lane), but the patch keys / values the CLS query attends come out of e4m3 GEMMs (2^-4 relative
rounding per operand element, per-tensor scale) - their error averages over ~1.5-7 k keys to ~2e-2 of the embedding."""
import os

import pytest
import torch

from OATrans.utils import seeded_init as si

pytestmark = pytest.mark.gpu
SEED = 20240917


def _quantise(x):
    from OATrans.ops import hip
    R, C = x.shape
    st = torch.zeros(3, device="cuda")
    hip.fp8_amax(x, R, C, st[0:1])
    amax = st[0].item()
    hip.fp8_update_scales(st[0:1], st[1:2], st[2:3], 1, 1.0)
    q = torch.empty(R, C, dtype=torch.uint8, device="cuda")
    hip.fp8_quant(x, q, R, C, st[1:2], st[0:1])
    return q, st, amax


@pytest.mark.parametrize("shape", [(512, 256, 256), (1000, 768, 768), (4113, 512, 3072)])
def test_fp8_quantise_and_gemm(shape):
    from OATrans.ops import hip
    m, n, k = shape
    torch.manual_seed(1)
    mp = (m + 255) // 256 * 256
    A = (torch.randn(mp, k, device="cuda") * 3).bfloat16()
    B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda")
    A8, sa, amax_a = _quantise(A)
    B8, sb, _ = _quantise(B)
    assert abs(amax_a - A.float().abs().max().item()) == 0.0 and abs(sa[0].item() - amax_a) == 0.0     # amax re-recorded by quant
    assert abs(sa[1].item() * amax_a - 448.0) < 1e-3 and abs(sa[1].item() * sa[2].item() - 1.0) < 1e-6
    want = (A.float() * sa[1]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(A8, want)                                  # OCP e4m3fn, saturating
    Aq = A8.view(torch.float8_e4m3fn).float() * sa[2]
    Bq = B8.view(torch.float8_e4m3fn).float() * sb[2]
    ref_q = Aq[:m] @ Bq.t() + bias
    ref = A[:m].float() @ B.float().t() + bias
    out = torch.full((mp, n), 7.0, device="cuda", dtype=torch.bfloat16)
    hip.gemm_nt_f8(A8, B8, m, n, k, hip.EPI_BF16, out, sa[2:3], sb[2:3], bias=bias)
    assert (out[m:] == 7.0).all()
    assert (out[:m].float() - ref_q).abs().max().item() <= 2 ** -8 * ref_q.abs().max().item() + 1e-3      # bf16 rounding of the output
    assert ((out[:m].float() - ref).norm() / ref.norm()).item() < 4e-2
    o1 = torch.empty(mp, n, device="cuda", dtype=torch.bfloat16)
    o2 = torch.empty_like(o1)
    hip.gemm_nt_f8(A8, B8, m, n, k, hip.EPI_GELU_GRAD, o1, sa[2:3], sb[2:3], out2=o2, bias=bias)
    assert (o2[:m].float() - torch.nn.functional.gelu(ref_q)).abs().max().item() <= 2 ** -7 * ref_q.abs().max().item() + 1e-2
    with pytest.raises(hip.OatError):
        hip.gemm_nt_f8(A8, B8, m, n - 8, k, hip.EPI_BF16, out, sa[2:3], sb[2:3])          # N % 256 != 0: refused, not approximated


@pytest.mark.parametrize("x32", [False, True])
def test_layernorm_r16_with_e4m3_copy(x32):
    """oat_layernorm_fwd_r16_f8 (fp8 forward on the bf16 residual stream): sum16 / y / mean / rstd bit-equal to the plain
    r16 kernel, y8 = the e4m3 bytes torch gives for y (fp32, before its bf16 rounding) * qscale, amax = max |y|."""
    from OATrans.ops import hip
    torch.manual_seed(3)
    M, D = 1003, 768
    x = torch.randn(M, D, device="cuda") * 2
    x = x if x32 else x.bfloat16()
    a, b = torch.randn(M, D, device="cuda").bfloat16(), torch.randn(M, D, device="cuda").bfloat16()
    outs = []
    for f8 in (False, True):
        s16 = torch.zeros(M, D, device="cuda", dtype=torch.bfloat16)
        y = torch.zeros_like(s16)
        mean, rstd = torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
        y8 = torch.zeros(M, D, device="cuda", dtype=torch.uint8)
        st = torch.tensor([0.0, 37.5], device="cuda")               # amax (recorded), qscale (given)
        kw = dict(y8=y8, qscale=st[1:2], amax=st[0:1]) if f8 else {}
        hip.layernorm_fwd_r16(x, M, D, 1e-6, add_a=a, add_b=b, sum16=s16, y=y, mean=mean, rstd=rstd, **kw)
        outs.append((s16, y, mean, rstd, y8, st))
    for u, v in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(u, v)
    s = x.float() + a.float() + b.float()
    yr = torch.nn.functional.layer_norm(s, (D,), eps=1e-6)
    y8, st = outs[1][4], outs[1][5]
    want = (yr * 37.5).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    diff = (y8 != want)
    # the kernel's fp32 y and torch's differ in the last bits, which flips an e4m3 rounding on a few elements per million
    assert diff.float().mean().item() < 1e-3
    assert (y8.view(torch.float8_e4m3fn).float() - want.view(torch.float8_e4m3fn).float()).abs().max().item() <= 32.0      # one e4m3 step at |q| <= 448
    assert abs(st[0].item() - yr.abs().max().item()) < 1e-4 * yr.abs().max().item()
    with pytest.raises(hip.OatError):
        hip.layernorm_fwd_r16(x, M, D, 1e-6, y=None, y8=y8, qscale=st[1:2], amax=st[0:1])


@pytest.mark.parametrize("frames", [4])
def test_frozen_in_time_fp8_forward_vs_reference_golden(golden_dir, frames):
    from OATrans import model as module_arch
    g = torch.load(os.path.join(golden_dir, f"full_T{frames}.pt"), map_location="cpu", weights_only=False)
    T, B, L = g["T"], g["B"], g["L"]
    m = module_arch.FrozenInTime(
        video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=T, pretrained=True, time_init="rand"),
        object_params=dict(model="", input_objects=False),
        text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text"),
        projection="minimal", load_checkpoint="")
    m.text_model.eval()
    m.load_state_dict(si.frozen_state_dict(SEED, dict(num_frames=T), {}), strict=False)
    m = m.cuda()
    m.video_model._engine.fp8 = True
    video = si.seeded_tensor(SEED, f"full.video.{T}", (B, T, 3, 224, 224)).cuda()
    ids = si.seeded_ints(SEED, f"full.ids.{T}", (B, L), 1000, 30000)
    ids[:, 0] = 101
    rel = lambda a, b: ((a.float().cpu() - b).norm() / b.norm()).item()
    for step in range(2):                       # step 0: current scaling (first use of every site), step 1: delayed scales
        m.begin_step()
        for prm in m.parameters():               # loose (autograd-accumulated) gradients: what optim.AdamW.zero_grad clears
            if not getattr(prm, "_oat_engine_grad", False):
                prm.grad = None
        t, v = m({"video": video, "text": {"input_ids": ids.cuda(), "attention_mask": g["mask"].cuda()}})
        sim = module_arch.sim_matrix(t, v)
        loss = module_arch.NormSoftmaxLoss()(sim)
        loss.backward()
        torch.cuda.synchronize()
        sim_err = (sim.detach().cpu() - g["sim"]).abs().max().item()
        print(f"fp8 step {step}: video rel {rel(v.detach(), g['video']):.4f} sim err {sim_err:.4f} loss {loss.item():.4f} vs {g['loss'].item():.4f}")
        assert rel(v.detach(), g["video"]) < 5e-2
        assert sim_err <= 6e-3
        assert abs(loss.item() - g["loss"].item()) < 1e-2
    f8 = m.video_model._engine._f8
    # per block: the 6 forward inputs have a delayed scale; every weight has one (the fp8 data-gradient mode of rounds 2-5 - measured
    # equal in time, gradient norms 10-12 % off - left the engine in round 6)
    assert len(f8["primed"]) == 6 * 12
    dq = f8["dq"].view(12, 18)
    assert bool((dq[:, :12] > 0).all())
    # the fp8 forward runs on the bf16 residual stream (the r16 LayerNorms emit the e4m3 operand)
    assert all(pl.res16 for pl in m.video_model._engine.plans.values())
    params = dict(m.named_parameters())
    errs = {k: abs(params[k].grad.norm().item() - pr["norm"].item()) / pr["norm"].item() for k, pr in g["grad_probe"].items()
            if pr["norm"] > 1e-6}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(f"worst gradient-norm errors {worst}")
    # bf16 backward behind an fp8 forward: <= 10 % (two stragglers allowed)
    assert sum(e > 0.1 for e in errs.values()) <= 2, worst


def test_config5_geometry_fp8_forward_vs_oracle():
    """BASELINE config 5's geometry at full width - ViT-B/16, 16 frames of 336^2 (441 patches per frame, 7057 tokens per
    clip), fp8 forward linears - against the fp32 CPU oracle (pinned by the reference goldens at 224^2 and, since round 6, at 336^2:
    tests/golden/video_336.pt; this 16-frame run is the oracle's own).  Two clips; stated tolerance as above: video embedding rel-L2 <= 5e-2, CLS cosine >= 0.9995."""
    from OATrans.model.video_transformer import SpaceTimeTransformer
    from oracle import oatrans_oracle as orc
    geo = dict(num_frames=16, patches_per_frame=441)
    sd = si.seeded_state_dict(si.video_param_shapes(**geo), SEED, "video_model.")
    m = SpaceTimeTransformer(img_size=336, patch_size=16, num_frames=16, time_init="rand")
    m.head = torch.nn.Identity()
    r = m.load_state_dict({k[len("video_model."):]: v for k, v in sd.items()}, strict=False)
    assert not r.unexpected_keys and not r.missing_keys, r
    m = m.cuda()
    m.need_patch_tokens = False
    m._engine.fp8 = True
    video = si.seeded_tensor(SEED, "in.video.c5", (2, 16, 3, 336, 336))
    with torch.no_grad():
        for _ in range(2):                        # second pass: delayed scales, producer-side quantisation
            cls = m(video.cuda())[0].float().cpu()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        ocls, _ = orc.video_encoder(video, sd)
    e = ((cls - ocls).norm() / ocls.norm()).item()
    cos = torch.nn.functional.cosine_similarity(cls, ocls, dim=1).min().item()
    print(f"config-5 geometry (16 x 336^2, fp8 forward): CLS rel-L2 {e:.4f}, min cosine {cos:.5f}")
    assert e < 5e-2 and cos > 0.9995


def test_config5_composed_global_local_16f_336_fp8_vs_oracle():
    """BASELINE config 5 COMPOSED, the form `bench.py --variant global_local --frames 16 --res 336 --dtype fp8` measures:
    oa_model_global_local.FrozenInTime (train_dist_multi_global_local.py) on one object frame + a 16-frame clip of
    336^2 per sample (441 patches per frame, native clip layout: two segments of one launch sequence), two DistilBERT
    passes, fp8 (e4m3) forward linears in the video tower.  B = 2 against the fp32 CPU oracle's run of the same graph:
    every embedding the trainer's three losses consume, and the loss itself (oa_model_global_local.py:149-208,
    trainer_global_local.py:187-211).  Then one backward: the fp8-forward step's parameter gradients against the bf16
    step's on the same weights and inputs (the oracle's backward at this size needs tens of GB).
    Stated tolerance for fp8: embeddings rel-L2 <= 5e-2, sim matrices <= 1e-2 max-abs (region x tags: mean over 10 region
    features of ONE object frame, 442 keys - the least averaging; measured 3.6e-3, text x video 1.3e-3), loss <= 1e-2; the text
    side and the CLS path of the video side stay fp32 (CLS lane), so what moves is the patch keys / values the CLS attends.
    The bf16 run of the same model is held to the bf16 bounds (1e-2 / 1e-3 / 2e-2; measured 2.4e-3 / 3e-4 / 0.05 %)."""
    from OATrans.data_loader.data_loader import MultiDistTextObjectVideoDataLoader
    from OATrans.model import NormSoftmaxLoss, sim_matrix
    from OATrans.model.oa_layers import mean_rows
    from OATrans.model.oa_model_global_local import FrozenInTime
    from oracle import oatrans_oracle as orc
    torch.set_num_threads(min(16, max(8, torch.get_num_threads())))
    B, T, R, L, O = 2, 16, 336, 32, 10
    p = si.frozen_state_dict(SEED, dict(num_frames=T, patches_per_frame=441), {})
    p.update(si.seeded_state_dict({"video_model.object_embed.weight": (768, 2054), "video_model.object_embed.bias": (768,),
                                   "text_local_proj.1.weight": (256, 768), "text_local_proj.1.bias": (256,),
                                   "vid_local_proj.0.weight": (256, 768), "vid_local_proj.0.bias": (256,)}, SEED))
    dl = MultiDistTextObjectVideoDataLoader("Synthetic", {"max_length": L}, {"input_res": R, "num_frames": 1}, "",
                                            batch_size=B, object_params={"input_objects": True, "num_objects": O})
    ex = dl.make_batch(55)
    video = si.seeded_tensor(SEED, "c5.gl.video", (B, T + 1, 3, R, R))
    ids = si.seeded_ints(SEED, "c5.ids", (B, L), 1000, 30000)
    ids[:, 0] = 101
    mask = torch.ones(B, L, dtype=torch.int64)
    mask[1, 20:] = 0
    pids, pmask, otm, pm = ex["pad_text"]["input_ids"], ex["pad_text"]["attention_mask"], ex["object_token_masks"], ex["patch_masks"]
    assert pm.shape == (B, O, 441)
    data = {"video": video.cuda(), "text": {"input_ids": ids.cuda(), "attention_mask": mask.cuda()},
            "pad_text": {"input_ids": pids.cuda(), "attention_mask": pmask.cuda()}, "patch_masks": pm.cuda(),
            "object_token_masks": otm.cuda(), "object_token_len": otm[:, -1].cuda()}
    Lf = NormSoftmaxLoss()
    runs = {}
    for fp8 in (True, False):
        m = FrozenInTime(dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=T, pretrained=True,
                              time_init="rand", two_outputs=False, object_clip="native", arch_kwargs=dict(img_size=R)),
                         dict(model="", input_objects=False),
                         dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text"))
        m.text_model.eval()
        r = m.load_state_dict(p, strict=False)
        assert not r.unexpected_keys and not r.missing_keys, r
        m = m.cuda()
        m.set_device(torch.device("cuda"))
        m.video_model._engine.fp8 = fp8
        for step in range(2 if fp8 else 1):       # fp8: the second step runs on delayed scales, producers quantise
            m.begin_step()
            for prm in m.parameters():
                if not getattr(prm, "_oat_engine_grad", False):
                    prm.grad = None
            t, pt, v, ov, extra = m(data)
            rf, tf = extra[4], extra[5]
            loss = Lf(sim_matrix(t, v)) + Lf(sim_matrix(pt, v)) + Lf(sim_matrix(mean_rows(rf), mean_rows(tf)))
            loss.backward()
        torch.cuda.synchronize()
        runs[fp8] = dict(out=[x.detach().float().cpu() for x in (t, pt, v, ov, rf, tf)], loss=loss.item(),
                         grads={k: prm.grad.detach().float().cpu().clone() for k, prm in m.named_parameters() if prm.grad is not None})
        if fp8:
            assert len(m.video_model._engine._f8["primed"]) == 6 * 12
        del m
        torch.cuda.empty_cache()
    with torch.no_grad():
        o = orc.gl_forward(p, video, (ids, mask), (pids, pmask), pm, otm, object_clip="native")
        oloss = orc.gl_loss(o[0], o[1], o[2], o[4], o[5]).item()
    names = ("text", "tagged text", "video", "object clip", "region_feat", "tags_feat")
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    for fp8 in (True, False):
        out = runs[fp8]["out"]
        rels = {n: rel(a, b) for n, a, b in zip(names, out, o)}
        sims = {"text x video": (orc.sim_matrix(out[0], out[2]) - orc.sim_matrix(o[0], o[2])).abs().max().item(),
                "tagged x video": (orc.sim_matrix(out[1], out[2]) - orc.sim_matrix(o[1], o[2])).abs().max().item(),
                "region x tags": (orc.sim_matrix(out[4].mean(1), out[5].mean(1)) - orc.sim_matrix(o[4].mean(1), o[5].mean(1))).abs().max().item()}
        print(f"config 5 composed (global_local 1+16 x 336^2, {'fp8' if fp8 else 'bf16'} forward): rel-L2 {rels}; sim errs {sims}; "
              f"loss {runs[fp8]['loss']:.4f} vs oracle {oloss:.4f}")
        tol_rel, tol_sim, tol_loss = (5e-2, 1e-2, 1e-2) if fp8 else (1e-2, 1e-3, 2e-2)
        assert all(e < tol_rel for e in rels.values()), rels
        assert all(e <= tol_sim for e in sims.values()), sims
        assert abs(runs[fp8]["loss"] - oloss) < tol_loss * max(1.0, abs(oloss))
    # backward of the fp8-forward step vs the bf16 step: same weights, same inputs
    g8, g16 = runs[True]["grads"], runs[False]["grads"]
    errs = {k: abs(g8[k].norm().item() - g16[k].norm().item()) / g16[k].norm().item() for k in g16 if g16[k].norm() > 1e-6 and "object_embed" not in k}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print("fp8-forward vs bf16 step, worst gradient-norm differences:", worst)
    assert all(torch.isfinite(x).all() for x in g8.values())
    assert sum(e > 0.1 for e in errs.values()) <= 2, worst
