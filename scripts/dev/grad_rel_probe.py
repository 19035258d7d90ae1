"""dev: every parameter gradient of the frozen model at B = 2 against the CPU oracle's autograd, worst tensors printed.
usage: [OAT_LIB=...] python scripts/dev/grad_rel_probe.py [frames ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans import model as module_arch
from OATrans.utils import seeded_init as si
from oracle import oatrans_oracle as orc
SEED = 20240917
torch.set_num_threads(16)
for frames in [int(a) for a in sys.argv[1:]] or [4]:
    T, B, L = frames, 2, 12
    sd = si.frozen_state_dict(SEED, dict(num_frames=T), {})
    video = si.seeded_tensor(SEED, f"full.video.{T}", (B, T, 3, 224, 224))
    ids = si.seeded_ints(SEED, f"full.ids.{T}", (B, L), 1000, 30000); ids[:, 0] = 101
    mask = torch.ones(B, L, dtype=torch.int64); mask[1, L - 3:] = 0
    p = {k: (w.clone().requires_grad_(True) if w.is_floating_point() else w) for k, w in sd.items()}
    oloss, _, _, _ = orc.train_step_loss(p, video, ids, mask); oloss.backward()
    for prune in (False, True):
        m = module_arch.FrozenInTime(
            video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=T, pretrained=True, time_init="rand"),
            object_params=dict(model="", input_objects=False),
            text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text"), projection="minimal", load_checkpoint="")
        m.text_model.eval(); m.load_state_dict(sd, strict=False); m = m.cuda()
        m.video_model._engine.prune_top = prune
        m.begin_step()
        t, v = m({"video": video.cuda(), "text": {"input_ids": ids.cuda(), "attention_mask": mask.cuda()}})
        loss = module_arch.NormSoftmaxLoss()(module_arch.sim_matrix(t, v)); loss.backward(); torch.cuda.synchronize()
        rows = []; num = den = 0.0
        for k, prm in m.named_parameters():
            ref = p[k].grad
            if ref is None or ref.norm() < 1e-6: continue
            mine = prm.grad.float().cpu()
            rows.append((((mine - ref).norm() / ref.norm()).item(), k))
            if k.startswith("video_model."):
                num += (mine - ref).pow(2).sum().item(); den += ref.pow(2).sum().item()
        rows.sort(reverse=True)
        pe = [r for r in rows if r[1] == "video_model.pos_embed"][0][0]
        print(f"frames {frames} prune {prune} lib {os.path.basename(os.environ.get('OAT_LIB', 'default'))}: loss {loss.item():.5f} vs {oloss.item():.5f}; video tower rel-L2 {(num / den) ** 0.5:.4e}; "
              f"pos_embed {pe:.4e}; worst {[(k, round(e, 4)) for e, k in rows[:4]]}", flush=True)
