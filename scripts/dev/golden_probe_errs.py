"""dev: the sampled-gradient errors of tests/test_model_gpu.py::test_frozen_in_time_vitb_vs_reference_golden, printed (top 6 per run).
usage: [OAT_LIB=...] python scripts/dev/golden_probe_errs.py [frames ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans import model as module_arch
from OATrans.utils import seeded_init as si
SEED = 20240917
for frames in [int(a) for a in sys.argv[1:]] or [4]:
    g = torch.load(os.path.join(ROOT, "tests", "golden", f"full_T{frames}.pt"), map_location="cpu", weights_only=False)
    T, B, L = g["T"], g["B"], g["L"]
    for prune in (False, True):
        m = module_arch.FrozenInTime(
            video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=T, pretrained=True, time_init="rand"),
            object_params=dict(model="", input_objects=False),
            text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text"), projection="minimal", load_checkpoint="")
        m.text_model.eval()
        m.load_state_dict(si.frozen_state_dict(SEED, dict(num_frames=T), {}), strict=False)
        m = m.cuda()
        m.video_model._engine.prune_top = prune
        video = si.seeded_tensor(SEED, f"full.video.{T}", (B, T, 3, 224, 224)).cuda()
        ids = si.seeded_ints(SEED, f"full.ids.{T}", (B, L), 1000, 30000); ids[:, 0] = 101
        m.begin_step()
        t, v = m({"video": video, "text": {"input_ids": ids.cuda(), "attention_mask": g["mask"].cuda()}})
        loss = module_arch.NormSoftmaxLoss()(module_arch.sim_matrix(t, v)); loss.backward(); torch.cuda.synchronize()
        params = dict(m.named_parameters()); rows = []
        for k, pr in g["grad_probe"].items():
            gr = params[k].grad
            if pr["norm"] < 1e-6: continue
            nerr = abs(gr.norm().item() - pr["norm"].item()) / pr["norm"].item()
            scale = pr["norm"].item() / gr.numel() ** 0.5
            perr = ((gr.flatten()[pr["idx"].cuda()].cpu() - pr["val"]).abs() / scale).max().item()
            rows.append((perr, nerr, k))
        rows.sort(reverse=True)
        print(f"frames {frames} prune {prune} lib {os.environ.get('OAT_LIB', 'default')}:", [(k, round(p, 3), round(n, 4)) for p, n, k in rows[:6]], flush=True)
