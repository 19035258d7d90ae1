"""Dev (CPU): emulate an fp32 'CLS lane' (CLS-row linears exact, patch rows bf16) and an exact text tower."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from OATrans.utils import seeded_init as si
from oracle import oatrans_oracle as orc
SEED = 20240917
torch.set_num_threads(8)

def run(T, text_exact, cls_lane, q_only=False):
    g = torch.load(os.path.join(ROOT, "tests/golden", f"full_T{T}.pt"), weights_only=False)
    B, L = g["B"], g["L"]
    sd = si.frozen_state_dict(SEED, dict(num_frames=T), {})
    video = si.seeded_tensor(SEED, f"full.video.{T}", (B, T, 3, 224, 224))
    ids = si.seeded_ints(SEED, f"full.ids.{T}", (B, L), 1000, 30000); ids[:, 0] = 101
    orig = orc._lin
    r = lambda x: x.bfloat16().float()
    def lin(x, p, name):
        W, b = p[name + ".weight"], p[name + ".bias"]
        if name.startswith("text_model") or name.startswith("txt_proj"):
            return F.linear(x, W, b) if text_exact else r(F.linear(r(x), r(W), b))
        y = r(F.linear(r(x), r(W), b))
        if cls_lane and x.dim() == 3 and x.shape[1] > 1:
            ex = F.linear(x[:, :1], W, b)
            if q_only and name.endswith(".qkv"):
                D = ex.shape[-1] // 3
                y[:, :1, :D] = ex[..., :D]
            else:
                y[:, :1] = ex
        return y
    orc._lin = lin
    try:
        with torch.no_grad():
            t, v = orc.frozen_forward(sd, video, ids, g["mask"])
            sim = orc.sim_matrix(t, v)
    finally:
        orc._lin = orig
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    return (sim - g["sim"]).abs().max().item(), rel(t, g["text"]), rel(v, g["video"])

for T in (1, 4, 8):
    for name, kw in [("bf16", dict(text_exact=False, cls_lane=False)),
                     ("bf16 + CLS lane", dict(text_exact=False, cls_lane=True)),
                     ("bf16 + CLS lane (q only) + exact text", dict(text_exact=True, cls_lane=True, q_only=True)),
                     ("bf16 + CLS lane + exact text", dict(text_exact=True, cls_lane=True))]:
        e, rt, rv = run(T, **kw)
        print(f"T={T} {name:40s} sim err {e:.2e}  text rel {rt:.2e}  video rel {rv:.2e}", flush=True)
