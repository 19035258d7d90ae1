"""Dev (CPU): where does the sim-matrix error of a 16-bit-operand pipeline come from?  Emulates operand / storage rounding
inside the CPU oracle on the inputs of tests/golden/full_T{1,8}.pt and compares with the reference golden sim matrix."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from OATrans.utils import seeded_init as si
from oracle import oatrans_oracle as orc

SEED = 20240917
torch.set_num_threads(8)


def run(T, dt_video, dt_text, round_out=True, branch_fp32=False):
    g = torch.load(os.path.join(ROOT, "tests/golden", f"full_T{T}.pt"), weights_only=False)
    B, L = g["B"], g["L"]
    sd = si.frozen_state_dict(SEED, dict(num_frames=T), {})
    video = si.seeded_tensor(SEED, f"full.video.{T}", (B, T, 3, 224, 224))
    ids = si.seeded_ints(SEED, f"full.ids.{T}", (B, L), 1000, 30000)
    ids[:, 0] = 101
    orig = orc._lin

    def rnd(x, dt):
        return x if dt is None else x.to(dt).float()

    def lin(x, p, name):
        dt = dt_text if name.startswith("text_model") else dt_video
        y = F.linear(rnd(x, dt), rnd(p[name + ".weight"], dt), p[name + ".bias"])
        keep32 = branch_fp32 and (name.endswith(".proj") or name.endswith("fc2") or name.endswith("out_lin") or name.endswith("lin2"))
        return y if (not round_out or keep32) else rnd(y, dt)

    orc._lin = lin
    try:
        with torch.no_grad():
            t, v = orc.frozen_forward(sd, video, ids, g["mask"])
            sim = orc.sim_matrix(t, v)
    finally:
        orc._lin = orig
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    return (sim - g["sim"]).abs().max().item(), rel(t, g["text"]), rel(v, g["video"])


for T in (1, 8):
    for name, kw in [("fp32 oracle", dict(dt_video=None, dt_text=None)),
                     ("bf16 everywhere", dict(dt_video=torch.bfloat16, dt_text=torch.bfloat16)),
                     ("bf16, branch outputs fp32", dict(dt_video=torch.bfloat16, dt_text=torch.bfloat16, branch_fp32=True)),
                     ("bf16 video, exact text", dict(dt_video=torch.bfloat16, dt_text=None)),
                     ("bf16 video (fp32 branches), exact text", dict(dt_video=torch.bfloat16, dt_text=None, branch_fp32=True)),
                     ("fp16 everywhere", dict(dt_video=torch.float16, dt_text=torch.float16))]:
        e, rt, rv = run(T, **kw)
        print(f"T={T} {name:42s} sim err {e:.2e}  text rel {rt:.2e}  video rel {rv:.2e}", flush=True)
