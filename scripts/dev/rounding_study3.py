"""Dev (CPU, no GPU): what does a bf16 RESIDUAL stream (and a bf16 residual-GRADIENT stream) cost in parity?

Emulates, inside the CPU oracle on the inputs of tests/golden/full_T{1,4,8}.pt, the engine's numerics
  cur      : bf16 GEMM operands / bf16 GEMM outputs, fp32 residual stream, fp32 gradient stream, exact text tower,
             fp32 CLS lane for the forward (what engine/video.py runs today)
  res16/k  : + the residual stream STORED as bf16, rounded k times per block (k = 1: block output only - today's
             schedule with a bf16 buffer; k = 2: y = x + space and the block output - residual adds in the GEMM epilogues;
             k = 3: also x + time)
  cls32    : ... with the CLS rows of the stream kept fp32 (patch rows only are rounded)
  g16      : + the residual-gradient stream rounded to bf16 once per block (dL/d(block input))
and reports (a) the sim-matrix error against the REFERENCE's golden sim matrix, (b) every video parameter gradient against
the fp32 oracle's autograd: worst / global relative L2 and cosine.

    python scripts/dev/rounding_study3.py [T ...]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from OATrans.utils import seeded_init as si
from oracle import oatrans_oracle as orc

SEED = 20240917
torch.set_num_threads(8)


class _RoundFwd(torch.autograd.Function):
    """value rounded to bf16, gradient passed through"""
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """value passed through, gradient rounded to bf16"""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _rows(fn, x, cls32):
    """apply fn to the patch rows only (cls32) or to all rows of [B, S, D]"""
    if not cls32:
        return fn(x)
    return torch.cat([x[:, :1], fn(x[:, 1:])], dim=1)


def make_block(k_round, cls32, g16):
    rf = lambda x: _rows(_RoundFwd.apply, x, cls32) if k_round else x

    def block(x, p, i, T, N, H, pre="video_model."):
        b = f"{pre}blocks.{i}."
        if g16:
            x = _rows(_RoundBwd.apply, x, cls32)          # dL/d(block input) leaves the block as bf16
        t_out = orc.divided_attention(orc._ln(x, p, b + "norm3", 1e-6), p, b + "timeattn", "time", T, N, H)
        xt = x + t_out
        if k_round >= 3:
            xt = rf(xt)
        s_out = orc.divided_attention(orc._ln(xt, p, b + "norm1", 1e-6), p, b + "attn", "space", T, N, H)
        y = x + s_out
        if k_round >= 2:
            y = rf(y)
        h = F.gelu(orc._lin(orc._ln(y, p, b + "norm2", 1e-6), p, b + "mlp.fc1"))
        out = y + orc._lin(h, p, b + "mlp.fc2")
        return rf(out)
    return block


def run(T, variant, grads=False):
    g = torch.load(os.path.join(ROOT, "tests/golden", f"full_T{T}.pt"), weights_only=False)
    B, L = g["B"], g["L"]
    sd = si.frozen_state_dict(SEED, dict(num_frames=T), {})
    video = si.seeded_tensor(SEED, f"full.video.{T}", (B, T, 3, 224, 224))
    ids = si.seeded_ints(SEED, f"full.ids.{T}", (B, L), 1000, 30000); ids[:, 0] = 101
    exact = variant == "fp32"
    k_round = int(variant.split("/")[1][0]) if variant.startswith("res16") else 0
    cls32 = "cls32" in variant
    g16 = "g16" in variant
    lane = not grads and not exact        # forward parity is judged on the CLS lane's output, gradients on the main path
    r = lambda x: x.bfloat16().float()    # rounds the value AND (autograd of the casts) the gradient
    orig_lin, orig_blk = orc._lin, orc.space_time_block

    def lin(x, p, name):
        W, b = p[name + ".weight"], p[name + ".bias"]
        if exact or name.startswith("text_model") or name.startswith("txt_proj"):
            return F.linear(x, W, b)
        y = r(F.linear(r(x), r(W), b))
        if lane and x.dim() == 3 and x.shape[1] > 1:
            y = torch.cat([F.linear(x[:, :1], W, b), y[:, 1:]], dim=1)
        return y

    if grads:
        sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    orc._lin = lin
    orc.space_time_block = make_block(k_round, cls32 or lane, g16)
    try:
        with torch.set_grad_enabled(grads):
            loss, sim, t, v = orc.train_step_loss(sd, video, ids, g["mask"])
            if grads:
                loss.backward()
    finally:
        orc._lin, orc.space_time_block = orig_lin, orig_blk
    err = (sim.detach() - g["sim"]).abs().max().item()
    gr = {k: v.grad for k, v in sd.items() if grads and k.startswith("video_model.") and v.grad is not None}
    return err, gr


def compare(ref, got):
    worst_rel, worst_cos, num, den, dot, na, nb = 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0
    wr = wc = ""
    for k, a in ref.items():
        b = got[k]
        if a.norm() == 0:
            continue
        rel = ((a - b).norm() / a.norm()).item()
        cos = (F.cosine_similarity(a.flatten(), b.flatten(), dim=0)).item()
        if rel > worst_rel:
            worst_rel, wr = rel, k
        if cos < worst_cos:
            worst_cos, wc = cos, k
        num += (a - b).pow(2).sum().item(); den += a.pow(2).sum().item()
        dot += (a * b).sum().item(); na += a.pow(2).sum().item(); nb += b.pow(2).sum().item()
    return worst_rel, wr, worst_cos, wc, (num / den) ** 0.5, dot / (na * nb) ** 0.5


if __name__ == "__main__":
    Ts = [int(a) for a in sys.argv[1:]] or [1, 4, 8]
    fwd_variants = ["cur", "res16/1", "res16/2", "res16/3"]
    print("== forward: sim-matrix max-abs error vs the reference golden (CLS lane fp32, bound 1e-3)")
    for T in Ts:
        for name in fwd_variants:
            e, _ = run(T, name)
            print(f"T={T} {name:22s} sim err {e:.2e}", flush=True)
    print("== gradients at T=%d: every video parameter gradient vs the fp32 oracle's autograd (main path, no lane)" % Ts[-1])
    T = Ts[-1]
    _, ref = run(T, "fp32", grads=True)
    for name in ["cur", "res16/1", "res16/1 g16", "res16/1 cls32 g16", "res16/2 g16", "res16/3 g16"]:
        e, gr = run(T, name, grads=True)
        wr, kr, wc, kc, grel, gcos = compare(ref, gr)
        print(f"T={T} {name:22s} main-path sim err {e:.2e} | worst rel-L2 {wr:.3e} ({kr[12:]}) worst cos {wc:.6f} ({kc[12:]}) | "
              f"all params rel-L2 {grel:.3e} cos {gcos:.6f}", flush=True)
