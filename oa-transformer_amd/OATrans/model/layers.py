"""Small HIP-backed layers shared by the model assemblies: Linear projections
(txt_proj / vid_proj, /root/reference/OATrans/model/oa_model.py:66-78), sim_matrix and loss."""
import torch
from torch import nn

from ..ops import hip


import os

HEAD_FUSED = True     # projection heads / InfoNCE in their few-launch forms (False, set by tests: the earlier many-launch path)


def _round_up(x, m):
    return (x + m - 1) // m * m


class _LinearFn(torch.autograd.Function):
    """y = (relu(x) if pre_relu else x) @ W^T + b, fp32 in/out.  Forward: exact-f32 MFMA on the master weight
    (oat_linear_f32; these are the 768 -> 256 projections on B rows, where bf16 operand rounding would eat a third of the
    1e-3 sim-matrix bound); backward: the bf16 MFMA GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias, pre_relu):
        x = x.float().contiguous()
        M, K = x.shape
        N = weight.shape[0]
        ctx.small = False
        if HEAD_FUSED and M <= 64 and K % 16 == 0 and weight.is_contiguous() and (M * (N + 1) + 32 * M + 16 * N) * 4 <= 96 * 1024:
            # the projection heads (B rows): one launch forward, one launch backward (oat_linear_small_bwd), no bf16 copies
            y = torch.empty(M, N, dtype=torch.float32, device=x.device)
            hip.linear_f32(x, weight.detach(), M, N, K, bias=bias.detach() if bias is not None else None, out32=y,
                           act=(hip.LIN_RELU_IN if pre_relu else hip.LIN_NONE) | hip.LIN_EXACT)
            ctx.save_for_backward(x, weight)
            ctx.small, ctx.pre_relu, ctx.has_bias = True, pre_relu, bias is not None
            return y
        a16 = torch.zeros(_round_up(M, 64), K, dtype=torch.bfloat16, device=x.device)
        if pre_relu:
            hip.relu_bf16(x, a16, M, K)
        else:
            hip.cast_bf16(x, a16[:M])
        wT16 = torch.empty(K, N, dtype=torch.bfloat16, device=x.device)
        hip.cast_bf16(weight.detach().contiguous(), None, wT16)
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        if K % 16 == 0:
            hip.linear_f32(x, weight.detach(), M, N, K, bias=bias.detach() if bias is not None else None, out32=y,
                           act=(hip.LIN_RELU_IN if pre_relu else hip.LIN_NONE) | hip.LIN_EXACT)
        else:
            w16 = torch.empty(N, K, dtype=torch.bfloat16, device=x.device)
            hip.cast_bf16(weight.detach().contiguous(), w16, None)
            hip.gemm_nt(a16, w16, M, N, K, hip.EPI_F32, y, bias=bias.detach() if bias is not None else None)
        ctx.save_for_backward(x, a16, wT16)
        ctx.pre_relu, ctx.has_bias = pre_relu, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.small:
            x, weight = ctx.saved_tensors
            dy = dy.float().contiguous()
            M, N = dy.shape
            dx, dW, db = hip.linear_small_bwd(x, dy, weight.detach(), M, N, x.shape[1], relu_in=ctx.pre_relu,
                                              want_dx=ctx.needs_input_grad[0], want_db=ctx.has_bias)
            return dx, dW, db, None
        x, a16, wT16 = ctx.saved_tensors
        dy = dy.float().contiguous()
        M, N = dy.shape
        K = x.shape[1]
        dy16 = torch.zeros(_round_up(M, 64), N, dtype=torch.bfloat16, device=dy.device)
        hip.cast_bf16(dy, dy16[:M])
        dW = torch.empty(N, K, dtype=torch.float32, device=dy.device)
        hip.gemm_tn(dy16, a16, M, N, K, dW)
        db = None
        if ctx.has_bias:
            db = torch.empty(N, dtype=torch.float32, device=dy.device)
            hip.colsum(dy, M, N, db)
        dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
        hip.gemm_nt(dy16, wT16, M, K, N, hip.EPI_F32, dx)
        if ctx.pre_relu:
            hip.relu_bwd(x, dx, dx, M, K)
        return dx, dW, db, None


class HipLinear(nn.Linear):
    """nn.Linear whose forward/backward run on liboatrans_hip (same parameters / state_dict)."""

    pre_relu = False

    def forward(self, x):
        if not x.is_cuda:
            raise hip.OatError("HipLinear runs on MI355X only (no CPU path)")
        lead = x.shape[:-1]
        y = _LinearFn.apply(x.reshape(-1, x.shape[-1]), self.weight, self.bias, self.pre_relu)
        return y.reshape(*lead, y.shape[-1])


class ReLULinear(nn.Sequential):
    """txt_proj = Sequential(ReLU, Linear) (oa_model.py:68-70) with the state_dict key '1.*';
    the ReLU is fused into the GEMM operand cast."""

    def __init__(self, in_features, out_features, xavier=False):
        lin = HipLinear(in_features, out_features)
        lin.pre_relu = True
        if xavier:
            nn.init.xavier_uniform_(lin.weight)          # oa_model_region_mem.py:12-14
        super().__init__(nn.Identity(), lin)


class _SimFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, eps):
        a, b = a.float().contiguous(), b.float().contiguous()
        sim, ws = hip.sim_matrix_fwd(a, b, eps)
        ctx.ws, ctx.shape = ws, (a.shape[0], b.shape[0], a.shape[1])
        return sim

    @staticmethod
    def backward(ctx, g):
        n, m, d = ctx.shape
        da, db = hip.sim_matrix_bwd(g.float().contiguous(), ctx.ws, n, m, d)
        return da, db, None


def sim_matrix(a, b, eps=1e-8):
    """Cosine-similarity matrix with the norm clamped at eps
    (/root/reference/OATrans/model/oa_model.py:192-200, model.py:164-172)."""
    if not a.is_cuda:
        raise hip.OatError("sim_matrix runs on MI355X only (no CPU path); use the oracle for CPU")
    return _SimFn.apply(a, b, eps)


class _InfoNCEFn(torch.autograd.Function):
    """NormSoftmaxLoss(sim_matrix(t, v)) (trainer_dist.py:159-163: loss.py:13-25 on oa_model.py:192-200) as ONE function on the
    (gathered) embeddings: oat_infonce produces the loss and both embedding gradients in its forward launches; backward scales
    them.  Same kernels as the two-function path (sim_matrix + NormSoftmaxLoss), without the autograd glue between them."""

    @staticmethod
    def forward(ctx, t, v, temperature, eps):
        t, v = t.float().contiguous(), v.float().contiguous()
        loss, _, dt, dv = hip.infonce(t, v, temperature, eps, want_grads=True)
        ctx.save_for_backward(dt, dv)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dt, dv = ctx.saved_tensors
        return dt * g, dv * g, None, None


def infonce_loss(t, v, temperature=0.05, eps=1e-8):
    """loss_fn(sim_matrix(t, v)) for a NormSoftmaxLoss(temperature) in one autograd node (t, v: [n, d] embeddings of all ranks).
    Under no_grad, or when neither side wants a gradient (validation), only the loss is computed."""
    if not t.is_cuda:
        raise hip.OatError("infonce_loss runs on MI355X only (no CPU path); use the oracle for CPU")
    if not (torch.is_grad_enabled() and (t.requires_grad or v.requires_grad)):
        loss, _, _, _ = hip.infonce(t.float().contiguous(), v.float().contiguous(), temperature, eps, want_grads=False)
        return loss.reshape(())
    return _InfoNCEFn.apply(t, v, temperature, eps)


class _NormSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, temperature):
        loss, G = hip.norm_softmax_loss(x.float().contiguous(), temperature, want_grad=True)
        ctx.save_for_backward(G)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (G,) = ctx.saved_tensors
        return G * g, None
