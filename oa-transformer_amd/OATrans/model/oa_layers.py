"""HIP-backed autograd functions for the object-aware tails: patch-mean pooling,
the two einsums and the region BCE.  Each wraps oat_* entry points of csrc/oa_extras.hip."""
import torch

from ..ops import hip


def _need_cuda(t, what):
    if not t.is_cuda:
        raise hip.OatError(f"{what} runs on MI355X only (no CPU path); use the oracle for CPU")


class _MeanRowsFn(torch.autograd.Function):
    """x [G, R, D] fp32 -> mean over R -> [G, D]"""

    @staticmethod
    def forward(ctx, x):
        x = x.float().contiguous()
        G, R, D = x.shape
        out = torch.empty(G, D, dtype=torch.float32, device=x.device)
        # one workgroup per group: with few long groups (32 clips x 1568 patch rows) only G CUs would stream the 154 MB.
        # Two deterministic stages instead: G x S slices of R / S rows, then the S partial rows of every group.
        S = next((s_ for s_ in range(min(32, R // 16), 1, -1) if R % s_ == 0), 1) if G < 256 and R >= 64 else 1
        if S > 1:
            part = torch.empty(G * S, D, dtype=torch.float32, device=x.device)
            hip.grouped_rowsum(x.view(G * R, D), G * S, R // S, D, part)
            hip.grouped_rowsum(part, G, S, D, out)
        else:
            hip.grouped_rowsum(x.view(G * R, D), G, R, D, out)
        out.mul_(1.0 / R)
        ctx.shape = (G, R, D)
        return out

    @staticmethod
    def backward(ctx, g):
        G, R, D = ctx.shape
        dx = torch.empty(G * R, D, dtype=torch.float32, device=g.device)
        hip.grouped_broadcast(g.float().contiguous(), dx, G, R, D, scale=1.0 / R)
        return dx.view(G, R, D)


def mean_rows(x):
    _need_cuda(x, "mean_rows")
    return _MeanRowsFn.apply(x)


class _MixFn(torch.autograd.Function):
    """alpha * a + beta * b"""

    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        a, b = a.float().contiguous(), b.float().contiguous()
        out = torch.empty_like(a)
        hip.axpby(a, b, out, alpha, beta)
        ctx.ab = (alpha, beta)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.float().contiguous()
        ga, gb = torch.empty_like(g), torch.empty_like(g)
        hip.axpby(g, None, ga, ctx.ab[0])
        hip.axpby(g, None, gb, ctx.ab[1])
        return ga, gb, None, None


def mix(a, b, alpha, beta):
    _need_cuda(a, "mix")
    return _MixFn.apply(a, b, alpha, beta)


class _BmmFn(torch.autograd.Function):
    """Batched C = A @ B (nt=False: B is [b,k,j]; nt=True: B is [b,j,k]), optional sigmoid on C.
    Inputs may be arbitrary strided views (no copies)."""

    @staticmethod
    def forward(ctx, A, B, nt, sigmoid):
        A, B = A.float(), B.float()
        nb, I, K = A.shape
        J = B.shape[1] if nt else B.shape[2]
        C = torch.empty(nb, I, J, dtype=torch.float32, device=A.device)
        sB = (B.stride(0), B.stride(2), B.stride(1)) if nt else tuple(B.stride())
        hip.bmm_strided(A, B, C, nb, I, J, K, tuple(A.stride()), sB, tuple(C.stride()), sigmoid=sigmoid)
        ctx.save_for_backward(A, B, C if sigmoid else None)
        ctx.nt, ctx.sigmoid = nt, sigmoid
        return C

    @staticmethod
    def backward(ctx, dC):
        A, B, C = ctx.saved_tensors
        dC = dC.float().contiguous()
        if ctx.sigmoid:
            dZ = torch.empty_like(dC)
            hip.sigmoid_bwd(C, dC, dZ)
            dC = dZ
        nb, I, K = A.shape
        J = dC.shape[2]
        dA = torch.empty(nb, I, K, dtype=torch.float32, device=dC.device)
        # dA[b,i,k] = sum_j dC[b,i,j] * B(k,j)
        sBt = (B.stride(0), B.stride(1), B.stride(2)) if ctx.nt else (B.stride(0), B.stride(2), B.stride(1))
        hip.bmm_strided(dC, B, dA, nb, I, K, J, tuple(dC.stride()), sBt, tuple(dA.stride()))
        # dB: nt -> [b,j,k] = sum_i dC[b,i,j] A[b,i,k] ; else [b,k,j] = sum_i A[b,i,k] dC[b,i,j]
        sAt = (A.stride(0), A.stride(2), A.stride(1))          # A viewed as [b,k,i]
        sdCt = (dC.stride(0), dC.stride(2), dC.stride(1))      # dC viewed as [b,j,i]
        if ctx.nt:
            dB = torch.empty(nb, J, K, dtype=torch.float32, device=dC.device)
            hip.bmm_strided(dC, A, dB, nb, J, K, I, sdCt, tuple(A.stride()), tuple(dB.stride()))
        else:
            dB = torch.empty(nb, K, J, dtype=torch.float32, device=dC.device)
            hip.bmm_strided(A, dC, dB, nb, K, J, I, sAt, tuple(dC.stride()), tuple(dB.stride()))
        return dA, dB, None, None


def mask_pool(masks, feats):
    """einsum('b o l, b l c -> b o c') (oa_model_global_local.py:178,200)"""
    _need_cuda(feats, "mask_pool")
    if masks.dim() != 3 or feats.dim() != 3 or masks.shape[0] != feats.shape[0] or masks.shape[2] != feats.shape[1]:
        # the reference's einsum raises on the same inputs: patch masks live on ONE frame's 14x14 grid, so the OA
        # clips must be single-frame (2-frame loader batches)
        raise RuntimeError(f"mask_pool: einsum('b o l, b l c -> b o c') needs masks [B,O,L] and feats [B,L,C]; got "
                           f"{tuple(masks.shape)} and {tuple(feats.shape)}")
    return _BmmFn.apply(masks, feats, False, False)


def region_sim(text_regions, object_regions):
    """sigmoid(einsum('b k f, b n f -> b k n')) (oa_model_region_mem.py:147-151)"""
    _need_cuda(object_regions, "region_sim")
    if text_regions.dim() != 3 or object_regions.dim() != 3 or text_regions.shape[0] != object_regions.shape[0] \
            or text_regions.shape[2] != object_regions.shape[2]:
        raise RuntimeError(f"region_sim: einsum('b k f, b n f -> b k n') got {tuple(text_regions.shape)} and "
                           f"{tuple(object_regions.shape)}")
    return _BmmFn.apply(text_regions, object_regions, True, True)


class _BceSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, y):
        p, y = p.float().contiguous(), y.float().contiguous()
        ctx.save_for_backward(p, y)
        return hip.bce_sum(p, y).reshape(())

    @staticmethod
    def backward(ctx, g):
        p, y = ctx.saved_tensors
        dp = torch.empty_like(p)
        hip.bce_bwd(p, y, g.float().reshape(1).contiguous(), dp)
        return dp, None


def bce_sum(p, y):
    """nn.BCELoss(reduction='sum') (trainer_region_mem.py:97,166)"""
    _need_cuda(p, "bce_sum")
    if p.shape != y.shape:        # F.binary_cross_entropy raises the same way (target size != input size)
        raise ValueError(f"Using a target size ({tuple(y.shape)}) that is different to the input size ({tuple(p.shape)}) "
                         "is deprecated. Please ensure they have the same size.")
    return _BceSumFn.apply(p, y)
