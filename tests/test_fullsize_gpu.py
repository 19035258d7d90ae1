"""Size-independent properties of the kernels at the FULL benchmark size (B=32, 8 frames, 196 patches, M=50208 rows),
where a CPU oracle comparison would take minutes: identities that must hold exactly or to fp32 round-off."""
import pytest
import torch

pytestmark = pytest.mark.gpu
B, T, N, H = 32, 8, 196, 12
D = H * 64
M = B * T * N + B
Mp = (M + 255) // 256 * 256


def _hip():
    from OATrans.ops import hip
    return hip


def test_gemm_nt_with_identity_weight_returns_its_input():
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda")
    A[:M] = torch.randn(M, D, device="cuda", generator=g).bfloat16()
    eye = torch.eye(D, device="cuda").bfloat16()
    out = torch.full((Mp, D), 7.0, dtype=torch.bfloat16, device="cuda")
    hip.gemm_nt(A, eye, M, D, D, hip.EPI_BF16, out)
    assert torch.equal(out[:M], A[:M])
    assert torch.all(out[M:] == 7.0)                       # rows beyond M are never written
    # the 3-tile-wide shapes too (persistent walk over 1773 tiles): out = [A | A | A]
    eye3 = torch.cat([eye, eye, eye], 0).contiguous()
    out3 = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda")
    hip.gemm_nt(A, eye3, M, 3 * D, D, hip.EPI_BF16, out3)
    assert torch.equal(out3[:M], torch.cat([A[:M]] * 3, 1))


def test_gemm_tn_against_column_sums():
    """dW = P^T Q with Q = all-ones reduces to column sums of P; the fused bias output must agree with it."""
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(2)
    P = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda")
    P[:M] = torch.randn(M, D, device="cuda", generator=g).bfloat16()
    Q = torch.zeros(Mp, 256, dtype=torch.bfloat16, device="cuda")
    Q[:M] = 1.0
    out = torch.zeros(D, 256, device="cuda")
    bias = torch.zeros(D, device="cuda")
    hip.gemm_tn(P, Q, M, D, 256, out, bias_out=bias)
    ref = P[:M].double().sum(0)
    tol = 1e-5 * P[:M].double().abs().sum(0).max().item() + 1e-3
    assert (out.double() - ref[:, None]).abs().max().item() < tol
    assert (bias.double() - ref).abs().max().item() < tol
    assert (out - out[:, :1]).abs().max().item() < 1e-3     # every column of dW holds the same sum


@pytest.mark.parametrize("mode", ["space", "time"])
def test_attention_of_constant_values_returns_the_constant(mode):
    """softmax rows sum to one: with every V row equal to c the output is c for any Q, K; the LSE stays finite."""
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda")
    qkv[:M, :2 * D] = (2.0 * torch.randn(M, 2 * D, device="cuda", generator=g)).bfloat16()
    c = torch.randn(D, device="cuda", generator=g).bfloat16()
    qkv[:M, 2 * D:] = c
    out = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(Mp, H, device="cuda")
    (hip.attn_space_fwd if mode == "space" else hip.attn_time_fwd)(qkv, out, lse, B, T, N, H, D, 0.125)
    hip.attn_cls_fwd(qkv, out, lse, B, T, N, H, D, 0.125)
    err = (out[:M].float() - c.float()).abs().max().item()
    assert err <= 2e-2 * c.float().abs().max().item(), err          # one bf16 rounding of a convex combination
    assert torch.isfinite(lse[:M]).all()


def test_layernorm_rows_are_standardised():
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.zeros(Mp, D, device="cuda")
    x[:M] = 3.0 * torch.randn(M, D, device="cuda", generator=g) + 5.0
    y = torch.zeros(Mp, D, device="cuda")
    mean = torch.zeros(Mp, device="cuda")
    rstd = torch.zeros(Mp, device="cuda")
    hip.layernorm_fwd(x, torch.ones(D, device="cuda"), torch.zeros(D, device="cuda"), M, D, 1e-6, y32=y, mean=mean, rstd=rstd)
    assert y[:M].mean(1).abs().max().item() < 1e-4
    assert (y[:M].var(1, unbiased=False) - 1).abs().max().item() < 1e-3
    assert (mean[:M] - x[:M].mean(1)).abs().max().item() < 1e-4
