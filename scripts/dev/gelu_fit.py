"""Where the coefficients of csrc/common.h: gelu_pair come from, and what they cost in accuracy (CPU only, numpy + scipy).

q(a) = Phi(-a) exp(a^2 / 2) = erfcx(a / sqrt 2) / 2 on [0, 6] is fitted by a degree-8 polynomial in a, minimax in the RELATIVE
error (Lawson's iteratively re-weighted least squares on 6000 Chebyshev nodes).  Then the kernel's arithmetic is emulated in
float32 on 2e6 points of [-8, 8] and compared with float64 erf-GELU: bf16(gelu) within 0.58 bf16 ulp or 1e-7 absolute,
gelu' within 1.5e-4 absolute; the Abramowitz-Stegun form it replaced (absolute error 1.5e-7 on erf) reaches 1.8 ulp in the
negative tail.  The GPU test of the same statement: tests/test_kernels_gpu.py::test_gelu_epilogue_dense_sweep_vs_fp64_erf."""
import numpy as np
from scipy.special import erfcx, erf, erfc
np.set_printoptions(precision=10)
A = 6.0
def q(a): return 0.5 * erfcx(a / np.sqrt(2))
n = 8
k = np.arange(6000)
xs = 0.5 * A * (1 - np.cos(np.pi * (k + 0.5) / 6000))
V = np.vander(xs, n + 1, increasing=True)
w = 1.0 / q(xs)
lw = np.ones_like(xs) / len(xs)
for it in range(400):
    sw = np.sqrt(lw)
    c, *_ = np.linalg.lstsq(V * (w * sw)[:, None], q(xs) * w * sw, rcond=None)
    r = (V @ c - q(xs)) * w
    lw = lw * (np.abs(r) + 1e-30); lw /= lw.sum()
a = np.linspace(0, A, 200001)
p = np.polyval(c[::-1], a)
print("max rel err", np.max(np.abs(p / q(a) - 1)))
c32 = c.astype(np.float32)
print("coef (c0..c8):", ", ".join(f"{v:.9e}f" for v in c32))

# float32 emulation of the kernel arithmetic (fma emulated in float64 then rounded: close enough for an error survey)
def f32(x): return np.asarray(x, dtype=np.float32)
def fma(a, b, c): return f32(a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64))
def kernel(x):
    x = f32(x)
    a = np.minimum(np.abs(x), f32(6.0))
    t = f32(a * a)
    earg = f32(t * f32(-0.72134752))
    e = f32(np.exp2(earg.astype(np.float64)))
    qq = np.full_like(a, c32[8])
    for j in range(7, -1, -1):
        qq = fma(qq, a, np.full_like(a, c32[j]))
    u = f32(e * qq)
    r = np.maximum(x, f32(0))
    gl = fma(-a, u, r)
    wv = fma(a, np.full_like(a, f32(0.3989422804)), -qq)
    z = fma(wv, e, np.full_like(a, f32(0.5)))
    dgh = np.copysign(z, x)
    return gl, f32(dgh + f32(0.5))
x = np.linspace(-8, 8, 2000001)
gl, dg = kernel(x)
xd = x.astype(np.float32).astype(np.float64)
Phi = 0.5 * erfc(-xd / np.sqrt(2))
gle = xd * Phi
dge = Phi + xd * np.exp(-xd * xd / 2) / np.sqrt(2 * np.pi)
def bf16_round(v):
    u = np.asarray(v, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)
def ulp_bf16(v):
    e = np.floor(np.log2(np.maximum(np.abs(v), 1e-45)))
    return 2.0 ** (e - 7)
g16 = bf16_round(gl).astype(np.float64)
err = np.abs(g16 - gle)
ok = (err <= ulp_bf16(gle)) | (err <= 1e-7)
print("gelu: worst err/ulp where ulp criterion applies:", np.max(np.where(err > 1e-7, err / ulp_bf16(gle), 0)), "all ok:", ok.all())
print("gelu fp32 value: max rel err (|x|<6)", np.max(np.abs(gl.astype(np.float64) - gle)[np.abs(xd) < 6] / np.abs(gle[np.abs(xd) < 6] + 1e-300)))
print("gelu' max abs err", np.max(np.abs(dg.astype(np.float64) - dge)))
# current A-S kernel for comparison
def as_kernel(x):
    x = f32(x); m = f32(np.abs(x) * f32(0.8493218002880191))
    t = f32(1.0 / (m.astype(np.float64) * 0.2727374808792225 + 1.0))
    p = np.full_like(m, f32(0.5307027145))
    for cc in (-0.7265760135, 0.7107068705, -0.142248368, 0.127414796):
        p = fma(p, t, np.full_like(m, f32(cc)))
    e = f32(np.exp2(-(m.astype(np.float64) ** 2)))
    cdf = f32(0.5) + np.copysign(fma(-f32(p * t), e, np.full_like(m, f32(0.5))), x)
    return f32(x * cdf)
g_as = as_kernel(x)
err2 = np.abs(bf16_round(g_as).astype(np.float64) - gle)
print("A-S kernel: worst err/ulp:", np.max(np.where(err2 > 1e-7, err2 / ulp_bf16(gle), 0)))
