// HBM-bound row kernels: LayerNorm fwd/bwd, column reductions, im2col, positional tables.
// One wave (64 lanes) owns one token row; lanes read float4 (16 B) so a wave instruction
// moves 1 KiB contiguous.  All statistics and reductions are fp32.
//
// Reference ops replaced (all /root/reference/OATrans/model/video_transformer.py):
//   norm1/norm2/norm3/norm   :164,167,174,346  (nn.LayerNorm eps=1e-6, :228)
//   VideoPatchEmbed conv     :69-75            (im2col feeding the patch GEMM)
//   cls/pos/temporal tables  :313-324
#include "common.h"
#include "fp8.h"
#include <stdlib.h>

namespace oat {

constexpr int LN_MAXV = 4;   // up to 4 float4 per lane -> D <= 1024

// ------------------------------------------------------------------ LayerNorm forward
// y = (x - mean) * rstd * gamma + beta ; writes bf16 y (GEMM operand) and optionally fp32 y32.
// Optional fused residual add: when `add16` is given the normalised row is s = x + add16 (bf16 branch
// output of the preceding GEMM) and s is also written to `sum32` (the new fp32 residual stream).  This
// moves the fp32 read-modify-write of the stream out of the GEMM epilogue (where it runs un-overlapped
// at ~2.3 TB/s) into this streaming kernel (5.5+ TB/s) without adding bytes.
// fp8 forward (fp8.hip): y8 != nullptr additionally writes the normalised row as OCP e4m3, quantised with the site's
// delayed scale *qscale, and records max |y| for the next step's scale.
struct LnF8 { uint8_t* y8; int ld8; const float* qscale; float* amax; };
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16* y, int ldy,
                                                     float* y32, int ldy32, float* mean, float* rstd,
                                                     int M, int D, float eps, const bf16* add16, int ldadd,
                                                     float* sum32, int ldsum, const float* add32, int ldadd32, LnF8 f8,
                                                     const bf16* add16b, int ldaddb) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const float qs = f8.y8 ? f8.qscale[0] : 0.f;
  float m8 = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float* xr = x + (size_t)row * ldx;
    f32x4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        v[i] = *reinterpret_cast<const f32x4*>(xr + c);
        if (add16) {
          const bf16x4 t = *reinterpret_cast<const bf16x4*>(add16 + (size_t)row * ldadd + c);
          v[i] += f32x4{bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])};
          if (add16b) {                            // second bf16 branch output (x + space + mlp formed in one pass)
            const bf16x4 u = *reinterpret_cast<const bf16x4*>(add16b + (size_t)row * ldaddb + c);
            v[i] += f32x4{bf2f(u[0]), bf2f(u[1]), bf2f(u[2]), bf2f(u[3])};
          }
          if (sum32) *reinterpret_cast<f32x4*>(sum32 + (size_t)row * ldsum + c) = v[i];
        } else if (add32) {                        // fp32 addend (the precise CLS lane of the video tower)
          v[i] += *reinterpret_cast<const f32x4*>(add32 + (size_t)row * ldadd32 + c);
          if (sum32) *reinterpret_cast<f32x4*>(sum32 + (size_t)row * ldsum + c) = v[i];
        }
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
      }
    }
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; q += d * d; }
      }
    }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
    if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        // gamma == nullptr: the plain normalised row xhat (the affine map is folded into the next linear layer's weights)
        const f32x4 g = gamma ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 b = beta ? *reinterpret_cast<const f32x4*>(beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + b[e];
        if (y) {
          bf16x4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
          *reinterpret_cast<bf16x4*>(y + (size_t)row * ldy + c) = ob;
        }
        if (y32) *reinterpret_cast<f32x4*>(y32 + (size_t)row * ldy32 + c) = o;
        if (f8.y8) {
          m8 = fmaxf(fmaxf(m8, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
          *reinterpret_cast<uint32_t*>(f8.y8 + (size_t)row * f8.ld8 + c) = pack_fp8x4(o[0] * qs, o[1] * qs, o[2] * qs, o[3] * qs);
        }
      }
    }
  }
  if (f8.y8) amax_commit(m8, f8.amax);        // block-uniform branch (kernel argument)
}

// ------------------------------------------------------------------ LayerNorm forward on a bf16 residual stream
// s = x + add_a + add_b ; sum16 = bf16(s) ; y = LN(s) computed from the UNROUNDED fp32 sum.
// The residual stream of the video tower is STORED as bf16 (round 4; scripts/dev/rounding_study3.py: the cosine-similarity
// matrix - which is taken from the fp32 CLS lane - does not move, parameter gradients go from 1.8e-2 to 2.0e-2 relative L2
// against the fp32 oracle): x is 2 bytes per element instead of 4 and so is the new stream the kernel leaves, i.e.
// 385 instead of 539 MB for the block-opening LayerNorm (x + space + mlp), 231 instead of 308 MB for the other two at
// M = 50208, D = 768.  X32: x is still fp32 (block 0 reads the patch embedding's fp32 output).
// Access shape: these kernels are bound by memory REQUESTS in flight, not by bytes - the first form (a wave per row, 8-byte
// bf16x4 lane accesses) ran exactly as long as the fp32 kernel it replaced.  So: HALF a wave (32 lanes) per row, every lane
// access 16 bytes (8 bf16 / 2 x 4 fp32), 8 rows per workgroup; row sums by xor-shuffles below 32 (they stay inside the half).
constexpr int LN_MAXC = 4;   // up to 4 chunks of 8 elements per lane -> D <= 1024, D % 8 == 0
OAT_DEV float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
OAT_DEV void ld8(const bf16* p, float (&v)[8]) {
  const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = bf2f(t[e]);
}
OAT_DEV void ld8(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
OAT_DEV void add8(const bf16* p, float (&v)[8]) {
  const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] += bf2f(t[e]);
}
OAT_DEV void st8(bf16* p, const float (&v)[8]) {
  bf16x8 t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = f2bf(v[e]);
  *reinterpret_cast<bf16x8*>(p) = t;
}
OAT_DEV void st8(float* p, const float (&v)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
// F8 (fp8 forward on the bf16 stream): the normalised row additionally leaves as OCP e4m3 in f8.y8, quantised with the site's
// delayed scale, 8 bytes per lane and chunk, and max |y| is recorded for the next step's scale - as ln_fwd_kernel does on
// the fp32 stream; the fp8 GEMM that consumes the row needs no quantisation pass.
template <bool X32, bool F8 = false>
__global__ __launch_bounds__(256) void ln_fwd_r16_kernel(const void* __restrict__ x_, int ldx, const bf16* __restrict__ add_a,
                                                         int lda, const bf16* __restrict__ add_b, int ldb, bf16* sum16,
                                                         int ldsum, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, bf16* y, int ldy, float* y32,
                                                         int ldy32, float* mean, float* rstd, int M, int D, float eps,
                                                         LnF8 f8 = LnF8{nullptr, 0, nullptr, nullptr}) {
  const int lane = threadIdx.x & 31;
  const int hw = threadIdx.x >> 5;
  const float qs = F8 ? f8.qscale[0] : 0.f;
  float m8 = 0.f;
  for (int row = blockIdx.x * 8 + hw; row < M; row += gridDim.x * 8) {
    float v[LN_MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
      const int c = (i * 32 + lane) * 8;
      if (c < D) {
        if constexpr (X32) ld8((const float*)x_ + (size_t)row * ldx + c, v[i]);
        else ld8((const bf16*)x_ + (size_t)row * ldx + c, v[i]);
        if (add_a) add8(add_a + (size_t)row * lda + c, v[i]);
        if (add_b) add8(add_b + (size_t)row * ldb + c, v[i]);
        if (sum16) st8(sum16 + (size_t)row * ldsum + c, v[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
      }
    }
    const float mu = half_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
      const int c = (i * 32 + lane) * 8;
      if (c < D) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q += d * d; }
      }
    }
    const float rs = rsqrtf(half_sum(q) / D + eps);
    if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
      const int c = (i * 32 + lane) * 8;
      if (c < D) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mu) * rs;
        if (gamma) {
          float g[8];
          ld8(gamma + c, g);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] *= g[e];
        }
        if (beta) {
          float b[8];
          ld8(beta + c, b);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += b[e];
        }
        if (y) st8(y + (size_t)row * ldy + c, o);
        if (y32) st8(y32 + (size_t)row * ldy32 + c, o);
        if constexpr (F8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) m8 = fmaxf(m8, fabsf(o[e]));
          uint2 w;
          w.x = pack_fp8x4(o[0] * qs, o[1] * qs, o[2] * qs, o[3] * qs);
          w.y = pack_fp8x4(o[4] * qs, o[5] * qs, o[6] * qs, o[7] * qs);
          *reinterpret_cast<uint2*>(f8.y8 + (size_t)row * f8.ld8 + c) = w;
        }
      }
    }
  }
  if constexpr (F8) amax_commit(m8, f8.amax);
}

// ------------------------------------------------------------------ LayerNorm backward
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// out fp32 dx (+ optional fp32 addend `dres`), optional bf16 copy; per-block partial
// (dgamma, dbeta) sums go to part[blockIdx][2][D] and are finished by reduce_partials.
// X16: the forward input x is the bf16 residual stream; dres16: a bf16 residual-gradient addend (may alias dx16: a lane reads
// and writes the same elements).
template <bool DY_BF16, bool X16>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* dy_, int lddy, const void* __restrict__ x_,
                                                     int ldx, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* dres,
                                                     int lddres, float* dx, int lddx, bf16* dx16, int lddx16,
                                                     int dx16_excl_res, float* part, int M, int D, LnF8 f8,
                                                     const bf16* dres16, int lddres16) {
  __shared__ float red[4][2][LN_MAXV * 256];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const float qs = f8.y8 ? f8.qscale[0] : 0.f;        // fp8 backward: e5m2 copy of dx16 for the next data-gradient GEMM
  float m8 = 0.f;
  f32x4 ag[LN_MAXV], ab[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) { ag[i] = f32x4{0, 0, 0, 0}; ab[i] = f32x4{0, 0, 0, 0}; }
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[LN_MAXV], g[LN_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        f32x4 xv;
        if constexpr (X16) {
          const bf16x4 t = *reinterpret_cast<const bf16x4*>((const bf16*)x_ + (size_t)row * ldx + c);
          xv = f32x4{bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])};
        } else {
          xv = *reinterpret_cast<const f32x4*>((const float*)x_ + (size_t)row * ldx + c);
        }
        f32x4 dyv;
        if constexpr (DY_BF16) {
          const bf16x4 t = *reinterpret_cast<const bf16x4*>((const bf16*)dy_ + (size_t)row * lddy + c);
          dyv = f32x4{bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])};
        } else {
          dyv = *reinterpret_cast<const f32x4*>((const float*)dy_ + (size_t)row * lddy + c);
        }
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[i][e] = (xv[e] - mu) * rs;
          g[i][e] = dyv[e] * gm[e];
          s1 += g[i][e];
          s2 += g[i][e] * xh[i][e];
          ag[i][e] += dyv[e] * xh[i][e];
          ab[i][e] += dyv[e];
        }
      }
    }
    const float c1 = wave_sum(s1) / D, c2 = wave_sum(s2) / D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (g[i][e] - c1 - xh[i][e] * c2);
        const f32x4 o_nores = o;
        if (dres) o += *reinterpret_cast<const f32x4*>(dres + (size_t)row * lddres + c);
        if (dres16) {
          const bf16x4 t = *reinterpret_cast<const bf16x4*>(dres16 + (size_t)row * lddres16 + c);
          o += f32x4{bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])};
        }
        if (dx) *reinterpret_cast<f32x4*>(dx + (size_t)row * lddx + c) = o;
        if (dx16) {
          const f32x4 w = dx16_excl_res ? o_nores : o;
          bf16x4 ob = {f2bf(w[0]), f2bf(w[1]), f2bf(w[2]), f2bf(w[3])};
          *reinterpret_cast<bf16x4*>(dx16 + (size_t)row * lddx16 + c) = ob;
          if (f8.y8) {
            m8 = fmaxf(fmaxf(m8, fmaxf(fabsf(w[0]), fabsf(w[1]))), fmaxf(fabsf(w[2]), fabsf(w[3])));
            *reinterpret_cast<uint32_t*>(f8.y8 + (size_t)row * f8.ld8 + c) = pack_bf8x4(w[0] * qs, w[1] * qs, w[2] * qs, w[3] * qs);
          }
        }
      }
    }
  }
  if (f8.y8) amax_commit(m8, f8.amax);
  if (!part) return;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[wave][0][c + e] = ag[i][e]; red[wave][1][c + e] = ab[i][e]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    part[((size_t)blockIdx.x * 2 + 0) * D + c] = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
    part[((size_t)blockIdx.x * 2 + 1) * D + c] = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
  }
}

// LayerNorm backward in "folded" form: the layer's affine map lives in the next linear layer's weights (W' = W diag(gamma),
// b' = b + W beta, see oat_fold_bias_multi / oat_ln_fold_grads), so the data-gradient GEMM delivers d(xhat) directly and the
// saved GEMM operand IS xhat (bf16):   dx = rstd * (dxh - mean(dxh) - xhat * mean(dxh * xhat)).
// Against ln_bwd_kernel: reads 2 B instead of 4 B per element of the forward input and produces no (dgamma, dbeta)
// partials - 539 instead of 616 MB per call at M = 50208, D = 768.
// The residual-gradient stream G (fp32) need not be read and re-written by every LayerNorm of a block: norm2's backward
// only READS it (for the bf16 sum the space branch's data gradient needs) and leaves its own dx as bf16 (`dxp16`), norm1's
// touches neither, and norm3's adds both bf16 increments (`add_a`, `add_b`) when it forms the block's outgoing G - the
// increments, not the stream, are rounded to bf16.  1386 instead of 1617 MB per block.
__global__ __launch_bounds__(256) void ln_bwd_xhat_kernel(const bf16* __restrict__ dxh, int lddxh, const bf16* __restrict__ xh16,
                                                          int ldxh, const float* __restrict__ rstd, const float* dres,
                                                          int lddres, float* dx, int lddx, bf16* dx16, int lddx16,
                                                          int dx16_excl_res, int M, int D, const bf16* add_a, int ldadd_a,
                                                          const bf16* add_b, int ldadd_b, bf16* dxp16, int lddxp) {
  // half a wave per row, 16-byte lane accesses (see ln_fwd_r16_kernel)
  const int lane = threadIdx.x & 31;
  const int hw = threadIdx.x >> 5;
  for (int row = blockIdx.x * 8 + hw; row < M; row += gridDim.x * 8) {
    const float rs = rstd[row];
    float xh[LN_MAXC][8], g[LN_MAXC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
      const int c = (i * 32 + lane) * 8;
      if (c < D) {
        ld8(xh16 + (size_t)row * ldxh + c, xh[i]);
        ld8(dxh + (size_t)row * lddxh + c, g[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1 += g[i][e]; s2 += g[i][e] * xh[i][e]; }
      }
    }
    const float c1 = half_sum(s1) / D, c2 = half_sum(s2) / D;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
      const int c = (i * 32 + lane) * 8;
      if (c < D) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rs * (g[i][e] - c1 - xh[i][e] * c2);
        if (dxp16) st8(dxp16 + (size_t)row * lddxp + c, o);
        if (dx16 && dx16_excl_res) st8(dx16 + (size_t)row * lddx16 + c, o);
        if (dres) {
          float r[8];
          ld8(dres + (size_t)row * lddres + c, r);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += r[e];
        }
        if (add_a) add8(add_a + (size_t)row * ldadd_a + c, o);
        if (add_b) add8(add_b + (size_t)row * ldadd_b + c, o);
        if (dx) st8(dx + (size_t)row * lddx + c, o);
        if (dx16 && !dx16_excl_res) st8(dx16 + (size_t)row * lddx16 + c, o);
      }
    }
  }
}

// Folded LayerNorm, weights side.  A linear layer z = W (gamma * xhat + beta) + b is run as z = W' xhat + b' with
// W' = W diag(gamma) (cast_bf16_multi's column scale) and b' = b + W beta:
//   fold_bias: out[n] = b[n] + sum_k W[n, k] beta[k]     one wave per output row; table-driven (all folded layers in one launch)
struct FoldBiasDesc { const float* W; const float* beta; const float* b; float* out; long long N, K, first_row, pad; };
__global__ __launch_bounds__(256) void fold_bias_multi_kernel(const FoldBiasDesc* desc, const int* block_desc) {
  const FoldBiasDesc d = desc[block_desc[blockIdx.x]];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = (int)(blockIdx.x * 4 + wave - d.first_row);
  if (n >= (int)d.N) return;
  const float* w = d.W + (size_t)n * d.K;
  float s = 0.f;
  for (int k = lane * 4; k < (int)d.K; k += 256) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k), bv = *reinterpret_cast<const f32x4*>(d.beta + k);
    s += wv[0] * bv[0] + wv[1] * bv[1] + wv[2] * bv[2] + wv[3] * bv[3];
  }
  s = wave_sum(s);
  if (lane == 0) d.out[n] = s + (d.b ? d.b[n] : 0.f);
}
// ... and the gradients.  The weight-gradient GEMM on xhat yields dW' = dz^T xhat and db' = colsum(dz); then
//   dW[n, k]   (+)= dW'[n, k] gamma[k] + db'[n] beta[k]        db (+)= db'   (only when db != db': accumulate mode)
//   dgamma[k]  (+)= sum_n W[n, k] dW'[n, k]                      dbeta[k] (+)= sum_n W[n, k] db'[n]
// Grid: (64-column strip, row slice) per layer - FG_SPLIT row slices per strip so that ~300 workgroups stream the 70 MB of
// a block's three layers (one workgroup per strip took 107 us: 190 dependent iterations per thread).  A workgroup is 16
// float4 column lanes x 16 row lanes, 4 rows in flight per thread; its (dgamma, dbeta) partial sums go to `part`, and the
// LAST slice of a strip to arrive (ticket counter; partials travel by agent-scope atomics only, so no fence is needed) adds
// the FG_SPLIT partials in slice order - a fixed order whoever comes last: deterministic.  The counter resets itself.
// (The hand-over below - relaxed agent-scope atomics ordered by `s_waitcnt vmcnt(0)` alone - relies on gfx9 behaviour: vmcnt
// covers stores and returnless atomics, atomics execute at the memory side.  This file is built for gfx950 only.)
constexpr int FG_SPLIT = 16, FG_COLS = 64;
struct FoldGradDesc {
  const float* dWp; const float* dbp; const float* W; const float* gamma; const float* beta;
  float* dW; float* db; float* dgamma; float* dbeta;
  long long N, K, first_block, accumulate;
};
__global__ __launch_bounds__(256) void ln_fold_grads_kernel(const FoldGradDesc* desc, int n_desc, float* part, int* ticket) {
  __shared__ float red[2][16][FG_COLS + 4];
  __shared__ int s_last;
  int di = 0;
  while (di + 1 < n_desc && desc[di + 1].first_block <= (long long)blockIdx.x) ++di;
  const FoldGradDesc d = desc[di];
  const int local = (int)(blockIdx.x - d.first_block), strip = local / FG_SPLIT, slice = local - strip * FG_SPLIT;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = strip * FG_COLS + cl * 4;
  const int N = (int)d.N, K = (int)d.K;
  const bool acc = d.accumulate != 0;
  const int rows = (N + FG_SPLIT - 1) / FG_SPLIT, n0 = slice * rows, n1 = min(n0 + rows, N);
  f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
  if (c < K) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(d.gamma + c), bt = *reinterpret_cast<const f32x4*>(d.beta + c);
    for (int n = n0 + rl; n < n1; n += 64) {
      f32x4 w[4], gw[4];
      float gb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int nn = n + u * 16;
        const bool ok = nn < n1;
        const size_t o = (size_t)(ok ? nn : n) * K + c;
        w[u] = *reinterpret_cast<const f32x4*>(d.W + o);
        gw[u] = *reinterpret_cast<const f32x4*>(d.dWp + o);
        gb[u] = d.dbp[ok ? nn : n];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int nn = n + u * 16;
        if (nn < n1) {
          sg += w[u] * gw[u];
          sb += w[u] * gb[u];
          f32x4 v = gw[u] * gm + bt * gb[u];
          f32x4* dst = reinterpret_cast<f32x4*>(d.dW + (size_t)nn * K + c);
          if (acc) v += *dst;
          *dst = v;
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][rl][cl * 4 + e] = sg[e]; red[1][rl][cl * 4 + e] = sb[e]; }
  __syncthreads();
  // this slice's partial sums of the strip: part[(block) * 2 * FG_COLS + {0: dgamma, 1: dbeta} * FG_COLS + column]
  float* mine = part + (size_t)blockIdx.x * 2 * FG_COLS;
  if (threadIdx.x < 2 * FG_COLS) {
    const int which = threadIdx.x / FG_COLS, col = threadIdx.x % FG_COLS;
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += red[which][r][col];
    // published by an atomic exchange (performed at the memory side, acknowledged before the ticket is drawn): no release
    // fence, i.e. no write-back of the ~40 KB of dW this workgroup has just dirtied in its L2
    (void)__hip_atomic_exchange(mine + threadIdx.x, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (d.db != d.dbp && local == 0) {                            // accumulate mode: the bias gradient arrived in a scratch vector
    for (int n = threadIdx.x; n < N; n += 256) d.db[n] = acc ? d.db[n] + d.dbp[n] : d.dbp[n];
  }
  // publish the partials, take a ticket; the last slice of the strip finishes it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(ticket + (blockIdx.x - slice), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == FG_SPLIT - 1;
    if (s_last) (void)__hip_atomic_exchange(ticket + (blockIdx.x - slice), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  }
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x < 2 * FG_COLS) {
    const int which = threadIdx.x / FG_COLS, col = threadIdx.x % FG_COLS, cc = strip * FG_COLS + col;
    if (cc < K) {
      const float* base = part + (size_t)(blockIdx.x - slice) * 2 * FG_COLS + threadIdx.x;
      float a = 0.f;
#pragma unroll
      for (int sl = 0; sl < FG_SPLIT; ++sl)
        a += __hip_atomic_load(base + (size_t)sl * 2 * FG_COLS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: not from a stale L1 line
      float* o = (which == 0 ? d.dgamma : d.dbeta) + cc;
      *o = acc ? *o + a : a;
    }
  }
}

// out[c] (+)= sum_p part[p * stride + c]; block = 32 columns x 32 partial-lanes (coalesced 128 B rows).  Columns
// >= n1 go to out2[c - n1]: LayerNorm's (dgamma | dbeta) partials are finished by ONE launch.  These launches sit on
// the backward chain between two LayerNorm kernels, so their latency (few workgroups, little data) is what counts.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* part, int P, size_t stride, float* out,
                                                               int n, int accumulate, float* out2, int n1) {
  __shared__ float red[32][33];
  const int cl = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < n) {
    int p = pg;
    for (; p + 96 < P; p += 128) {
      s0 += part[(size_t)p * stride + c];
      s1 += part[(size_t)(p + 32) * stride + c];
      s2 += part[(size_t)(p + 64) * stride + c];
      s3 += part[(size_t)(p + 96) * stride + c];
    }
    for (; p < P; p += 32) s0 += part[(size_t)p * stride + c];
  }
  red[pg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pg == 0 && c < n) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += red[r][cl];
    float* o = c < n1 ? out + c : out2 + (c - n1);
    *o = accumulate ? *o + s : s;
  }
}

// ------------------------------------------------------------------ column sums (bias grads)
// part[blockIdx.y][c] = sum over this block's rows of A[r][c]; block = 32 column-groups x 8 rows
template <bool IN_BF16>
__global__ __launch_bounds__(256) void colsum_kernel(const void* A_, int lda, int M, int N, float* part) {
  __shared__ float red[8][256];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cg) * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < N) {
    for (int r = blockIdx.y * 8 + rl; r < M; r += gridDim.y * 8) {
      if constexpr (IN_BF16) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>((const bf16*)A_ + (size_t)r * lda + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
      } else {
        const float* p = (const float*)A_ + (size_t)r * lda + c;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] += v0[e]; acc[4 + e] += v1[e]; }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = acc[e];
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < N) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][threadIdx.x];
    part[(size_t)blockIdx.y * N + cc] = s;
  }
}

// ------------------------------------------------------------------ periodic row sum
// out[p][:] = sum_r in[(r * P + p)][:]   (fp32, row length D; used for pos/temporal grads)
__global__ void periodic_rowsum_kernel(const float* in, int ld, int R, int P, int D, float* out, int accumulate) {
  const int p = blockIdx.x;
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
    f32x4 s = {0, 0, 0, 0};
    for (int r = 0; r < R; ++r) s += *reinterpret_cast<const f32x4*>(in + (size_t)(r * P + p) * ld + c);
    f32x4* o = reinterpret_cast<f32x4*>(out + (size_t)p * D + c);
    *o = accumulate ? *o + s : s;
  }
}
// out[g][:] = sum_r in[(g * R + r)][:]
// Few groups (the T frames of temporal_embed's gradient, ONE group for cls_token's) over many rows: one thread per four
// columns walking all R rows was a chain of R dependent-latency loads on 1..8 workgroups (94 us at R = 196 on the
// backward chain).  Now RG row lanes per column group, four independent loads in flight per lane, fixed-order LDS sum.
constexpr int GRS_RG = 4, GRS_THREADS = 1024;
__global__ __launch_bounds__(GRS_THREADS) void grouped_rowsum_kernel(const float* in, int ld, int G, int R, int D, float* out, int accumulate) {
  __shared__ f32x4 red[GRS_RG][GRS_THREADS / GRS_RG];
  const int g = blockIdx.x;
  const int cl = threadIdx.x % (GRS_THREADS / GRS_RG), rg = threadIdx.x / (GRS_THREADS / GRS_RG);
  for (int c0 = 0; c0 < D; c0 += GRS_THREADS / GRS_RG * 4) {                 // D = 768: one pass of 256 column lanes (192 used)
    const int c = c0 + cl * 4;
    f32x4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    if (c < D) {
      const float* base = in + (size_t)g * R * ld + c;
      int r = rg;
      for (; r + 3 * GRS_RG < R; r += 4 * GRS_RG) {
        s0 += *reinterpret_cast<const f32x4*>(base + (size_t)r * ld);
        s1 += *reinterpret_cast<const f32x4*>(base + (size_t)(r + GRS_RG) * ld);
        s2 += *reinterpret_cast<const f32x4*>(base + (size_t)(r + 2 * GRS_RG) * ld);
        s3 += *reinterpret_cast<const f32x4*>(base + (size_t)(r + 3 * GRS_RG) * ld);
      }
      for (; r < R; r += GRS_RG) s0 += *reinterpret_cast<const f32x4*>(base + (size_t)r * ld);
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < D) {
      f32x4 s = red[0][cl];
#pragma unroll
      for (int q = 1; q < GRS_RG; ++q) s += red[q][cl];
      f32x4* o = reinterpret_cast<f32x4*>(out + (size_t)g * D + c);
      *o = accumulate ? *o + s : s;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ patch gather (im2col)
// video [BT, C, R, R] (fp32 or bf16) -> A [BT * g * g, C * ps * ps] bf16, k = (c, i, j)
template <bool IN_BF16>
__global__ void im2col_kernel(const void* video, bf16* A, int BT, int C, int R, int ps, int lda) {
  const int g = R / ps;
  const int Kp = C * ps * ps;
  const int k8n = Kp / 8;
  const size_t total = (size_t)BT * g * g * k8n;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(t % k8n) * 8;
    const size_t row = t / k8n;
    const int gx = (int)(row % g), gy = (int)((row / g) % g);
    const size_t bt = row / ((size_t)g * g);
    const int c = k / (ps * ps), i = (k / ps) % ps, j = k % ps;
    const size_t src = ((bt * C + c) * R + (size_t)gy * ps + i) * R + (size_t)gx * ps + j;
    bf16x8 o;
    if constexpr (IN_BF16) {
      o = *reinterpret_cast<const bf16x8*>((const bf16*)video + src);
    } else {
      const float* p = (const float*)video + src;
      const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
      o = bf16x8{f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
    }
    *reinterpret_cast<bf16x8*>(A + row * lda + k) = o;
  }
}

// table[f * N + n][:] = pos[1 + n][:] + temporal[f][:] ; cls0[:] = cls_token[:] + pos[0][:]
__global__ void pos_table_kernel(const float* pos, const float* temporal, const float* cls_token, float* table,
                                 float* cls0, int T, int N, int D) {
  const int r = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    if (r < T * N) table[(size_t)r * D + c] = pos[(size_t)(1 + r % N) * D + c] + temporal[(size_t)(r / N) * D + c];
    else cls0[c] = cls_token[c] + pos[c];
  }
}
// dst[r][:] = src[:]  for r in [0, R)
__global__ void broadcast_rows_kernel(const float* src, float* dst, int ld, int R, int D) {
  const int r = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += blockDim.x) dst[(size_t)r * ld + c] = src[c];
}

// fp32 -> bf16 cast with optional transposed copy: dst[r][c] = src[r][c]; dstT[c][r] = src[r][c]
__global__ void cast_bf16_kernel(const float* src, bf16* dst, bf16* dstT, int R, int C) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) { v = src[(size_t)r * C + c]; if (dst) dst[(size_t)r * C + c] = f2bf(v); }
    tile[i][tx] = v;
  }
  if (!dstT) return;
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < R && c < C) dstT[(size_t)c * R + r] = f2bf(tile[tx][i]);
  }
}

// Every weight shadow of a module in ONE launch.  desc[m] = {src, dst, dstT, R, C, ldd, ldT, first_tile}: the
// CAST_TILE x CAST_TILE tiles of matrix m start at block first_tile; dst / dstT may carry their own leading dimension,
// so the casts can also assemble concatenated shadows (DistilBERT's q|k|v weights) without a separate copy.
// 64x64 tiles, 16-byte loads, 8-byte stores for both copies (the transposed one through a bf16 LDS tile): the first
// version moved one element per thread through 32x32 tiles and ran at 2 TB/s.
constexpr int CAST_TILE = 64;
struct CastDesc { const float* src; bf16* dst; bf16* dstT; long long R, C, ldd, ldT, first_tile; const float* colscale; };
// colscale (or nullptr): column c of src is multiplied by colscale[c] before the cast - the LayerNorm scale folded into the
// weight of the linear layer that follows it (W' = W diag(gamma))
__global__ __launch_bounds__(256) void cast_bf16_multi_kernel(const CastDesc* desc, int n, const int* tile_matrix) {
  __shared__ __attribute__((aligned(8))) bf16 tile[CAST_TILE][CAST_TILE + 4];     // 136-byte pitch: rows stay 8-byte aligned
  int lo = 0, hi = n - 1;                               // last matrix whose first_tile <= blockIdx.x
  if (tile_matrix) {
    lo = tile_matrix[blockIdx.x];                       // caller-built map: a binary search here is ~8 dependent global loads
  } else {
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (desc[mid].first_tile <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
  }
  const CastDesc d = desc[lo];
  const int R = (int)d.R, C = (int)d.C;
  const int t = blockIdx.x - (int)d.first_tile, tpr = (C + CAST_TILE - 1) / CAST_TILE;
  const int c0 = (t % tpr) * CAST_TILE, r0 = (t / tpr) * CAST_TILE;
  const bool vec = C % 4 == 0 && R % 4 == 0 && d.ldd % 4 == 0 && d.ldT % 4 == 0 &&
                   (reinterpret_cast<uintptr_t>(d.src) & 15) == 0 && (reinterpret_cast<uintptr_t>(d.dst) & 7) == 0 &&
                   (reinterpret_cast<uintptr_t>(d.dstT) & 7) == 0 && (reinterpret_cast<uintptr_t>(d.colscale) & 15) == 0;
  const int q = threadIdx.x >> 4, e4 = (threadIdx.x & 15) * 4;       // 16 rows x 16 groups of 4 elements per pass
  if (vec) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = r0 + p * 16 + q, c = c0 + e4;
      bf16x4 o = {f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
      if (r < R && c < C) {
        f32x4 v = *reinterpret_cast<const f32x4*>(d.src + (size_t)r * C + c);
        if (d.colscale) v *= *reinterpret_cast<const f32x4*>(d.colscale + c);
        o = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        if (d.dst) *reinterpret_cast<bf16x4*>(d.dst + (size_t)r * d.ldd + c) = o;
      }
      *reinterpret_cast<bf16x4*>(&tile[p * 16 + q][e4]) = o;
    }
    if (!d.dstT) return;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int cc = p * 16 + q, rr = e4;                 // output row c0 + cc, four consecutive source rows
      if (c0 + cc < C && r0 + rr < R) {
        const bf16x4 o = {tile[rr][cc], tile[rr + 1][cc], tile[rr + 2][cc], tile[rr + 3][cc]};
        *reinterpret_cast<bf16x4*>(d.dstT + (size_t)(c0 + cc) * d.ldT + r0 + rr) = o;
      }
    }
  } else {                                                // odd shapes / alignments: one element at a time
    for (int i = threadIdx.x; i < CAST_TILE * CAST_TILE; i += 256) {
      const int lr = i / CAST_TILE, lc = i % CAST_TILE, r = r0 + lr, c = c0 + lc;
      bf16 o = f2bf(0.f);
      if (r < R && c < C) { o = f2bf(d.src[(size_t)r * C + c] * (d.colscale ? d.colscale[c] : 1.f)); if (d.dst) d.dst[(size_t)r * d.ldd + c] = o; }
      tile[lr][lc] = o;
    }
    if (!d.dstT) return;
    __syncthreads();
    for (int i = threadIdx.x; i < CAST_TILE * CAST_TILE; i += 256) {
      const int lc = i / CAST_TILE, lr = i % CAST_TILE, r = r0 + lr, c = c0 + lc;
      if (r < R && c < C) d.dstT[(size_t)c * d.ldT + r] = tile[lr][lc];
    }
  }
}

}  // namespace oat

using namespace oat;

static int ln_fwd_launch_f8(const float* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, float* mean,
                            float* rstd, int M, int D, float eps, const void* add16, int ldadd, float* sum32, int ldsum,
                            oat::LnF8 f8, void* stream);
// grid cap of the streaming LayerNorm forwards (4 rows per workgroup; x 0.5 / x 2 measured equal, profiles/round4i_knob_sweep.log)
static int ln_fwd_cap() { return 8192; }
static int ln_fwd_launch(const float* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, float* y32,
                         int ldy32, float* mean, float* rstd, int M, int D, float eps, const void* add16, int ldadd,
                         float* sum32, int ldsum, void* stream) {
  if (M <= 0) return 0;
  if (D % 4 || D > LN_MAXV * 256 || ldx % 4 || (y && ldy % 4)) { set_error("layernorm_fwd: D%4==0, D<=1024 required"); return -3; }
  int blocks = (M + 3) / 4; if (blocks > ln_fwd_cap()) blocks = ln_fwd_cap();
  OAT_LAUNCH(ln_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta,
             (bf16*)y, ldy, y32, ldy32, mean, rstd, M, D, eps, (const bf16*)add16, ldadd, sum32, ldsum, (const float*)nullptr, 0,
             oat::LnF8{nullptr, 0, nullptr, nullptr}, (const bf16*)nullptr, 0);
  return check_launch("layernorm_fwd");
}
extern "C" int oat_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, void* y,
                                 int ldy, float* y32, int ldy32, float* mean, float* rstd, int M, int D,
                                 float eps, void* stream) {
  return ln_fwd_launch(x, ldx, gamma, beta, y, ldy, y32, ldy32, mean, rstd, M, D, eps, nullptr, 0, nullptr, 0, stream);
}
// s = x + add16 (bf16) ; sum32 = s (may alias x) ; y = LN(s)
extern "C" int oat_add_layernorm_fwd(const float* x, int ldx, const void* add16, int ldadd, float* sum32, int ldsum,
                                     const float* gamma, const float* beta, void* y, int ldy, float* y32, int ldy32,
                                     float* mean, float* rstd, int M, int D, float eps, void* stream) {
  if (!add16) { set_error("add_layernorm_fwd: add16 is required"); return -4; }
  return ln_fwd_launch(x, ldx, gamma, beta, y, ldy, y32, ldy32, mean, rstd, M, D, eps, add16, ldadd, sum32, ldsum, stream);
}

// s = x + add16 + add16b (both bf16) ; sum32 = s (may alias x) ; y = LN(s).  With folded LayerNorms a block never stores
// y = x + space: the next block's first LayerNorm forms out = x + space + mlp from the block's input and its two bf16
// branch outputs (154 MB written less, 77 MB read more per block).
extern "C" int oat_add2_layernorm_fwd(const float* x, int ldx, const void* add16, int ldadd, const void* add16b, int ldaddb,
                                      float* sum32, int ldsum, const float* gamma, const float* beta, void* y, int ldy,
                                      float* y32, int ldy32, float* mean, float* rstd, int M, int D, float eps, void* stream) {
  if (!add16 || !add16b) { set_error("add2_layernorm_fwd: both addends are required"); return -4; }
  if (M <= 0) return 0;
  if (D % 4 || D > LN_MAXV * 256 || ldx % 4 || ldadd % 4 || ldaddb % 4 || (y && ldy % 4)) { set_error("layernorm_fwd: D%4==0, D<=1024 required"); return -3; }
  int blocks = (M + 3) / 4; if (blocks > ln_fwd_cap()) blocks = ln_fwd_cap();
  OAT_LAUNCH(ln_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, (bf16*)y, ldy, y32, ldy32,
             mean, rstd, M, D, eps, (const bf16*)add16, ldadd, sum32, ldsum, (const float*)nullptr, 0,
             oat::LnF8{nullptr, 0, nullptr, nullptr}, (const bf16*)add16b, ldaddb);
  return check_launch("add2_layernorm_fwd");
}

// LayerNorm on the bf16 residual stream (ln_fwd_r16_kernel): s = x + add_a + add_b ; sum16 = bf16(s) ; y / y32 = LN(s).
// x: bf16, or fp32 when x_is_f32 (block 0: the patch embedding's output); add_a / add_b / sum16 / y / y32 / mean / rstd
// optional; sum16 may alias x (bf16 x only).
extern "C" int oat_layernorm_fwd_r16(const void* x, int x_is_f32, int ldx, const void* add_a, int ldadd_a, const void* add_b,
                                     int ldadd_b, void* sum16, int ldsum, const float* gamma, const float* beta, void* y,
                                     int ldy, float* y32, int ldy32, float* mean, float* rstd, int M, int D, float eps,
                                     void* stream) {
  if (M <= 0) return 0;
  if (!x || (!y && !y32 && !sum16)) { set_error("layernorm_fwd_r16: null pointer"); return -4; }
  if (D % 8 || D > LN_MAXC * 256 || ldx % 8 || ldadd_a % 8 || ldadd_b % 8 || ldsum % 8 || ldy % 8 || ldy32 % 8) {
    set_error("layernorm_fwd_r16: D%8==0, D<=1024, ld%8==0 required"); return -3;
  }
  int blocks = (M + 7) / 8; if (blocks > ln_fwd_cap()) blocks = ln_fwd_cap();
  if (x_is_f32)
    OAT_LAUNCH(ln_fwd_r16_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, (const bf16*)add_a, ldadd_a,
               (const bf16*)add_b, ldadd_b, (bf16*)sum16, ldsum, gamma, beta, (bf16*)y, ldy, y32, ldy32, mean, rstd, M, D, eps, oat::LnF8{nullptr, 0, nullptr, nullptr});
  else
    OAT_LAUNCH(ln_fwd_r16_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, (const bf16*)add_a, ldadd_a,
               (const bf16*)add_b, ldadd_b, (bf16*)sum16, ldsum, gamma, beta, (bf16*)y, ldy, y32, ldy32, mean, rstd, M, D, eps, oat::LnF8{nullptr, 0, nullptr, nullptr});
  return check_launch("layernorm_fwd_r16");
}

// oat_layernorm_fwd_r16 with the e4m3 copy of y for an fp8 GEMM (y8 = e4m3(y * *qscale), max |y| -> *amax): the fp8 forward on the
// bf16 residual stream.  y (bf16) is still written - backward and the weight gradient read it.
extern "C" int oat_layernorm_fwd_r16_f8(const void* x, int x_is_f32, int ldx, const void* add_a, int ldadd_a, const void* add_b,
                                        int ldadd_b, void* sum16, int ldsum, const float* gamma, const float* beta, void* y,
                                        int ldy, void* y8, int ld8, const float* qscale, float* amax, float* mean, float* rstd,
                                        int M, int D, float eps, void* stream) {
  if (M <= 0) return 0;
  if (!x || !y8 || !qscale || !amax) { set_error("layernorm_fwd_r16_f8: null pointer"); return -4; }
  if (D % 8 || D > LN_MAXC * 256 || ldx % 8 || ldadd_a % 8 || ldadd_b % 8 || ldsum % 8 || ldy % 8 || ld8 % 8) {
    set_error("layernorm_fwd_r16_f8: D%8==0, D<=1024, ld%8==0 required"); return -3;
  }
  int blocks = (M + 7) / 8; if (blocks > ln_fwd_cap()) blocks = ln_fwd_cap();
  const oat::LnF8 f8{(uint8_t*)y8, ld8, qscale, amax};
  if (x_is_f32)
    OAT_LAUNCH((ln_fwd_r16_kernel<true, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, (const bf16*)add_a, ldadd_a,
               (const bf16*)add_b, ldadd_b, (bf16*)sum16, ldsum, gamma, beta, (bf16*)y, ldy, (float*)nullptr, 0, mean, rstd, M, D, eps, f8);
  else
    OAT_LAUNCH((ln_fwd_r16_kernel<false, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, (const bf16*)add_a, ldadd_a,
               (const bf16*)add_b, ldadd_b, (bf16*)sum16, ldsum, gamma, beta, (bf16*)y, ldy, (float*)nullptr, 0, mean, rstd, M, D, eps, f8);
  return check_launch("layernorm_fwd_r16_f8");
}

// LayerNorm with the fp8 copy of its output for an fp8 GEMM: y (bf16, kept for backward) and y8 = e4m3(y * *qscale),
// amax of y recorded.  add16 optional (then sum32 = x + add16 as in oat_add_layernorm_fwd).
static int ln_fwd_launch_f8(const float* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, float* mean,
                            float* rstd, int M, int D, float eps, const void* add16, int ldadd, float* sum32, int ldsum,
                            oat::LnF8 f8, void* stream) {
  if (M <= 0) return 0;
  if (D % 4 || D > LN_MAXV * 256 || ldx % 4 || (y && ldy % 4) || f8.ld8 % 4) { set_error("layernorm_fwd_f8: D%4==0, D<=1024 required"); return -3; }
  if (!f8.y8 || !f8.qscale || !f8.amax) { set_error("layernorm_fwd_f8: null pointer"); return -4; }
  int blocks = (M + 3) / 4; if (blocks > 4096) blocks = 4096;
  OAT_LAUNCH(ln_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, (bf16*)y, ldy,
                     (float*)nullptr, 0, mean, rstd, M, D, eps, (const bf16*)add16, ldadd, sum32, ldsum, (const float*)nullptr, 0, f8,
                     (const bf16*)nullptr, 0);
  return check_launch("layernorm_fwd_f8");
}
extern "C" int oat_layernorm_fwd_f8(const float* x, int ldx, const void* add16_or_null, int ldadd, float* sum32, int ldsum,
                                    const float* gamma, const float* beta, void* y, int ldy, void* y8, int ld8,
                                    const float* qscale, float* amax, float* mean, float* rstd, int M, int D, float eps,
                                    void* stream) {
  return ln_fwd_launch_f8(x, ldx, gamma, beta, y, ldy, mean, rstd, M, D, eps, add16_or_null, ldadd, sum32, ldsum,
                          oat::LnF8{(uint8_t*)y8, ld8, qscale, amax}, stream);
}

// s = x + add32 (fp32) ; sum32 = s (may alias x) ; y / y32 = LN(s)
extern "C" int oat_add32_layernorm_fwd(const float* x, int ldx, const float* add32, int ldadd, float* sum32, int ldsum,
                                       const float* gamma, const float* beta, void* y, int ldy, float* y32, int ldy32,
                                       float* mean, float* rstd, int M, int D, float eps, void* stream) {
  if (!add32) { set_error("add32_layernorm_fwd: add32 is required"); return -4; }
  if (M <= 0) return 0;
  if (D % 4 || D > LN_MAXV * 256 || ldx % 4 || ldadd % 4 || (y && ldy % 4)) { set_error("layernorm_fwd: D%4==0, D<=1024 required"); return -3; }
  const int blocks = (M + 3) / 4;
  OAT_LAUNCH(ln_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, (bf16*)y, ldy, y32,
                     ldy32, mean, rstd, M, D, eps, (const bf16*)nullptr, 0, sum32, ldsum, add32, ldadd, oat::LnF8{nullptr, 0, nullptr, nullptr},
                     (const bf16*)nullptr, 0);
  return check_launch("add32_layernorm_fwd");
}

static int ln_bwd_cap() { return 1024; }
extern "C" int oat_ln_bwd_blocks(int M) { int b = (M + 3) / 4; const int cap = ln_bwd_cap(); return b > cap ? cap : b; }

// part: fp32 workspace of oat_ln_bwd_blocks(M) * 2 * D floats (or NULL to skip dgamma/dbeta)
static int ln_bwd_launch(const void* dy, int dy_is_bf16, int lddy, const void* x, int x_is_bf16, int ldx,
                         const float* mean, const float* rstd, const float* gamma, const float* dres,
                         int lddres, float* dx, int lddx, void* dx16, int lddx16, int dx16_excl_res,
                         float* dgamma, float* dbeta, int accumulate, float* part, int M, int D, oat::LnF8 f8,
                         const void* dres16, int lddres16, void* stream);
extern "C" int oat_layernorm_bwd(const void* dy, int dy_is_bf16, int lddy, const float* x, int ldx,
                                 const float* mean, const float* rstd, const float* gamma, const float* dres,
                                 int lddres, float* dx, int lddx, void* dx16, int lddx16, int dx16_excl_res,
                                 float* dgamma, float* dbeta, int accumulate, float* part, int M, int D,
                                 void* stream) {
  return ln_bwd_launch(dy, dy_is_bf16, lddy, x, 0, ldx, mean, rstd, gamma, dres, lddres, dx, lddx, dx16, lddx16, dx16_excl_res,
                       dgamma, dbeta, accumulate, part, M, D, oat::LnF8{nullptr, 0, nullptr, nullptr}, nullptr, 0, stream);
}
// LayerNorm backward on the bf16 residual stream: x (the layer's forward input) is bf16, the residual-gradient addend
// dres16 is bf16 (may be dx16 itself: in place), outputs dx (fp32, optional) and dx16 (bf16) = result + dres16.
extern "C" int oat_layernorm_bwd_r16(const void* dy, int dy_is_bf16, int lddy, const void* x_bf16, int ldx,
                                     const float* mean, const float* rstd, const float* gamma, const void* dres16,
                                     int lddres16, float* dx, int lddx, void* dx16, int lddx16, float* dgamma,
                                     float* dbeta, int accumulate, float* part, int M, int D, void* stream) {
  if (ldx % 4 || lddres16 % 4 || lddx16 % 4) { oat::set_error("layernorm_bwd_r16: ld%4==0 required"); return -3; }
  return ln_bwd_launch(dy, dy_is_bf16, lddy, x_bf16, 1, ldx, mean, rstd, gamma, nullptr, 0, dx, lddx, dx16, lddx16, 0,
                       dgamma, dbeta, accumulate, part, M, D, oat::LnF8{nullptr, 0, nullptr, nullptr}, dres16, lddres16, stream);
}
static int ln_bwd_launch(const void* dy, int dy_is_bf16, int lddy, const void* x, int x_is_bf16, int ldx,
                         const float* mean, const float* rstd, const float* gamma, const float* dres,
                         int lddres, float* dx, int lddx, void* dx16, int lddx16, int dx16_excl_res,
                         float* dgamma, float* dbeta, int accumulate, float* part, int M, int D, oat::LnF8 f8,
                         const void* dres16, int lddres16, void* stream) {
  if (M <= 0) return 0;
  if (D % 4 || D > LN_MAXV * 256) { set_error("layernorm_bwd: D%4==0, D<=1024 required"); return -3; }
  if ((dgamma || dbeta) && !part) { set_error("layernorm_bwd: dgamma/dbeta need the partial workspace"); return -4; }
  const int blocks = oat_ln_bwd_blocks(M);
  hipStream_t s = (hipStream_t)stream;
#define OAT_LN_BWD(DYB, XB) OAT_LAUNCH(ln_bwd_kernel<DYB, XB>, dim3(blocks), dim3(256), 0, s, dy, lddy, x, ldx, mean, rstd, gamma, dres, \
                                      lddres, dx, lddx, (bf16*)dx16, lddx16, dx16_excl_res, part, M, D, f8, (const bf16*)dres16, lddres16)
  if (dy_is_bf16) { if (x_is_bf16) OAT_LN_BWD(true, true); else OAT_LN_BWD(true, false); }
  else { if (x_is_bf16) OAT_LN_BWD(false, true); else OAT_LN_BWD(false, false); }
#undef OAT_LN_BWD
  int rc = check_launch("layernorm_bwd");
  if (rc || !part) return rc;
  if (dgamma && dbeta)
    OAT_LAUNCH(reduce_partials_kernel, dim3((2 * D + 31) / 32), dim3(1024), 0, s, part, blocks, (size_t)2 * D, dgamma,
                       2 * D, accumulate, dbeta, D);
  else if (dgamma)
    OAT_LAUNCH(reduce_partials_kernel, dim3((D + 31) / 32), dim3(1024), 0, s, part, blocks, (size_t)2 * D, dgamma, D,
                       accumulate, (float*)nullptr, D);
  else if (dbeta)
    OAT_LAUNCH(reduce_partials_kernel, dim3((D + 31) / 32), dim3(1024), 0, s, part + D, blocks, (size_t)2 * D, dbeta, D,
                       accumulate, (float*)nullptr, D);
  return check_launch("layernorm_bwd_finish");
}

// LayerNorm backward, folded form (ln_bwd_xhat_kernel): dxh = bf16 gradient w.r.t. the normalised row (what the data-
// gradient GEMM of the folded weights delivers), xhat = the saved bf16 normalised row, rstd fp32 per row.
// dx (fp32, optional) = result (+ dres); dx16 (bf16, optional) = the same, or without dres when dx16_excl_res.
// add_a / add_b (bf16, optional): further addends of dx / dx16 (earlier LayerNorms' increments of the residual gradient);
// dxp16 (bf16, optional): the plain result, before any addend.
extern "C" int oat_layernorm_bwd_xhat(const void* dxh, int lddxh, const void* xhat, int ldxh, const float* rstd,
                                      const float* dres, int lddres, float* dx, int lddx, void* dx16, int lddx16,
                                      int dx16_excl_res, const void* add_a, int ldadd_a, const void* add_b, int ldadd_b,
                                      void* dxp16, int lddxp, int M, int D, void* stream) {
  if (M <= 0) return 0;
  if (D % 8 || D > LN_MAXC * 256 || lddxh % 8 || ldxh % 8 || ldadd_a % 8 || ldadd_b % 8 || lddxp % 8 || lddres % 8 || lddx % 8 || lddx16 % 8) {
    set_error("layernorm_bwd_xhat: D%8==0, D<=1024, ld%8==0 required"); return -3;
  }
  if (!dxh || !xhat || !rstd || (!dx && !dx16 && !dxp16)) { set_error("layernorm_bwd_xhat: null pointer"); return -4; }
  // no partial-sum rows hang on the grid here (ln_bwd_kernel's cap of 1024 blocks keeps its (dgamma, dbeta) partials small):
  // 4096 blocks measure 90 / 40 / 128 us for the three forms of a block at M = 50208 against 102 / 48 / 151 at 1024; in the
  // step 1024 -> 4096 -> 8192: 48.42 -> 47.97 -> 47.88 ms
  constexpr int cap = 8192;
  int blocks = (M + 7) / 8; if (blocks > cap) blocks = cap;
  OAT_LAUNCH(ln_bwd_xhat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)dxh, lddxh,
             (const bf16*)xhat, ldxh, rstd, dres, lddres, dx, lddx, (bf16*)dx16, lddx16, dx16_excl_res, M, D,
             (const bf16*)add_a, ldadd_a, (const bf16*)add_b, ldadd_b, (bf16*)dxp16, lddxp);
  return check_launch("layernorm_bwd_xhat");
}
// desc: device array of {W, beta, b, out, N, K, first_row, 0} (8 x 8 bytes); block_desc[i] = descriptor of block i (4 rows
// per block; first_row = 4 x the descriptor's first block)
extern "C" int oat_fold_bias_multi(const void* desc, const int* block_desc, int total_blocks, void* stream) {
  if (total_blocks <= 0) return 0;
  if (!desc || !block_desc) { set_error("fold_bias_multi: null pointer"); return -4; }
  OAT_LAUNCH(fold_bias_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const FoldBiasDesc*)desc, block_desc);
  return check_launch("fold_bias_multi");
}
// desc: device array of {dWp, dbp, W, gamma, beta, dW, db, dgamma, dbeta, N, K, first_block, accumulate} (13 x 8 bytes);
// a layer owns oat_ln_fold_blocks(K) consecutive blocks from first_block.  work: caller-owned device memory,
// total_blocks * 128 floats of partial sums followed by total_blocks ints of ticket counters that must be ZERO at the first
// launch (every launch leaves them zero); K % 4 == 0.
extern "C" int oat_ln_fold_blocks(int K) { return ((K + FG_COLS - 1) / FG_COLS) * FG_SPLIT; }
extern "C" int oat_ln_fold_grads(const void* desc, int n_desc, int total_blocks, void* work, void* stream) {
  if (n_desc <= 0 || total_blocks <= 0) return 0;
  if (!desc || !work) { set_error("ln_fold_grads: null pointer"); return -4; }
  float* part = static_cast<float*>(work);
  int* ticket = reinterpret_cast<int*>(part + (size_t)total_blocks * 2 * FG_COLS);
  OAT_LAUNCH(ln_fold_grads_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const FoldGradDesc*)desc, n_desc,
             part, ticket);
  return check_launch("ln_fold_grads");
}

extern "C" int oat_colsum_rows(int M) { int r = (M + 7) / 8; return r > 256 ? 256 : (r < 1 ? 1 : r); }

// out[N] (+)= column sums of A[M,N]; part = workspace of oat_colsum_rows(M) * N floats
extern "C" int oat_colsum(const void* A, int is_bf16, int lda, int M, int N, float* out, int accumulate,
                          float* part, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (N % 8 || lda % 8) { set_error("colsum: N%8, lda%8 required"); return -3; }
  const int rows = oat_colsum_rows(M);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((N + 255) / 256, rows);
  if (is_bf16) OAT_LAUNCH(colsum_kernel<true>, grid, dim3(256), 0, s, A, lda, M, N, part);
  else OAT_LAUNCH(colsum_kernel<false>, grid, dim3(256), 0, s, A, lda, M, N, part);
  OAT_LAUNCH(reduce_partials_kernel, dim3((N + 31) / 32), dim3(1024), 0, s, part, rows, (size_t)N, out, N,
                     accumulate, (float*)nullptr, N);
  return check_launch("colsum");
}

extern "C" int oat_periodic_rowsum(const float* in, int ld, int R, int P, int D, float* out, int accumulate, void* stream) {
  if (D % 4 || ld % 4) { set_error("periodic_rowsum: D%4 required"); return -3; }
  if (P <= 0) return 0;
  OAT_LAUNCH(periodic_rowsum_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, in, ld, R, P, D, out, accumulate);
  return check_launch("periodic_rowsum");
}
extern "C" int oat_grouped_rowsum(const float* in, int ld, int G, int R, int D, float* out, int accumulate, void* stream) {
  if (D % 4 || ld % 4) { set_error("grouped_rowsum: D%4 required"); return -3; }
  if (G <= 0) return 0;
  OAT_LAUNCH(grouped_rowsum_kernel, dim3(G), dim3(GRS_THREADS), 0, (hipStream_t)stream, in, ld, G, R, D, out, accumulate);
  return check_launch("grouped_rowsum");
}

extern "C" int oat_im2col(const void* video, int is_bf16, void* A, int BT, int C, int R, int ps, int lda, void* stream) {
  if (ps % 8 || R % ps || lda % 8) { set_error("im2col: patch size must be a multiple of 8 and divide R"); return -3; }
  const size_t total = (size_t)BT * (R / ps) * (R / ps) * (C * ps * ps / 8);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipStream_t s = (hipStream_t)stream;
  if (is_bf16) OAT_LAUNCH(im2col_kernel<true>, dim3(blocks), dim3(256), 0, s, video, (bf16*)A, BT, C, R, ps, lda);
  else OAT_LAUNCH(im2col_kernel<false>, dim3(blocks), dim3(256), 0, s, video, (bf16*)A, BT, C, R, ps, lda);
  return check_launch("im2col");
}

extern "C" int oat_pos_table(const float* pos, const float* temporal, const float* cls_token, float* table,
                             float* cls0, int T, int N, int D, void* stream) {
  OAT_LAUNCH(pos_table_kernel, dim3(T * N + 1), dim3(256), 0, (hipStream_t)stream, pos, temporal, cls_token,
                     table, cls0, T, N, D);
  return check_launch("pos_table");
}
extern "C" int oat_broadcast_rows(const float* src, float* dst, int ld, int R, int D, void* stream) {
  if (R <= 0) return 0;
  OAT_LAUNCH(broadcast_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, src, dst, ld, R, D);
  return check_launch("broadcast_rows");
}
extern "C" int oat_cast_bf16_tile(void) { return CAST_TILE; }
extern "C" int oat_cast_bf16_multi(const void* desc, int n_matrices, int total_tiles, const int* tile_matrix, void* stream) {
  if (n_matrices <= 0 || total_tiles <= 0) return 0;
  if (!desc) { set_error("cast_bf16_multi: null descriptor table"); return -4; }
  OAT_LAUNCH(cast_bf16_multi_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream,
                     (const CastDesc*)desc, n_matrices, tile_matrix);
  return check_launch("cast_bf16_multi");
}
extern "C" int oat_cast_bf16(const float* src, void* dst, void* dstT, int R, int C, void* stream) {
  if (R <= 0 || C <= 0) return 0;
  OAT_LAUNCH(cast_bf16_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16*)dst, (bf16*)dstT, R, C);
  return check_launch("cast_bf16");
}
