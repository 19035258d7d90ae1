// OCP fp8 (e4m3fn) path of BASELINE.json config 5 ("16-frame 336^2 fp8 MFMA"): per-tensor scaled quantisation kernels
// and the C entry of the fp8 GEMM (gemm_nt_pp.hip, PPF_F8).  Replaces, for the forward linears, the bf16 GEMMs behind
// the nn.Linear calls at /root/reference/OATrans/model/video_transformer.py:46-50,102,133 (the reference has no fp8
// code of its own; the scaling recipe is the usual per-tensor "delayed scaling": a tensor is quantised with the scale
// derived from the amax it had one step earlier while its current amax is recorded for the next step).
//
// A quantisation SITE i owns three device floats: amax[i] (running max |x| of the current step, written with atomicMax
// on the bit pattern - non-negative floats order like unsigned integers), qscale[i] (x -> fp8: q = sat(x * qscale)) and
// dq[i] = 1 / qscale[i] (what the GEMM epilogue multiplies by).  oat_fp8_update_scales turns amax into the next
// qscale / dq and clears amax.  e4m3fn has no infinities and v_cvt_pk_fp8_f32 returns NaN above 448 (probe:
// scripts/dev/fp8_probe), so values are clamped to +-448 before conversion.
#include "gemm.h"
#include "fp8.h"

namespace oat {

// 8 elements per thread: x [M, K] (bf16 | f32, row stride ldx) -> out8 [M, K] (row stride ld8 bytes), amax of |x|
template <bool BF16IN, bool QUANT, bool E5M2 = false>
__global__ __launch_bounds__(256) void fp8_quant_kernel(const void* x, int ldx, uint8_t* out8, int ld8, int M, int K,
                                                        const float* qscale, float* amax) {
  const int kq = K >> 3;
  const long long total = (long long)M * kq;
  const float qs = QUANT ? qscale[0] : 0.f;
  float m = 0.f;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int r = (int)(q / kq), c = (int)(q - (long long)r * kq) << 3;
    float v[8];
    if constexpr (BF16IN) {
      const bf16x8 t = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(x) + (size_t)r * ldx + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = bf2f(t[e]);
    } else {
      const float* p = reinterpret_cast<const float*>(x) + (size_t)r * ldx + c;
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(p), t1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = t0[e]; v[4 + e] = t1[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
    if constexpr (QUANT) {
      uint2 o;
      o.x = pack_f8x4<E5M2>(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs);
      o.y = pack_f8x4<E5M2>(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs);
      *reinterpret_cast<uint2*>(out8 + (size_t)r * ld8 + c) = o;
    }
  }
  if (amax) amax_commit(m, amax);
}

// many matrices in one launch (the weight shadows): block b works on matrix owner[b], chunk b - first_block
struct F8Desc { const bf16* src; uint8_t* dst; long long n, site, first_block; };       // contiguous, n % 8 == 0
constexpr int F8_CHUNK = 256 * 8 * 4;                                                    // elements per block
template <bool QUANT>
__global__ __launch_bounds__(256) void fp8_multi_kernel(const F8Desc* desc, const int* owner, const float* qscale, float* amax) {
  const F8Desc d = desc[owner[blockIdx.x]];
  const long long base = ((long long)blockIdx.x - d.first_block) * F8_CHUNK;
  const float qs = QUANT ? qscale[d.site] : 0.f;
  float m = 0.f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const long long i = base + (long long)(p * 256 + threadIdx.x) * 8;
    if (i < d.n) {
      const bf16x8 t = *reinterpret_cast<const bf16x8*>(d.src + i);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[e] = bf2f(t[e]); m = fmaxf(m, fabsf(v[e])); }
      if constexpr (QUANT) {
        uint2 o;
        o.x = pack_fp8x4(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs);
        o.y = pack_fp8x4(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs);
        *reinterpret_cast<uint2*>(d.dst + i) = o;
      }
    }
  }
  if constexpr (!QUANT) amax_commit(m, amax + d.site);
}

// amax -> qscale = 448 / (margin * amax), dq = 1 / qscale; amax cleared.  Sites that saw no data keep their scales.
__global__ void fp8_update_scales_kernel(float* amax, float* qscale, float* dq, int n, float margin, float fmax) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = amax[i];
  if (a > 0.f && a < INFINITY) {
    const float q = fmax / (a * margin);
    qscale[i] = q;
    dq[i] = 1.f / q;
  }
  amax[i] = 0.f;
}

}  // namespace oat

using namespace oat;

extern "C" int oat_fp8_quant(const void* x, int is_bf16, int ldx, void* out8, int ld8, int M, int K, const float* qscale,
                             float* amax, void* stream) {
  if (M <= 0 || K <= 0) return 0;
  if (!x || !out8 || !qscale) { set_error("fp8_quant: null pointer"); return -4; }
  if (K % 8 || ldx % 8 || ld8 % 8) { set_error("fp8_quant: K, ldx, ld8 must be multiples of 8"); return -3; }
  const long long quads = (long long)M * (K / 8);
  const int grid = (int)((quads + 255) / 256 < 2048 ? (quads + 255) / 256 : 2048);
  hipStream_t s = (hipStream_t)stream;
  uint8_t* o = (uint8_t*)out8;
  if (is_bf16) OAT_LAUNCH((fp8_quant_kernel<true, true, false>), dim3(grid), dim3(256), 0, s, x, ldx, o, ld8, M, K, qscale, amax);
  else OAT_LAUNCH((fp8_quant_kernel<false, true, false>), dim3(grid), dim3(256), 0, s, x, ldx, o, ld8, M, K, qscale, amax);
  return check_launch("fp8_quant");
}
extern "C" int oat_fp8_amax(const void* x, int is_bf16, int ldx, int M, int K, float* amax, void* stream) {
  if (M <= 0 || K <= 0) return 0;
  if (!x || !amax) { set_error("fp8_amax: null pointer"); return -4; }
  if (K % 8 || ldx % 8) { set_error("fp8_amax: K, ldx must be multiples of 8"); return -3; }
  const long long quads = (long long)M * (K / 8);
  const int grid = (int)((quads + 255) / 256 < 2048 ? (quads + 255) / 256 : 2048);
  if (is_bf16) OAT_LAUNCH((fp8_quant_kernel<true, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, nullptr, 0, M, K, nullptr, amax);
  else OAT_LAUNCH((fp8_quant_kernel<false, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, nullptr, 0, M, K, nullptr, amax);
  return check_launch("fp8_amax");
}
extern "C" int oat_fp8_chunk_elems(void) { return F8_CHUNK; }
// desc: int64 rows {src bf16 ptr, dst fp8 ptr, n elements, site, first block}; owner: block -> row.  quant = 0: amax only
extern "C" int oat_fp8_multi(const void* desc, const int* owner, int total_blocks, const float* qscale, float* amax, int quant,
                             void* stream) {
  if (total_blocks <= 0) return 0;
  if (!desc || !owner || !amax || (quant && !qscale)) { set_error("fp8_multi: null pointer"); return -4; }
  if (quant) OAT_LAUNCH((fp8_multi_kernel<true>), dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const F8Desc*)desc, owner, qscale, amax);
  else OAT_LAUNCH((fp8_multi_kernel<false>), dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const F8Desc*)desc, owner, qscale, amax);
  return check_launch("fp8_multi");
}
extern "C" int oat_fp8_update_scales(float* amax, float* qscale, float* dq, int n, float margin, void* stream) {
  if (n <= 0) return 0;
  if (!amax || !qscale || !dq) { set_error("fp8_update_scales: null pointer"); return -4; }
  if (!(margin >= 1.f)) { set_error("fp8_update_scales: margin must be >= 1"); return -3; }
  OAT_LAUNCH(fp8_update_scales_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, amax, qscale, dq, n, margin,
             F8_MAX);
  return check_launch("fp8_update_scales");
}

// C[M, N] = dq_a dq_b (A8[M, K] . B8[N, K]^T) + bias with the bf16-output epilogues of oat_gemm_nt (EPI_BF16 = 0,
// EPI_GELU_GRAD = 5).  A8 / B8: OCP e4m3 bytes, lda / ldb in elements.  K % 256 == 0, N % 256 == 0, N <= 4096, M >= 256.
// epi: EPI_BF16 or EPI_GELU_GRAD (out = gelu'(h), out2 = gelu(h)).  out8 (optional, EPI_GELU_GRAD): e4m3 copy of out2 for the NEXT
// GEMM, quantised with *q_out, its amax recorded in *amax_out.  (The e5m2-operand form and the EPI_MUL_AUX epilogue of the fp8
// data-gradient mode of rounds 2-5 left the library in round 6 with that mode.)
extern "C" int oat_gemm_nt_f8(const void* A8, const void* B8, int M, int N, int K, int lda, int ldb, int epi, void* out, int ldc,
                              void* out2, int ld2, const float* bias, const float* dq_a, const float* dq_b,
                              void* out8, int ld8, const float* q_out, float* amax_out, void* stream) {
  const int h_u8 = (epi >> 8) & 1;          // epi | 0x100: 8-bit GELU derivative (as in oat_gemm_nt)
  epi &= 0xff;
  if (epi != EPI_BF16 && epi != EPI_GELU_GRAD) { set_error("gemm_nt_f8: EPI_BF16 or EPI_GELU_GRAD"); return -3; }
  if (out8 && (epi != EPI_GELU_GRAD || !q_out || !amax_out || ld8 % 4)) { set_error("gemm_nt_f8: out8 needs EPI_GELU_GRAD, q_out, amax_out, ld8 % 4 == 0"); return -4; }
  if (M <= 0 || N <= 0 || K <= 0) { set_error("gemm_nt_f8: empty problem"); return -1; }
  if (!A8 || !B8 || !out || !dq_a || !dq_b) { set_error("gemm_nt_f8: null pointer"); return -4; }
  if (epi == EPI_GELU_GRAD && !out2) { set_error("gemm_nt_f8: EPI_GELU_GRAD needs out2"); return -4; }
  if (ldc % 8 != 0) { set_error("gemm_nt_f8: ldc must be a multiple of 8"); return -3; }
  if (h_u8 && epi == EPI_GELU_GRAD && ldc != N) {
    set_error("gemm_nt_f8: the 8-bit GELU derivative is a dense blocked tensor (ld == N)"); return -3;
  }
  GemmArgs g{(const bf16*)A8, (const bf16*)B8, M, N, K, lda, ldb, out, ldc, out2, ld2, bias, nullptr, 0, 0, nullptr, 0, 0, 0,
             nullptr, nullptr, dq_a, dq_b, out8, ld8, q_out, amax_out, h_u8};
  return launch_pp_f8(epi, g, 256, (hipStream_t)stream);
}
