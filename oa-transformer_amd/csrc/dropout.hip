// Dropout for the text tower's training mode (HF DistilBERT: after the embedding LayerNorm, on the attention
// probabilities, after ffn.lin2 - transformers/models/distilbert/modeling_distilbert.py, third party; call site
// /root/reference/OATrans/model/oa_model.py:113-121 with text_model.train() at :56).  Masks: rng.h.
#include "rng.h"

namespace oat {

__global__ void rng_tick_kernel(unsigned long long* rng) { rng[1] += 1ull; }

// out[m][c] = x[m][c] * mask(m * D + c) (+ resid[m][c]) ; out16 = bf16 of the same.  4 columns per thread = one Philox call.
__global__ void dropout_kernel(const float* x, int ldx, const float* resid, int ldr, float* out32, int ldo, bf16* out16,
                               int ld16, int M, int D, DropSite d) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;       // quad index over [M, D / 4]
  const int qpr = D >> 2;
  if (q >= M * qpr) return;
  const int m = q / qpr, c = (q - m * qpr) << 2;
  const u32x4 r = drop_draw4(d, ((unsigned long long)m * D + c) >> 2);
  f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)m * ldx + c);
  v[0] *= r.x < d.thresh ? 0.f : d.keep_scale;
  v[1] *= r.y < d.thresh ? 0.f : d.keep_scale;
  v[2] *= r.z < d.thresh ? 0.f : d.keep_scale;
  v[3] *= r.w < d.thresh ? 0.f : d.keep_scale;
  if (resid) v += *reinterpret_cast<const f32x4*>(resid + (size_t)m * ldr + c);
  if (out32) *reinterpret_cast<f32x4*>(out32 + (size_t)m * ldo + c) = v;
  if (out16) {
    bf16* o = out16 + (size_t)m * ld16 + c;
    o[0] = f2bf(v[0]); o[1] = f2bf(v[1]); o[2] = f2bf(v[2]); o[3] = f2bf(v[3]);
  }
}

__global__ void dropout_mask_kernel(float* out, long long n, DropSite d) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = drop_mult(d, (unsigned long long)i);
}

__global__ void philox_kat_kernel(const uint32_t* in, uint32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32x4 r = philox4x32_10(u32x4{in[6 * i], in[6 * i + 1], in[6 * i + 2], in[6 * i + 3]}, in[6 * i + 4], in[6 * i + 5]);
  out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

}  // namespace oat

using namespace oat;

extern "C" int oat_rng_tick(void* rng, void* stream) {
  if (!rng) { set_error("rng_tick: null state"); return -4; }
  OAT_LAUNCH(rng_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)rng);
  return check_launch("rng_tick");
}

extern "C" int oat_dropout(const float* x, int ldx, const float* resid, int ldr, float* out32, int ldo, void* out16, int ld16,
                           int M, int D, float p, const void* rng, unsigned site, void* stream) {
  if (M <= 0) return 0;
  if (!x || !rng || (!out32 && !out16)) { set_error("dropout: null pointer"); return -4; }
  if (D % 4 || ldx % 4 || (resid && ldr % 4) || (out32 && ldo % 4)) { set_error("dropout: D and row strides must be multiples of 4"); return -3; }
  if (!(p >= 0.f && p < 1.f)) { set_error("dropout: p must be in [0, 1)"); return -3; }
  const int quads = M * (D / 4);
  OAT_LAUNCH(dropout_kernel, dim3((quads + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, resid, ldr, out32, ldo,
                     (bf16*)out16, ld16, M, D, make_drop_site(rng, site, p));
  return check_launch("dropout");
}

// The multipliers (0 or 1/(1-p)) of elements [0, n) of a site, for tests and for replaying a step in the CPU oracle.
extern "C" int oat_dropout_mask(float* out, long long n, float p, const void* rng, unsigned site, void* stream) {
  if (n <= 0) return 0;
  if (!out || !rng) { set_error("dropout_mask: null pointer"); return -4; }
  OAT_LAUNCH(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n,
                     make_drop_site(rng, site, p));
  return check_launch("dropout_mask");
}

// Known-answer access to the generator: in = n x (4 counter words, 2 key words), out = n x 4 words.
extern "C" int oat_philox4x32_10(const void* in, void* out, int n, void* stream) {
  if (n <= 0) return 0;
  OAT_LAUNCH(philox_kat_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const uint32_t*)in, (uint32_t*)out, n);
  return check_launch("philox4x32_10");
}
