// Object-aware extras: tiny batched contractions and element-wise losses.
//   mask-pool   einsum('b o l, b l c -> b o c')   /root/reference/OATrans/model/oa_model_global_local.py:178,200
//   region-sim  sigmoid(einsum('b k f, b n f -> b k n'))   /root/reference/OATrans/model/oa_model_region_mem.py:147-151
//   BCELoss(reduction='sum')                      /root/reference/OATrans/trainer/trainer_region_mem.py:97,166
//   patch-mean pooling of the GL / region tails   oa_video_transformer_global_local.py:356, oa_model_region_mem.py:117
// All of them are < 0.02 % of the step's FLOPs: one generic fp32 strided batched matmul (any transposition
// is a stride choice, so forward and both backward products share it) plus element-wise kernels.
#include "common.h"

namespace oat {

struct BmmArgs {
  const float* A; const float* Bm; float* C;
  int nb, I, J, K;
  long long sAb, sAi, sAk, sBb, sBk, sBj, sCb, sCi, sCj;
  int sigmoid, accumulate;
};

// C[b,i,j] (+)= act( sum_k A[b,i,k] * Bm[b,k,j] ) ; 16x16 outputs per block, K tiled by 16 through LDS
__global__ __launch_bounds__(256) void bmm_strided_kernel(BmmArgs a) {
  __shared__ float sa[16][17], sb[16][17];
  const int b = blockIdx.z;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
  const float* A = a.A + (long long)b * a.sAb;
  const float* Bm = a.Bm + (long long)b * a.sBb;
  float acc = 0.f;
  for (int k0 = 0; k0 < a.K; k0 += 16) {
    const int ka = k0 + tx, kb = k0 + ty;
    sa[ty][tx] = (i < a.I && ka < a.K) ? A[(long long)i * a.sAi + (long long)ka * a.sAk] : 0.f;
    sb[ty][tx] = (kb < a.K && j < a.J) ? Bm[(long long)kb * a.sBk + (long long)j * a.sBj] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += sa[ty][k] * sb[k][tx];
    __syncthreads();
  }
  if (i < a.I && j < a.J) {
    float* c = a.C + (long long)b * a.sCb + (long long)i * a.sCi + (long long)j * a.sCj;
    if (a.sigmoid) acc = 1.f / (1.f + __expf(-acc));
    *c = a.accumulate ? *c + acc : acc;
  }
}

// dz = ds * s * (1 - s)
__global__ void sigmoid_bwd_kernel(const float* s, const float* ds, float* dz, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dz[i] = ds[i] * s[i] * (1.f - s[i]);
}

// BCELoss(reduction='sum'): -(y log p + (1-y) log(1-p)), logs clamped at -100 like torch
__global__ void bce_sum_kernel(const float* p, const float* y, size_t n, float* partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float lp = fmaxf(logf(p[i]), -100.f), lq = fmaxf(logf(1.f - p[i]), -100.f);
    s -= y[i] * lp + (1.f - y[i]) * lq;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sum_partials_kernel(const float* partial, int n, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}
// dp = g * (p - y) / max(p (1 - p), 1e-12)
__global__ void bce_bwd_kernel(const float* p, const float* y, const float* g, float* dp, size_t n) {
  const float gg = g[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dp[i] = gg * (p[i] - y[i]) / fmaxf(p[i] * (1.f - p[i]), 1e-12f);
}

// dst[g*R + r][:] (+)= scale * src[g][:]     (backward of a per-group mean / sum over rows)
__global__ void grouped_broadcast_kernel(const float* src, int lds_, float* dst, int ldd, int R, int D, float scale,
                                         int accumulate) {
  const size_t row = blockIdx.x;
  const float* s = src + (row / R) * lds_;
  float* d = dst + row * ldd;
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
    f32x4 v = *reinterpret_cast<const f32x4*>(s + c) * scale;
    if (accumulate) v += *reinterpret_cast<const f32x4*>(d + c);
    *reinterpret_cast<f32x4*>(d + c) = v;
  }
}

// out[i] = alpha * a[i] + beta * b[i]   (fp32; the 1/2 CLS + 1/2 patch-mean mixes of the OA tails)
__global__ void axpby_kernel(const float* a, const float* b, float* out, size_t n, float alpha, float beta) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}

}  // namespace oat

using namespace oat;

extern "C" int oat_bmm_strided(const float* A, const float* Bm, float* C, int nb, int I, int J, int K, long long sAb,
                               long long sAi, long long sAk, long long sBb, long long sBk, long long sBj, long long sCb,
                               long long sCi, long long sCj, int sigmoid, int accumulate, void* stream) {
  if (nb <= 0 || I <= 0 || J <= 0 || K <= 0) { set_error("bmm_strided: empty problem"); return -1; }
  if (nb > 65535) { set_error("bmm_strided: batch > 65535"); return -3; }
  BmmArgs a{A, Bm, C, nb, I, J, K, sAb, sAi, sAk, sBb, sBk, sBj, sCb, sCi, sCj, sigmoid, accumulate};
  OAT_LAUNCH(bmm_strided_kernel, dim3((J + 15) / 16, (I + 15) / 16, nb), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("bmm_strided");
}
extern "C" int oat_sigmoid_bwd(const float* s, const float* ds, float* dz, size_t n, void* stream) {
  if (n == 0) return 0;
  int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048;
  OAT_LAUNCH(sigmoid_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s, ds, dz, n);
  return check_launch("sigmoid_bwd");
}
// partial: workspace of 256 floats
extern "C" int oat_bce_sum(const float* p, const float* y, size_t n, float* loss, float* partial, void* stream) {
  if (n == 0) { set_error("bce_sum: empty"); return -1; }
  int blocks = (int)((n + 255) / 256); if (blocks > 256) blocks = 256;
  OAT_LAUNCH(bce_sum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, y, n, partial);
  OAT_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, blocks, loss);
  return check_launch("bce_sum");
}
extern "C" int oat_bce_bwd(const float* p, const float* y, const float* g, float* dp, size_t n, void* stream) {
  if (n == 0) return 0;
  int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048;
  OAT_LAUNCH(bce_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, y, g, dp, n);
  return check_launch("bce_bwd");
}
extern "C" int oat_grouped_broadcast(const float* src, int lds_, float* dst, int ldd, int G, int R, int D, float scale,
                                     int accumulate, void* stream) {
  if (G <= 0 || R <= 0) return 0;
  if (D % 4 || lds_ % 4 || ldd % 4) { set_error("grouped_broadcast: D%4 required"); return -3; }
  OAT_LAUNCH(grouped_broadcast_kernel, dim3((unsigned)((size_t)G * R)), dim3(192), 0, (hipStream_t)stream, src,
                     lds_, dst, ldd, R, D, scale, accumulate);
  return check_launch("grouped_broadcast");
}
extern "C" int oat_axpby(const float* a, const float* b, float* out, size_t n, float alpha, float beta, void* stream) {
  if (n == 0) return 0;
  int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048;
  OAT_LAUNCH(axpby_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, n, alpha, beta);
  return check_launch("axpby");
}

// ---------------------------------------------------------------------------------------------------------
// tag-token masks (oa_model_global_local.py:183-196 builds them with a Python loop over B x O on the host):
// out[b][o][l] = 1 for  n_txt[b]-1+end[b][o-1] <= l < n_txt[b]-1+end[b][o]   (end[b][-1] = 0), else 0
namespace oat {
__global__ void tag_masks_kernel(const long long* ends, const long long* ntxt, float* out, int B, int O, int L) {
  const int b = blockIdx.x / O, o = blockIdx.x % O;
  const long long base = ntxt[b] - 1;
  const long long lo = base + (o > 0 ? ends[(size_t)b * O + o - 1] : 0), hi = base + ends[(size_t)b * O + o];
  for (int l = threadIdx.x; l < L; l += blockDim.x) out[((size_t)b * O + o) * L + l] = (l >= lo && l < hi) ? 1.f : 0.f;
}
}  // namespace oat
// bbox -> patch-grid masks (the dataset-side numpy loops of base_dataset_global_local.py:348-356 and
// base_dataset_region_mem.py:233-247): out[b, o, r * P + c] = 1 iff some box nb that belongs to mask o
// (nb == o without classes; box_class[b, nb] == sel_class[b, o] with) covers cell (r, c):
// int(y0 * P) <= r < ceil(y1 * P) and int(x0 * P) <= c < ceil(x1 * P), with numpy's slice rules for the bounds.
namespace oat {
__device__ __forceinline__ int np_bound(int v, int P) { if (v < 0) v += P; return v < 0 ? 0 : (v > P ? P : v); }
__global__ void patch_masks_kernel(const float* bbox, int ldb, const int* box_class, const int* sel_class, float* out,
                                   int NB, int O, int P) {
  const int b = blockIdx.x / O, o = blockIdx.x % O;
  const int cell = threadIdx.x;
  if (cell >= P * P) return;
  const int r = cell / P, c = cell % P;
  float v = 0.f;
  const int lo = box_class ? 0 : o, hi = box_class ? NB : o + 1;
  for (int nb = lo; nb < hi; ++nb) {
    if (box_class && box_class[(size_t)b * NB + nb] != sel_class[(size_t)b * O + o]) continue;
    const float* bx = bbox + ((size_t)b * NB + nb) * ldb;
    const float x0 = bx[0] * (float)P, y0 = bx[1] * (float)P, x1 = bx[2] * (float)P, y1 = bx[3] * (float)P;
    const int c0 = np_bound((int)x0, P), c1 = np_bound((int)ceilf(x1), P);
    const int r0 = np_bound((int)y0, P), r1 = np_bound((int)ceilf(y1), P);
    if (r >= r0 && r < r1 && c >= c0 && c < c1) v = 1.f;
  }
  out[((size_t)b * O + o) * P * P + cell] = v;
}
}  // namespace oat

extern "C" int oat_patch_masks(const float* bbox, int ldb, const int* box_class, const int* sel_class, float* out, int B,
                               int NB, int O, int P, void* stream) {
  using namespace oat;
  if (B <= 0 || O <= 0) return 0;
  if (!bbox || !out || ldb < 4) { set_error("patch_masks: null pointer or ldb < 4"); return -4; }
  if (P <= 0 || P * P > 1024) { set_error("patch_masks: patch grid must be 1..32 cells per side"); return -3; }
  if ((box_class == nullptr) != (sel_class == nullptr)) { set_error("patch_masks: box_class and sel_class go together"); return -4; }
  if (!box_class && O != NB) { set_error("patch_masks: one mask per box needs O == NB"); return -3; }
  OAT_LAUNCH(patch_masks_kernel, dim3(B * O), dim3((P * P + 63) / 64 * 64), 0, (hipStream_t)stream, bbox, ldb,
                     box_class, sel_class, out, NB, O, P);
  return check_launch("patch_masks");
}

extern "C" int oat_tag_masks(const void* ends, const void* ntxt, float* out, int B, int O, int L, void* stream) {
  if (B <= 0 || O <= 0 || L <= 0) return 0;
  OAT_LAUNCH(oat::tag_masks_kernel, dim3(B * O), dim3(64), 0, (hipStream_t)stream, (const long long*)ends,
                     (const long long*)ntxt, out, B, O, L);
  return oat::check_launch("tag_masks");
}
