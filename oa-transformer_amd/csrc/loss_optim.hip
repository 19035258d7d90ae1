// Fused InfoNCE (sim_matrix + NormSoftmaxLoss, forward AND backward) and multi-tensor AdamW.
//
//   sim_matrix        /root/reference/OATrans/model/oa_model.py:192-200 (identical copy model.py:164-172)
//   NormSoftmaxLoss   /root/reference/OATrans/model/loss.py:13-25   (temperature 0.05)
//   AllGather_multi   /root/reference/OATrans/trainer/trainer_dist.py:29-45 - backward keeps only the
//                     local rank's rows, so gradients are produced for rows [r0, r0 + nloc) only
//   optimizer         transformers.AdamW (4.6.0 defaults; train_dist_multi.py:66) - third party,
//                     published algorithm: decoupled weight decay, eps added OUTSIDE the bias-corrected
//                     sqrt (denom = sqrt(v) + eps; step = lr * sqrt(1-b2^t) / (1-b1^t)).
//
// The loss problem is tiny (n <= 512 rows of 256 floats): everything is fp32 VALU, a handful of
// launches, no host synchronisation (the loss scalar stays on the device).
#include "common.h"

namespace oat {

// xn[i] = x[i] / max(||x[i]||, eps) ; nrm[i] = max(||x[i]||, eps) ; clamped[i] = (||x|| < eps)
__global__ void l2norm_rows_kernel(const float* x, float* xn, float* nrm, int n, int d, float eps) {
  __shared__ float red[4];
  const int i = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < d; c += blockDim.x) { const float v = x[(size_t)i * d + c]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float tot = red[0] + red[1] + red[2] + red[3];
  const float nr = fmaxf(sqrtf(tot), eps);
  for (int c = threadIdx.x; c < d; c += blockDim.x) xn[(size_t)i * d + c] = x[(size_t)i * d + c] / nr;
  if (threadIdx.x == 0) nrm[i] = sqrtf(tot) < eps ? -nr : nr;     // sign flags the clamped branch
}

// sim[i][j] = sum_c a[i][c] * b[j][c]      (block = 16x16 outputs)
__global__ void sim_kernel(const float* a, const float* b, float* sim, int n, int m, int d) {
  __shared__ float sa[16][33], sb[16][33];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
  float acc = 0.f;
  for (int c0 = 0; c0 < d; c0 += 32) {
    for (int t = threadIdx.x; t < 16 * 32; t += 256) {
      const int r = t >> 5, c = t & 31;
      sa[r][c] = (blockIdx.y * 16 + r < n && c0 + c < d) ? a[(size_t)(blockIdx.y * 16 + r) * d + c0 + c] : 0.f;
      sb[r][c] = (blockIdx.x * 16 + r < m && c0 + c < d) ? b[(size_t)(blockIdx.x * 16 + r) * d + c0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 32; ++c) acc += sa[ty][c] * sb[tx][c];
    __syncthreads();
  }
  if (i < n && j < m) sim[(size_t)i * m + j] = acc;
}

// lse_r[i] = logsumexp_j(sim[i][j] / tau) ; lse_c[j] = logsumexp_i(sim[i][j] / tau)   (n x n)
__global__ void lse_kernel(const float* sim, float* lse_r, float* lse_c, int n, float inv_tau) {
  __shared__ float red[4];
  const int k = blockIdx.x % n;
  const bool col = blockIdx.x >= n;
  float mx = -INFINITY;
  for (int t = threadIdx.x; t < n; t += blockDim.x) mx = fmaxf(mx, (col ? sim[(size_t)t * n + k] : sim[(size_t)k * n + t]) * inv_tau);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int t = threadIdx.x; t < n; t += blockDim.x) s += __expf((col ? sim[(size_t)t * n + k] : sim[(size_t)k * n + t]) * inv_tau - mx);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) (col ? lse_c : lse_r)[k] = mx + logf(red[0] + red[1] + red[2] + red[3]);
}

// loss = -(1/n) sum_i (x_ii/tau - lse_r[i]) - (1/n) sum_j (x_jj/tau - lse_c[j])
// G[i][j] = dL/dsim = (1/(n tau)) * (softmax_r + softmax_c - 2 delta_ij)       (one block, then grid)
__global__ void loss_kernel(const float* sim, const float* lse_r, const float* lse_c, float* loss, int n, float inv_tau) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += 2.f * sim[(size_t)i * n + i] * inv_tau - lse_r[i] - lse_c[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = -(red[0] + red[1] + red[2] + red[3]) / n;
}
__global__ void gsim_kernel(const float* sim, const float* lse_r, const float* lse_c, float* G, int n, float inv_tau) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * n) return;
  const int i = idx / n, j = idx % n;
  const float x = sim[idx] * inv_tau;
  G[idx] = (__expf(x - lse_r[i]) + __expf(x - lse_c[j]) - (i == j ? 2.f : 0.f)) * inv_tau / n;
}

// d(an)[i] = sum_j G[i][j] bn[j]  (transpose = 0)   or   sum_j G[j][i] bn[j]  (transpose = 1), rows r0..r0+nloc;
// then through the normalisation: da = (dan - an (an . dan)) / nrm   (dan / eps on the clamped branch)
// G is [n, m]; transpose = 0: row i of G against bn [m, d]; transpose = 1: column i of G against bn [n, d]
__global__ void gemb_kernel(const float* G, const float* bn, const float* an, const float* nrm, float* da, int n, int m,
                            int d, int r0, int transpose) {
  __shared__ float red[4];
  extern __shared__ float dan[];               // [d]
  const int i = r0 + blockIdx.x;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float s = 0.f;
    const int cnt = transpose ? n : m;
    for (int j = 0; j < cnt; ++j) s += (transpose ? G[(size_t)j * m + i] : G[(size_t)i * m + j]) * bn[(size_t)j * d + c];
    dan[c] = s;
  }
  __syncthreads();
  float dot = 0.f;
  for (int c = threadIdx.x; c < d; c += blockDim.x) dot += dan[c] * an[(size_t)i * d + c];
  dot = wave_sum(dot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
  __syncthreads();
  dot = red[0] + red[1] + red[2] + red[3];
  const float nr = nrm[i];
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    const float v = nr < 0.f ? dan[c] / (-nr) : (dan[c] - an[(size_t)i * d + c] * dot) / nr;
    da[(size_t)blockIdx.x * d + c] = v;
  }
}

// ---------------------------------------------------------------- AdamW over a flat fp32 range
// `coef` (optional, device memory): {lr, 1 - b1^step, 1 - b2^step} written by adam_tick_kernel - the step-dependent
// scalars then live on the device and a captured launch (hipGraph replay) stays valid from step to step.
__global__ void adam_tick_kernel(int* step, const float* lr, float b1, float b2, float* coef) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int t = *step + 1;
    *step = t;
    coef[0] = *lr;
    coef[1] = 1.f - powf(b1, (float)t);
    coef[2] = 1.f - powf(b2, (float)t);
  }
}

__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2, int hf_style, float gscale,
                             const float* coef = nullptr) {
  if (coef != nullptr) { lr = coef[0]; bc1 = coef[1]; bc2 = coef[2]; }
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  // The four loads of a lane's NEXT quad are requested before the three stores of the current one: vmcnt retires in issue order, so a
  // load issued behind the stores would only be usable once they are written (one store round trip per iteration and lane).
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  f32x4 pp = {}, gg = {}, mm = {}, vv = {};
  if (i + 4 <= n) {
    pp = *reinterpret_cast<f32x4*>(p + i); gg = *reinterpret_cast<const f32x4*>(g + i);
    mm = *reinterpret_cast<f32x4*>(m + i); vv = *reinterpret_cast<f32x4*>(v + i);
  }
  for (; i < n; i += stride) {
    if (i + 4 <= n) {
      const size_t j = i + stride;
      f32x4 pn = {}, gn = {}, mn = {}, vn = {};
      if (j + 4 <= n) {
        pn = *reinterpret_cast<f32x4*>(p + j); gn = *reinterpret_cast<const f32x4*>(g + j);
        mn = *reinterpret_cast<f32x4*>(m + j); vn = *reinterpret_cast<f32x4*>(v + j);
      }
      gg *= gscale;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mm[e] = b1 * mm[e] + (1.f - b1) * gg[e];
        vv[e] = b2 * vv[e] + (1.f - b2) * gg[e] * gg[e];
        if (hf_style) {
          pp[e] -= lr * sqrtf(bc2) / bc1 * mm[e] / (sqrtf(vv[e]) + eps);
          pp[e] -= lr * wd * pp[e];
        } else {
          pp[e] *= 1.f - lr * wd;
          pp[e] -= lr / bc1 * mm[e] / (sqrtf(vv[e]) / sqrtf(bc2) + eps);
        }
      }
      *reinterpret_cast<f32x4*>(p + i) = pp;
      *reinterpret_cast<f32x4*>(m + i) = mm;
      *reinterpret_cast<f32x4*>(v + i) = vv;
      pp = pn; gg = gn; mm = mn; vv = vn;
    } else {
      for (size_t k = i; k < n; ++k) {
        const float gk = g[k] * gscale;
        const float mk = b1 * m[k] + (1.f - b1) * gk, vk = b2 * v[k] + (1.f - b2) * gk * gk;
        float pk = p[k];
        if (hf_style) { pk -= lr * sqrtf(bc2) / bc1 * mk / (sqrtf(vk) + eps); pk -= lr * wd * pk; }
        else { pk *= 1.f - lr * wd; pk -= lr / bc1 * mk / (sqrtf(vk) / sqrtf(bc2) + eps); }
        p[k] = pk; m[k] = mk; v[k] = vk;
      }
    }
  }
}

}  // namespace oat

using namespace oat;

// ws layout for sim_matrix: tn [n,d] | vn [m,d] | tnrm [n] | vnrm [m]
extern "C" size_t oat_sim_workspace_floats(int n, int m, int d) { return (size_t)(n + m) * d + n + m; }

// sim[n,m] = normalise(t) @ normalise(v)^T  (oa_model.py:192-200); ws keeps what backward needs
extern "C" int oat_sim_matrix_fwd(const float* t, const float* v, int n, int m, int d, float eps, float* sim, float* ws,
                                  void* stream) {
  if (n <= 0 || m <= 0 || d <= 0) { set_error("sim_matrix: empty problem"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  float* tn = ws; float* vn = tn + (size_t)n * d; float* tnrm = vn + (size_t)m * d; float* vnrm = tnrm + n;
  OAT_LAUNCH(l2norm_rows_kernel, dim3(n), dim3(256), 0, s, t, tn, tnrm, n, d, eps);
  OAT_LAUNCH(l2norm_rows_kernel, dim3(m), dim3(256), 0, s, v, vn, vnrm, m, d, eps);
  OAT_LAUNCH(sim_kernel, dim3((m + 15) / 16, (n + 15) / 16), dim3(256), 0, s, tn, vn, sim, n, m, d);
  return check_launch("sim_matrix_fwd");
}
// dt [tl, d] = d/d t[t0 : t0+tl], dv [vl, d] = d/d v[v0 : v0+vl] given G = dL/dsim [n, m]
extern "C" int oat_sim_matrix_bwd(const float* G, const float* ws, int n, int m, int d, float* dt, int t0, int tl,
                                  float* dv, int v0, int vl, void* stream) {
  if (d > 8192) { set_error("sim_matrix: embedding dim too large"); return -3; }
  if (t0 < 0 || t0 + tl > n || v0 < 0 || v0 + vl > m) { set_error("sim_matrix_bwd: row range outside the matrix"); return -2; }
  hipStream_t s = (hipStream_t)stream;
  const float* tn = ws; const float* vn = tn + (size_t)n * d; const float* tnrm = vn + (size_t)m * d; const float* vnrm = tnrm + n;
  if (dt && tl > 0) OAT_LAUNCH(gemb_kernel, dim3(tl), dim3(256), d * sizeof(float), s, G, vn, tn, tnrm, dt, n, m, d, t0, 0);
  if (dv && vl > 0) OAT_LAUNCH(gemb_kernel, dim3(vl), dim3(256), d * sizeof(float), s, G, tn, vn, vnrm, dv, n, m, d, v0, 1);
  return check_launch("sim_matrix_bwd");
}
// NormSoftmaxLoss (loss.py:13-25) on a square sim matrix: loss[1] and, if G != NULL, G = dloss/dsim.  ws: 2n floats
extern "C" int oat_norm_softmax_loss(const float* sim, int n, float temperature, float* loss, float* G, float* ws,
                                     void* stream) {
  if (n <= 0) { set_error("norm_softmax_loss: empty problem"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  float* lse_r = ws; float* lse_c = ws + n;
  const float inv_tau = 1.f / temperature;
  OAT_LAUNCH(lse_kernel, dim3(2 * n), dim3(256), 0, s, sim, lse_r, lse_c, n, inv_tau);
  OAT_LAUNCH(loss_kernel, dim3(1), dim3(256), 0, s, sim, lse_r, lse_c, loss, n, inv_tau);
  if (G) OAT_LAUNCH(gsim_kernel, dim3((n * n + 255) / 256), dim3(256), 0, s, sim, lse_r, lse_c, G, n, inv_tau);
  return check_launch("norm_softmax_loss");
}

extern "C" size_t oat_infonce_workspace_floats(int n, int d) {
  return oat_sim_workspace_floats(n, n, d) + 2 * (size_t)n + 2 * (size_t)n * n;
}

// Fused trainer path (trainer_dist.py:159-163): t, v fp32 [n, d] = all ranks' rows in rank order.
// Outputs loss[1], sim [n,n] (optional), dt / dv fp32 [nloc, d] for the local rows [r0, r0+nloc).
extern "C" int oat_infonce(const float* t, const float* v, int n, int d, float temperature, float eps, float* loss,
                           float* sim_out, float* dt, float* dv, int r0, int nloc, float* ws, void* stream) {
  if (r0 < 0 || r0 + nloc > n) { set_error("infonce: local row range outside [0, n)"); return -2; }
  float* lsews = ws + oat_sim_workspace_floats(n, n, d);
  float* sim = lsews + 2 * (size_t)n; float* G = sim + (size_t)n * n;
  int rc = oat_sim_matrix_fwd(t, v, n, n, d, eps, sim, ws, stream);
  if (rc) return rc;
  rc = oat_norm_softmax_loss(sim, n, temperature, loss, (dt || dv) ? G : nullptr, lsews, stream);
  if (rc) return rc;
  if (sim_out) (void)hipMemcpyAsync(sim_out, sim, (size_t)n * n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (dt || dv) return oat_sim_matrix_bwd(G, ws, n, n, d, dt, r0, nloc, dv, r0, nloc, stream);
  return 0;
}

// Device-resident step state for captured (hipGraph) training steps: *step += 1; coef = {*lr, 1 - b1^step, 1 - b2^step}.
extern "C" int oat_adam_tick(int* step, const float* lr, float beta1, float beta2, float* coef, void* stream) {
  if (!step || !lr || !coef) { set_error("adam_tick: null pointer"); return -4; }
  OAT_LAUNCH(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step, lr, beta1, beta2, coef);
  return check_launch("adam_tick");
}
// oat_adamw with the step-dependent scalars read from `coef` (device memory, see oat_adam_tick).
extern "C" int oat_adamw_dev(float* p, const float* g, float* m, float* v, size_t n, const float* coef, float beta1,
                             float beta2, float eps, float weight_decay, int hf_style, float gscale, void* stream) {
  if (n == 0) return 0;
  if (!coef) { set_error("adamw_dev: null pointer"); return -4; }
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  OAT_LAUNCH(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, 0.f, beta1,
                     beta2, eps, weight_decay, 1.f, 1.f, hf_style, gscale, coef);
  return check_launch("adamw_dev");
}

// One AdamW step over a flat fp32 range.  step >= 1.  hf_style = 1: transformers.AdamW; 0: torch.optim.AdamW.
// gscale multiplies the gradient first (1/world_size when gradients were SUM-all-reduced).
extern "C" int oat_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int step, int hf_style, float gscale, void* stream) {
  if (n == 0) return 0;
  if (step < 1) { set_error("adamw: step must be >= 1"); return -2; }
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  OAT_LAUNCH(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, bc1, bc2, hf_style, gscale, (const float*)nullptr);
  return check_launch("adamw");
}
