// Text-encoder specific kernels (DistilBERT, called from oa_model.py:113 in the reference via
// HF transformers - third-party code, algorithm restated in oracle/oatrans_oracle.py):
//   embeddings gather (word + position), masked multi-head self-attention fwd / bwd.
// The linear layers, LayerNorms and GELU reuse gemm_nt / gemm_tn / layernorm kernels.
//
// Sequences are short (L <= 512, typically <= 40), so attention is latency-bound VALU work:
// 8 lanes own one (b, h, query) [forward, backward pass A] or one (b, h, key) [backward pass B],
// lane p holding dims [8p, 8p+8) of the 64-dim head (same idiom as attn_time.hip).
// Masked keys (attention_mask == 0) are skipped, which equals HF's masked_fill(finfo.min)
// whenever a row has at least one unmasked key (always true: token 0 = [CLS]).
// Training mode: dropout on the attention probabilities (HF MultiHeadSelfAttention).  With P' = P o m (m = 0 or
// 1/(1-p), element ((b*H + h)*L + i)*L + j of the layer's site, rng.h) and O = P' V:  dP = m o (dO V^T),
// delta_i = sum_j P_ij dP_ij = dO_i . O_i as without dropout, dV_j = sum_i P'_ij dO_i; the softmax statistics are untouched.
#include "rng.h"

namespace oat {

constexpr float X_LOG2E = 1.4426950408889634f;
constexpr float X_LN2 = 0.6931471805599453f;

OAT_DEV float xdot8(const bf16x8 a, const bf16x8 b) {
  // v_dot2c_f32_bf16: two bf16 products per instruction, fp32 accumulate, no conversion temporaries
  const bf16x2* a2 = reinterpret_cast<const bf16x2*>(&a);
  const bf16x2* b2 = reinterpret_cast<const bf16x2*>(&b);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_fdot2_f32_bf16(a2[e], b2[e], s, false);
  return s;
}
OAT_DEV float xdot8x(const bf16x8 a, const bf16x8 b) {      // exact fp32 chain for forward scores
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += bf2f(a[e]) * bf2f(b[e]);
  return s;
}
// sum over the 8 lanes of a problem group, result in all 8: three v_add_f32_dpp (quad_perm xor 1, xor 2,
// row_half_mirror) instead of three ds_bpermute round trips through the LDS crossbar
OAT_DEV float xred8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

// out[m][:] = word[ids[m]][:] + pos[m % L][:]
__global__ void embed_fwd_kernel(const long long* ids, const float* word, const float* pos, float* out, int ld,
                                 int M, int L, int D) {
  const int m = blockIdx.x;
  const float* w = word + (size_t)ids[m] * D;
  const float* p = pos + (size_t)(m % L) * D;
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4)
    *reinterpret_cast<f32x4*>(out + (size_t)m * ld + c) =
        *reinterpret_cast<const f32x4*>(w + c) + *reinterpret_cast<const f32x4*>(p + c);
}
// dword[ids[m]][:] += g[m][:]   (dword zeroed by the caller)
__global__ void embed_bwd_kernel(const long long* ids, const float* g, int ld, float* dword, int M, int D) {
  const int m = blockIdx.x;
  float* w = dword + (size_t)ids[m] * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) atomicAdd(w + c, g[(size_t)m * ld + c]);
}

struct TextArgs {
  const bf16* qkv; int ldqkv;
  const long long* mask;            // [B, L], nonzero = attend
  bf16* out; int ldo;
  float* lse;                       // [M, H]
  float* delta;                     // [M, H]  (backward scratch)
  const bf16* dout; int lddo;
  bf16* dqkv; int lddqkv;
  int B, L, H, D;
  float scale;
  const float* qkv32; int ldqkv32;  // forward only: fp32 q|k|v (precise path); qkv then holds their bf16 roundings
  float* out32; int ldo32;          // forward only: precise context
  DropSite drop;                    // drop.rng == nullptr: no dropout (eval mode)
};

// One workgroup = (b, h, 32 consecutive queries [forward, backward pass A] or keys [pass B]); 8 lanes per row, lane p
// holding dims [8p, 8p + 8).  The rows of the OTHER side (keys / queries) are staged through the LDS in chunks of 64,
// together with their mask flags and - in training mode - the chunk's tile of dropout decisions: the first form of
// these kernels walked the keys with two dependent global loads per step (89 us for an L = 32 attention) and evaluated
// Philox4x32-10 once per (query, key) element in EVERY one of the 8 lanes, twice in the dual forward (80 VALU
// instructions per element against ~30 for the attention itself); now a draw of 4 elements is made once per workgroup.
constexpr int TX_KC = 64;

// dm[r * pitch + jj] = 1 if element (row i0 + r, column j0 + jj) of the [L, L] probability matrix of (b, h) is KEPT
// (same element numbering as attn_drop: ((b H + h) L + i) L + j, four consecutive indices per Philox draw)
OAT_DEV void stage_drop_tile(unsigned char* dm, const DropSite& d, unsigned long long bh, int L, int i0, int nr, int j0,
                             int nk, int pitch) {
  const int nq = (nk + 3) / 4 + 1;                          // quads a row of nk elements can touch
  for (int t = threadIdx.x; t < nr * nq; t += blockDim.x) {
    const int r = t / nq, qi = t % nq;
    const unsigned long long idx0 = (bh * L + i0 + r) * (unsigned long long)L + j0;
    const unsigned long long q = (idx0 >> 2) + qi;
    if (q > ((idx0 + nk - 1) >> 2)) continue;
    const u32x4 dr = drop_draw4(d, q);
    const uint32_t v[4] = {dr.x, dr.y, dr.z, dr.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long jj = (long long)(q * 4 + e) - (long long)idx0;
      if (jj >= 0 && jj < nk) dm[r * pitch + jj] = v[e] >= d.thresh;
    }
  }
}

__global__ __launch_bounds__(256) void attn_text_fwd_kernel(TextArgs a) {
  __shared__ __attribute__((aligned(16))) bf16 k16[TX_KC * 64], v16[TX_KC * 64];
  __shared__ __attribute__((aligned(16))) float k32[TX_KC * 64], v32[TX_KC * 64];
  __shared__ unsigned char keep[TX_KC], dm[32 * TX_KC];
  const int nqc = (a.L + 31) / 32;
  const int qc = blockIdx.x % nqc, h = (blockIdx.x / nqc) % a.H, b = blockIdx.x / (nqc * a.H);
  const int grp = threadIdx.x >> 3, pl = threadIdx.x & 7;
  const int i = qc * 32 + grp;
  const bool valid = i < a.L;
  const size_t row = (size_t)b * a.L + (valid ? i : a.L - 1);
  const int col = h * 64 + pl * 8;
  const bool dual = a.qkv32 != nullptr, drop = a.drop.rng != nullptr;
  const bf16x8 q = *reinterpret_cast<const bf16x8*>(a.qkv + row * a.ldqkv + col);
  f32x4 q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0};
  if (dual) {
    const float* qp = a.qkv32 + row * a.ldqkv32 + col;
    q0 = *reinterpret_cast<const f32x4*>(qp); q1 = *reinterpret_cast<const f32x4*>(qp + 4);
  }
  const float c2 = a.scale * X_LOG2E;
  float m = -INFINITY, l = 0.f, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float m2 = -INFINITY, l2 = 0.f, o2[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // precise path (fp32 q | k | v)
  for (int j0 = 0; j0 < a.L; j0 += TX_KC) {
    const int nk = min(TX_KC, a.L - j0);
    if (j0) __syncthreads();
    for (int t = threadIdx.x; t < nk * 8; t += 256) {
      const int j = t >> 3, c = t & 7;
      const bf16* src = a.qkv + ((size_t)b * a.L + j0 + j) * a.ldqkv + h * 64 + c * 8;
      *reinterpret_cast<bf16x8*>(k16 + j * 64 + c * 8) = *reinterpret_cast<const bf16x8*>(src + a.D);
      *reinterpret_cast<bf16x8*>(v16 + j * 64 + c * 8) = *reinterpret_cast<const bf16x8*>(src + 2 * a.D);
    }
    if (dual) {
      for (int t = threadIdx.x; t < nk * 16; t += 256) {
        const int j = t >> 4, c = t & 15;
        const float* src = a.qkv32 + ((size_t)b * a.L + j0 + j) * a.ldqkv32 + h * 64 + c * 4;
        *reinterpret_cast<f32x4*>(k32 + j * 64 + c * 4) = *reinterpret_cast<const f32x4*>(src + a.D);
        *reinterpret_cast<f32x4*>(v32 + j * 64 + c * 4) = *reinterpret_cast<const f32x4*>(src + 2 * a.D);
      }
    }
    if (threadIdx.x < nk) keep[threadIdx.x] = a.mask[(size_t)b * a.L + j0 + threadIdx.x] != 0;
    if (drop) stage_drop_tile(dm, a.drop, (unsigned long long)b * a.H + h, a.L, qc * 32, min(32, a.L - qc * 32), j0, nk, TX_KC);
    __syncthreads();
    for (int jj = 0; jj < nk; ++jj) {
      if (!keep[jj]) continue;                               // uniform across the workgroup
      const float mult = !drop ? 1.f : (dm[grp * TX_KC + jj] ? a.drop.keep_scale : 0.f);
      {
        const bf16x8 kk = *reinterpret_cast<const bf16x8*>(k16 + jj * 64 + pl * 8);
        const bf16x8 vv = *reinterpret_cast<const bf16x8*>(v16 + jj * 64 + pl * 8);
        const float s = xred8(xdot8x(q, kk)) * c2;
        const float mn = fmaxf(m, s);
        const float alpha = exp2f(m - mn), p = exp2f(s - mn);
        l = l * alpha + p;
        const float pd = p * mult;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = o[e] * alpha + pd * bf2f(vv[e]);
        m = mn;
      }
      // Precise path (forward value of the layer): the same attention on the fp32 q | k | v.  The bf16 results above -
      // what backward differentiates and recomputes its probabilities from - stay exactly self-consistent.
      if (dual) {
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(k32 + jj * 64 + pl * 8), k1 = *reinterpret_cast<const f32x4*>(k32 + jj * 64 + pl * 8 + 4);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(v32 + jj * 64 + pl * 8), v1 = *reinterpret_cast<const f32x4*>(v32 + jj * 64 + pl * 8 + 4);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) d += q0[e] * k0[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) d += q1[e] * k1[e];
        const float s = xred8(d) * c2;
        const float mn = fmaxf(m2, s);
        const float alpha = exp2f(m2 - mn), p = exp2f(s - mn);
        l2 = l2 * alpha + p;
        const float pd = p * mult;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o2[e] = o2[e] * alpha + pd * v0[e];
          o2[4 + e] = o2[4 + e] * alpha + pd * v1[e];
        }
        m2 = mn;
      }
    }
  }
  if (valid) {
    const float inv = 1.0f / l;
    const bf16x8 ob = {f2bf(o[0] * inv), f2bf(o[1] * inv), f2bf(o[2] * inv), f2bf(o[3] * inv),
                       f2bf(o[4] * inv), f2bf(o[5] * inv), f2bf(o[6] * inv), f2bf(o[7] * inv)};
    *reinterpret_cast<bf16x8*>(a.out + row * a.ldo + col) = ob;
    if (pl == 0) a.lse[row * a.H + h] = (m + log2f(l)) * X_LN2;
    if (dual) {
      const float inv2 = 1.0f / l2;
      float* op = a.out32 + row * a.ldo32 + col;
      *reinterpret_cast<f32x4*>(op) = f32x4{o2[0] * inv2, o2[1] * inv2, o2[2] * inv2, o2[3] * inv2};
      *reinterpret_cast<f32x4*>(op + 4) = f32x4{o2[4] * inv2, o2[5] * inv2, o2[6] * inv2, o2[7] * inv2};
    }
  }
}

// pass A: per query -> delta, dq
__global__ __launch_bounds__(256) void attn_text_bwd_q_kernel(TextArgs a) {
  __shared__ __attribute__((aligned(16))) bf16 k16[TX_KC * 64], v16[TX_KC * 64];
  __shared__ unsigned char keep[TX_KC], dm[32 * TX_KC];
  const int nqc = (a.L + 31) / 32;
  const int qc = blockIdx.x % nqc, h = (blockIdx.x / nqc) % a.H, b = blockIdx.x / (nqc * a.H);
  const int grp = threadIdx.x >> 3, pl = threadIdx.x & 7;
  const int i = qc * 32 + grp;
  const bool valid = i < a.L;
  const size_t row = (size_t)b * a.L + (valid ? i : a.L - 1);
  const int col = h * 64 + pl * 8;
  const bool drop = a.drop.rng != nullptr;
  const bf16x8 q = *reinterpret_cast<const bf16x8*>(a.qkv + row * a.ldqkv + col);
  const bf16x8 go = *reinterpret_cast<const bf16x8*>(a.dout + row * a.lddo + col);
  const bf16x8 oo = *reinterpret_cast<const bf16x8*>(a.out + row * a.ldo + col);
  const float delta = xred8(xdot8(go, oo));
  const float lse2 = a.lse[row * a.H + h] * X_LOG2E;
  const float c2 = a.scale * X_LOG2E;
  float dq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j0 = 0; j0 < a.L; j0 += TX_KC) {
    const int nk = min(TX_KC, a.L - j0);
    if (j0) __syncthreads();
    for (int t = threadIdx.x; t < nk * 8; t += 256) {
      const int j = t >> 3, c = t & 7;
      const bf16* src = a.qkv + ((size_t)b * a.L + j0 + j) * a.ldqkv + h * 64 + c * 8;
      *reinterpret_cast<bf16x8*>(k16 + j * 64 + c * 8) = *reinterpret_cast<const bf16x8*>(src + a.D);
      *reinterpret_cast<bf16x8*>(v16 + j * 64 + c * 8) = *reinterpret_cast<const bf16x8*>(src + 2 * a.D);
    }
    if (threadIdx.x < nk) keep[threadIdx.x] = a.mask[(size_t)b * a.L + j0 + threadIdx.x] != 0;
    if (drop) stage_drop_tile(dm, a.drop, (unsigned long long)b * a.H + h, a.L, qc * 32, min(32, a.L - qc * 32), j0, nk, TX_KC);
    __syncthreads();
    for (int jj = 0; jj < nk; ++jj) {
      if (!keep[jj]) continue;
      const float mult = !drop ? 1.f : (dm[grp * TX_KC + jj] ? a.drop.keep_scale : 0.f);
      const bf16x8 kk = *reinterpret_cast<const bf16x8*>(k16 + jj * 64 + pl * 8);
      const bf16x8 vv = *reinterpret_cast<const bf16x8*>(v16 + jj * 64 + pl * 8);
      const float p = exp2f(xred8(xdot8(q, kk)) * c2 - lse2);
      const float ds = p * (xred8(xdot8(go, vv)) * mult - delta) * a.scale;
#pragma unroll
      for (int e = 0; e < 8; ++e) dq[e] += ds * bf2f(kk[e]);
    }
  }
  if (valid) {
    const bf16x8 ob = {f2bf(dq[0]), f2bf(dq[1]), f2bf(dq[2]), f2bf(dq[3]), f2bf(dq[4]), f2bf(dq[5]), f2bf(dq[6]), f2bf(dq[7])};
    *reinterpret_cast<bf16x8*>(a.dqkv + row * a.lddqkv + col) = ob;
    if (pl == 0) a.delta[row * a.H + h] = delta;
  }
}

// pass B: per key -> dk, dv (masked keys receive exact zeros); queries staged in chunks with their lse / delta
__global__ __launch_bounds__(256) void attn_text_bwd_kv_kernel(TextArgs a) {
  __shared__ __attribute__((aligned(16))) bf16 q16[TX_KC * 64], g16[TX_KC * 64];
  __shared__ float lse_s[TX_KC], del_s[TX_KC];
  __shared__ unsigned char dm[TX_KC * 32];
  const int nkc = (a.L + 31) / 32;
  const int kc = blockIdx.x % nkc, h = (blockIdx.x / nkc) % a.H, b = blockIdx.x / (nkc * a.H);
  const int grp = threadIdx.x >> 3, pl = threadIdx.x & 7;
  const int j = kc * 32 + grp;
  const bool valid = j < a.L;
  const size_t row = (size_t)b * a.L + (valid ? j : a.L - 1);
  const int col = h * 64 + pl * 8;
  const bool drop = a.drop.rng != nullptr;
  const bf16x8 kk = *reinterpret_cast<const bf16x8*>(a.qkv + row * a.ldqkv + a.D + col);
  const bf16x8 vv = *reinterpret_cast<const bf16x8*>(a.qkv + row * a.ldqkv + 2 * a.D + col);
  const bool keepk = valid && a.mask[(size_t)b * a.L + (valid ? j : a.L - 1)] != 0;
  const float c2 = a.scale * X_LOG2E;
  float dk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i0 = 0; i0 < a.L; i0 += TX_KC) {
    const int nq = min(TX_KC, a.L - i0);
    if (i0) __syncthreads();
    for (int t = threadIdx.x; t < nq * 8; t += 256) {
      const int i = t >> 3, c = t & 7;
      const size_t r = (size_t)b * a.L + i0 + i;
      *reinterpret_cast<bf16x8*>(q16 + i * 64 + c * 8) = *reinterpret_cast<const bf16x8*>(a.qkv + r * a.ldqkv + h * 64 + c * 8);
      *reinterpret_cast<bf16x8*>(g16 + i * 64 + c * 8) = *reinterpret_cast<const bf16x8*>(a.dout + r * a.lddo + h * 64 + c * 8);
    }
    if (threadIdx.x < nq) {
      const size_t r = (size_t)b * a.L + i0 + threadIdx.x;
      lse_s[threadIdx.x] = a.lse[r * a.H + h] * X_LOG2E;
      del_s[threadIdx.x] = a.delta[r * a.H + h];
    }
    if (drop) stage_drop_tile(dm, a.drop, (unsigned long long)b * a.H + h, a.L, i0, nq, kc * 32, min(32, a.L - kc * 32), 32);
    __syncthreads();
    if (keepk) {
      for (int ii = 0; ii < nq; ++ii) {
        const bf16x8 q = *reinterpret_cast<const bf16x8*>(q16 + ii * 64 + pl * 8);
        const bf16x8 go = *reinterpret_cast<const bf16x8*>(g16 + ii * 64 + pl * 8);
        const float p = exp2f(xred8(xdot8(q, kk)) * c2 - lse_s[ii]);
        const float mij = !drop ? 1.f : (dm[ii * 32 + grp] ? a.drop.keep_scale : 0.f);
        const float ds = p * (xred8(xdot8(go, vv)) * mij - del_s[ii]) * a.scale;
        const float pd = p * mij;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dk[e] += ds * bf2f(q[e]); dv[e] += pd * bf2f(go[e]); }
      }
    }
  }
  if (valid) {
    const bf16x8 kb = {f2bf(dk[0]), f2bf(dk[1]), f2bf(dk[2]), f2bf(dk[3]), f2bf(dk[4]), f2bf(dk[5]), f2bf(dk[6]), f2bf(dk[7])};
    const bf16x8 vb = {f2bf(dv[0]), f2bf(dv[1]), f2bf(dv[2]), f2bf(dv[3]), f2bf(dv[4]), f2bf(dv[5]), f2bf(dv[6]), f2bf(dv[7])};
    *reinterpret_cast<bf16x8*>(a.dqkv + row * a.lddqkv + a.D + col) = kb;
    *reinterpret_cast<bf16x8*>(a.dqkv + row * a.lddqkv + 2 * a.D + col) = vb;
  }
}

// y = relu(x) as bf16 (txt_proj = ReLU -> Linear, oa_model.py:68) ; backward mask: dx = dy * (x > 0)
__global__ void relu_bf16_kernel(const float* x, int ldx, bf16* y, int ldy, int M, int D) {
  const int m = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += blockDim.x) y[(size_t)m * ldy + c] = f2bf(fmaxf(x[(size_t)m * ldx + c], 0.f));
}
__global__ void relu_bwd_kernel(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int M, int D) {
  const int m = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += blockDim.x)
    dx[(size_t)m * lddx + c] = x[(size_t)m * ldx + c] > 0.f ? dy[(size_t)m * lddy + c] : 0.f;
}

}  // namespace oat

using namespace oat;

extern "C" int oat_embed_fwd(const void* ids, const float* word, const float* pos, float* out, int ld, int M, int L,
                             int D, void* stream) {
  if (M <= 0) return 0;
  if (D % 4 || ld % 4) { set_error("embed_fwd: D%4 required"); return -3; }
  OAT_LAUNCH(embed_fwd_kernel, dim3(M), dim3(192), 0, (hipStream_t)stream, (const long long*)ids, word, pos, out,
                     ld, M, L, D);
  return check_launch("embed_fwd");
}
extern "C" int oat_embed_bwd(const void* ids, const float* g, int ld, float* dword, int M, int D, void* stream) {
  if (M <= 0) return 0;
  OAT_LAUNCH(embed_bwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const long long*)ids, g, ld, dword, M, D);
  return check_launch("embed_bwd");
}

static int attn_text_fwd_launch(oat::TextArgs a, void* stream) {
  using namespace oat;
  if (a.D != a.H * 64) { set_error("attn_text: head_dim must be 64"); return -3; }
  OAT_LAUNCH(attn_text_fwd_kernel, dim3(a.B * a.H * ((a.L + 31) / 32)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("attn_text_fwd");
}
extern "C" int oat_attn_text_fwd(const void* qkv, int ldqkv, const void* mask, void* out, int ldo, float* lse, int B,
                                 int L, int H, int D, float scale, void* stream) {
  oat::TextArgs a{(const bf16*)qkv, ldqkv, (const long long*)mask, (bf16*)out, ldo, lse, nullptr, nullptr, 0, nullptr, 0, B, L, H, D, scale,
                  nullptr, 0, nullptr, 0, DropSite{nullptr, 0, 0, 1.f}};
  return attn_text_fwd_launch(a, stream);
}
// As oat_attn_text_fwd, plus the PRECISE forward value: the same masked attention on fp32 q|k|v (qkv32; `qkv` holds their
// bf16 roundings) written to out32.  out / lse stay the bf16-path results backward uses.
// drop_p > 0 with a device rng state (oat_rng_tick): dropout on the probabilities, site `drop_site` (training mode).
extern "C" int oat_attn_text_fwd_dual(const void* qkv, int ldqkv, const float* qkv32, int ldqkv32, const void* mask, void* out,
                                      int ldo, float* out32, int ldo32, float* lse, int B, int L, int H, int D, float scale,
                                      float drop_p, const void* rng, unsigned drop_site, void* stream) {
  if (!qkv32 || !out32) { oat::set_error("attn_text_fwd_dual: null pointer"); return -4; }
  if (drop_p > 0.f && !rng) { oat::set_error("attn_text_fwd_dual: dropout needs an rng state"); return -4; }
  oat::TextArgs a{(const bf16*)qkv, ldqkv, (const long long*)mask, (bf16*)out, ldo, lse, nullptr, nullptr, 0, nullptr, 0, B, L, H, D, scale,
                  qkv32, ldqkv32, out32, ldo32,
                  drop_p > 0.f ? oat::make_drop_site(rng, drop_site, drop_p) : oat::DropSite{nullptr, 0, 0, 1.f}};
  return attn_text_fwd_launch(a, stream);
}
// delta: fp32 [B*L, H] scratch
extern "C" int oat_attn_text_bwd(const void* qkv, int ldqkv, const void* mask, const void* out, int ldo,
                                 const float* lse, float* delta, const void* dout, int lddo, void* dqkv, int lddqkv,
                                 int B, int L, int H, int D, float scale, float drop_p, const void* rng, unsigned drop_site,
                                 void* stream) {
  if (D != H * 64) { set_error("attn_text: head_dim must be 64"); return -3; }
  if (drop_p > 0.f && !rng) { set_error("attn_text_bwd: dropout needs an rng state"); return -4; }
  TextArgs a{(const bf16*)qkv, ldqkv, (const long long*)mask, (bf16*)out, ldo, (float*)lse, delta, (const bf16*)dout, lddo,
             (bf16*)dqkv, lddqkv, B, L, H, D, scale, nullptr, 0, nullptr, 0,
             drop_p > 0.f ? make_drop_site(rng, drop_site, drop_p) : DropSite{nullptr, 0, 0, 1.f}};
  const int blocks = B * H * ((L + 31) / 32);
  hipStream_t s = (hipStream_t)stream;
  OAT_LAUNCH(attn_text_bwd_q_kernel, dim3(blocks), dim3(256), 0, s, a);
  OAT_LAUNCH(attn_text_bwd_kv_kernel, dim3(blocks), dim3(256), 0, s, a);
  return check_launch("attn_text_bwd");
}

extern "C" int oat_relu_bf16(const float* x, int ldx, void* y, int ldy, int M, int D, void* stream) {
  if (M <= 0) return 0;
  OAT_LAUNCH(relu_bf16_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, ldx, (bf16*)y, ldy, M, D);
  return check_launch("relu_bf16");
}
extern "C" int oat_relu_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int M, int D,
                            void* stream) {
  if (M <= 0) return 0;
  OAT_LAUNCH(relu_bwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, ldx, dy, lddy, dx, lddx, M, D);
  return check_launch("relu_bwd");
}
