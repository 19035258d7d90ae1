// Divided TIME attention + the CLS-query attention of the SpaceTimeTransformer.
//   reference: video_transformer.py:99-135 with the '(b n) f d' pattern (:277-278) and the
//   cls_out = attn(cls_q, k, v) line (:110).
//
// Time attention is B*H*N tiny problems (T queries x (T+1) keys x 64 dims): HBM/latency bound,
// MFMA would idle.  Eight lanes own one problem; lane p holds the 8-dim slice [8p, 8p+8) of every
// q/k/v row of the problem (16-byte loads: 8 lanes cover one 128-byte line), scores are
// 8-lane butterfly reductions, everything else is in-lane fp32.  The reference's gather
// copies 'b (f n) d -> (b n) f d' (3 full passes over QKV) disappear into index math.
//
// CLS-query attention (1 query x all 1+T*N keys per (b,h)) is its own kernel and writes the
// global LSE that both backward kernels use to treat the CLS query as "one more query".
#include "common.h"

namespace oat {

constexpr float T_LOG2E = 1.4426950408889634f;
constexpr float T_LN2 = 0.6931471805599453f;

OAT_DEV float dot8(const bf16x8 a, const bf16x8 b) {
  // v_dot2c_f32_bf16: two bf16 products per instruction, fp32 accumulate, no conversion temporaries
  const bf16x2* a2 = reinterpret_cast<const bf16x2*>(&a);
  const bf16x2* b2 = reinterpret_cast<const bf16x2*>(&b);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_fdot2_f32_bf16(a2[e], b2[e], s, false);
  return s;
}
// exact fp32 FMA chain for the FORWARD scores (v_dot2c measurably doubles the sim-matrix error:
// 3.2e-4 -> 7.1e-4 at ViT-B/16, its accumulate is not round-to-nearest); backward keeps dot2c.
OAT_DEV float dot8x(const bf16x8 a, const bf16x8 b) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += bf2f(a[e]) * bf2f(b[e]);
  return s;
}
// sum over the 8 lanes of a problem group, result in all 8: three v_add_f32_dpp (quad_perm xor 1, xor 2,
// row_half_mirror) instead of three ds_bpermute round trips through the LDS crossbar
OAT_DEV float red8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

struct TimeArgs {
  const bf16* qkv; int ldqkv;
  bf16* out; int ldo;
  float* lse;
  const bf16* dout; int lddo;
  bf16* dqkv; int lddqkv;
  float* cls_side;
  int B, T, N, H, D;
  float scale;
};

// defined in attn_space.hip (the space backward kernel in its block-diagonal TIME mode)
int attn_time_bwd_mfma(const void* qkv, int ldqkv, const void* out, int ldo, const float* lse, const void* dout, int lddo,
                       void* dqkv, int lddqkv, float* cls_side, int B, int T, int N, int H, int D, float scale, hipStream_t s,
                       int* done);

// grid: B*H*ceil(N/8) waves (4 per block); wave -> (b, h, n0), 8-lane group gq -> n = n0 + gq
// The kernel is VALU-bound, not memory-bound (ISA count: ~450 VALU instructions per query frame and wave with packed K / V,
// i.e. 9600 waves x 3.6 K x 4 clk over 1024 SIMDs = 66 us of issue time in an 82 us launch; occupancy 2 -> 4 waves per SIMD
// had changed nothing): every one of the T queries of a position re-converted the SAME T + 1 keys and values from bf16.
// UNPACKED (T <= 8): K and V are converted once into fp32 registers (16 (T + 1) of them) and reused by all T queries -
// the fp32 fma chains are the same, the results bit-identical, the instruction count less than half.
template <int TT>
__global__ __launch_bounds__(256, 2) void attn_time_fwd_kernel(TimeArgs a) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ng = (a.N + 7) / 8;
  if (wid >= a.B * a.H * ng) return;
  const int bh = wid / ng, b = bh / a.H, h = bh % a.H;
  const int n = (wid % ng) * 8 + (lane >> 3);
  const int pl = lane & 7;
  const bool valid = n < a.N;
  const int nn = valid ? n : a.N - 1;
  const size_t cls_row = (size_t)a.B * a.T * a.N + b;
  const int col = h * 64 + pl * 8;
  constexpr bool UNPACKED = TT <= 8;
  bf16x8 k[TT + 1], v[TT + 1];
#pragma unroll
  for (int j = 0; j <= TT; ++j) {
    const size_t r = j < TT ? ((size_t)b * TT + j) * a.N + nn : cls_row;
    k[j] = *reinterpret_cast<const bf16x8*>(a.qkv + r * a.ldqkv + a.D + col);
    v[j] = *reinterpret_cast<const bf16x8*>(a.qkv + r * a.ldqkv + 2 * a.D + col);
  }
  const float c2 = a.scale * T_LOG2E;
  bf16x8 qn = *reinterpret_cast<const bf16x8*>(a.qkv + (((size_t)b * TT) * a.N + nn) * a.ldqkv + col);
  float kf[UNPACKED ? TT + 1 : 1][8], vf[UNPACKED ? TT + 1 : 1][8];
  if constexpr (UNPACKED) {
#pragma unroll
    for (int j = 0; j <= TT; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) { kf[j][e] = bf2f(k[j][e]); vf[j][e] = bf2f(v[j][e]); }
  }
  // rolled frame loop with the next query requested one iteration ahead
#pragma unroll 1
  for (int f = 0; f < TT; ++f) {
    const size_t r = ((size_t)b * TT + f) * a.N + nn;
    if constexpr (!UNPACKED) {
#pragma unroll
      for (int j = 0; j <= TT; ++j) { asm volatile("" : "+v"(k[j])); asm volatile("" : "+v"(v[j])); }   // keep K, V packed (no hoisted fp32 copies)
    }
    const bf16x8 q = qn;
    if (f + 1 < TT) qn = *reinterpret_cast<const bf16x8*>(a.qkv + (r + a.N) * a.ldqkv + col);
    float s[TT + 1], m = -INFINITY;
    if constexpr (UNPACKED) {
      float qf[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[e] = bf2f(q[e]);
#pragma unroll
      for (int j = 0; j <= TT; ++j) {
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d += qf[e] * kf[j][e];            // same chain as dot8x
        s[j] = red8(d) * c2; m = fmaxf(m, s[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j <= TT; ++j) { s[j] = red8(dot8x(q, k[j])) * c2; m = fmaxf(m, s[j]); }
    }
    float l = 0.f, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j <= TT; ++j) {
      const float p = __builtin_amdgcn_exp2f(s[j] - m);      // raw v_exp_f32 (s - m <= 0; no denormal fix-up code)
      l += p;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += p * (UNPACKED ? vf[UNPACKED ? j : 0][e] : bf2f(v[j][e]));
    }
    if (valid) {
      const float inv = 1.0f / l;
      const bf16x8 ob = {f2bf(o[0] * inv), f2bf(o[1] * inv), f2bf(o[2] * inv), f2bf(o[3] * inv),
                         f2bf(o[4] * inv), f2bf(o[5] * inv), f2bf(o[6] * inv), f2bf(o[7] * inv)};
      *reinterpret_cast<bf16x8*>(a.out + r * a.ldo + col) = ob;
      if (pl == 0) a.lse[r * a.H + h] = (m + log2f(l)) * T_LN2;
    }
  }
}

// backward: the MFMA kernel of attn_space.hip on 16-row mini problems (attn_time_bwd_mfma).  The 8-lane VALU backward kernels of rounds
// 1-3 (two-pass and single-read-LDS forms: 219 - 269 us against 140 us for the MFMA form) left the library in round 6.

// ---------------------------------------------------------------- CLS query over all keys
// one workgroup per (b,h): 32 groups of 8 lanes stride over the 1 + T*N keys with an online
// softmax each, then merge through LDS.  Writes out[cls row] and lse[cls row].
__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(TimeArgs a) {
  __shared__ float sm[32], sl[32], so[32][64];
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int grp = threadIdx.x >> 3, pl = threadIdx.x & 7;
  const int S1 = a.T * a.N;                       // patch keys; key S1 = CLS
  const size_t cls_row = (size_t)a.B * S1 + b;
  const size_t row0 = (size_t)b * S1;
  const int col = h * 64 + pl * 8;
  const bf16x8 q = *reinterpret_cast<const bf16x8*>(a.qkv + cls_row * a.ldqkv + col);
  const float c2 = a.scale * T_LOG2E;
  float m = -INFINITY, l = 0.f, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = grp; j <= S1; j += 32) {
    const size_t r = j < S1 ? row0 + j : cls_row;
    const bf16x8 kk = *reinterpret_cast<const bf16x8*>(a.qkv + r * a.ldqkv + a.D + col);
    const bf16x8 vv = *reinterpret_cast<const bf16x8*>(a.qkv + r * a.ldqkv + 2 * a.D + col);
    const float s = red8(dot8x(q, kk)) * c2;
    const float mn = fmaxf(m, s);
    const float alpha = exp2f(m - mn), p = exp2f(s - mn);
    l = l * alpha + p;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = o[e] * alpha + p * bf2f(vv[e]);
    m = mn;
  }
  if (pl == 0) { sm[grp] = m; sl[grp] = l; }
#pragma unroll
  for (int e = 0; e < 8; ++e) so[grp][pl * 8 + e] = o[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    float mm = -INFINITY;
    for (int gq = 0; gq < 32; ++gq) mm = fmaxf(mm, sm[gq]);
    float ll = 0.f, acc = 0.f;
    for (int gq = 0; gq < 32; ++gq) {
      const float w = exp2f(sm[gq] - mm);          // groups that saw no key have m = -inf -> w = 0
      ll += sl[gq] * w;
      acc += so[gq][threadIdx.x] * w;
    }
    a.out[cls_row * a.ldo + h * 64 + threadIdx.x] = f2bf(acc / ll);
    if (threadIdx.x == 0) a.lse[cls_row * a.H + h] = (mm + log2f(ll)) * T_LN2;
  }
}

// The same kernel with a second, PRECISE query: q32 (fp32, row b of [B, D]; e.g. the CLS query computed by
// oat_linear_f32 from the fp32 residual stream) attends the same bf16 keys / values and its context is written in fp32 to
// o32[b].  The bf16-path outputs (out / lse of the CLS row: what backward reads and recomputes from) are unchanged.
__global__ __launch_bounds__(256) void attn_cls_fwd_dual_kernel(TimeArgs a, const float* q32, int ldq32, float* o32, int ldo32) {
  __shared__ float sm[2][32], sl[2][32], so[2][32][64];
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int grp = threadIdx.x >> 3, pl = threadIdx.x & 7;
  const int S1 = a.T * a.N;
  const size_t cls_row = (size_t)a.B * S1 + b;
  const size_t row0 = (size_t)b * S1;
  const int col = h * 64 + pl * 8;
  const bf16x8 q = *reinterpret_cast<const bf16x8*>(a.qkv + cls_row * a.ldqkv + col);
  const float* qp = q32 + (size_t)b * ldq32 + col;
  const f32x4 qa = *reinterpret_cast<const f32x4*>(qp), qb = *reinterpret_cast<const f32x4*>(qp + 4);
  const float c2 = a.scale * T_LOG2E;
  float m = -INFINITY, l = 0.f, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float m2 = -INFINITY, l2 = 0.f, o2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // Four keys per 8-lane group and iteration: their eight 16-byte loads are in flight together (one key per iteration left
  // the kernel latency-bound: 49 dependent round trips, 92 us - longer than the time-attention kernel it runs beside), and
  // the online softmax rescales once per four keys.  The first iteration of every group holds a real key (grp <= 31 < S1).
  for (int j0 = grp; j0 <= S1; j0 += 128) {
    bf16x8 kk[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + 32 * u;
      const size_t r = j < S1 ? row0 + j : cls_row;           // beyond the last key: any valid row, its score is -inf below
      kk[u] = *reinterpret_cast<const bf16x8*>(a.qkv + r * a.ldqkv + a.D + col);
      vv[u] = *reinterpret_cast<const bf16x8*>(a.qkv + r * a.ldqkv + 2 * a.D + col);
    }
    float s1[4], s2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) d += qa[e] * bf2f(kk[u][e]);
#pragma unroll
      for (int e = 0; e < 4; ++e) d += qb[e] * bf2f(kk[u][4 + e]);
      const bool valid = j0 + 32 * u <= S1;
      s1[u] = valid ? red8(dot8x(q, kk[u])) * c2 : -INFINITY;
      s2[u] = valid ? red8(d) * c2 : -INFINITY;
    }
    {
      const float mn = fmaxf(fmaxf(fmaxf(m, s1[0]), fmaxf(s1[1], s1[2])), s1[3]);
      const float alpha = exp2f(m - mn);
      const float p0 = exp2f(s1[0] - mn), p1 = exp2f(s1[1] - mn), p2 = exp2f(s1[2] - mn), p3 = exp2f(s1[3] - mn);
      l = l * alpha + ((p0 + p1) + (p2 + p3));
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o[e] = o[e] * alpha + ((p0 * bf2f(vv[0][e]) + p1 * bf2f(vv[1][e])) + (p2 * bf2f(vv[2][e]) + p3 * bf2f(vv[3][e])));
      m = mn;
    }
    {
      const float mn = fmaxf(fmaxf(fmaxf(m2, s2[0]), fmaxf(s2[1], s2[2])), s2[3]);
      const float alpha = exp2f(m2 - mn);
      const float p0 = exp2f(s2[0] - mn), p1 = exp2f(s2[1] - mn), p2 = exp2f(s2[2] - mn), p3 = exp2f(s2[3] - mn);
      l2 = l2 * alpha + ((p0 + p1) + (p2 + p3));
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o2[e] = o2[e] * alpha + ((p0 * bf2f(vv[0][e]) + p1 * bf2f(vv[1][e])) + (p2 * bf2f(vv[2][e]) + p3 * bf2f(vv[3][e])));
      m2 = mn;
    }
  }
  if (pl == 0) { sm[0][grp] = m; sl[0][grp] = l; sm[1][grp] = m2; sl[1][grp] = l2; }
#pragma unroll
  for (int e = 0; e < 8; ++e) { so[0][grp][pl * 8 + e] = o[e]; so[1][grp][pl * 8 + e] = o2[e]; }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int w = threadIdx.x >> 6, c = threadIdx.x & 63;
    float mm = -INFINITY;
    for (int gq = 0; gq < 32; ++gq) mm = fmaxf(mm, sm[w][gq]);
    float ll = 0.f, acc = 0.f;
    for (int gq = 0; gq < 32; ++gq) {
      const float wt = exp2f(sm[w][gq] - mm);
      ll += sl[w][gq] * wt;
      acc += so[w][gq][c] * wt;
    }
    if (w == 0) {
      a.out[cls_row * a.ldo + h * 64 + c] = f2bf(acc / ll);
      if (c == 0) a.lse[cls_row * a.H + h] = (mm + log2f(ll)) * T_LN2;
    } else {
      o32[(size_t)b * ldo32 + h * 64 + c] = acc / ll;
    }
  }
}

}  // namespace oat

using namespace oat;

#define OAT_TIME_DISPATCH(KERNEL)                                                                     \
  switch (T) {                                                                                        \
    case 1: OAT_LAUNCH(KERNEL<1>, dim3(blocks), dim3(256), 0, s, a); break;                   \
    case 2: OAT_LAUNCH(KERNEL<2>, dim3(blocks), dim3(256), 0, s, a); break;                   \
    case 3: OAT_LAUNCH(KERNEL<3>, dim3(blocks), dim3(256), 0, s, a); break;                   \
    case 4: OAT_LAUNCH(KERNEL<4>, dim3(blocks), dim3(256), 0, s, a); break;                   \
    case 5: OAT_LAUNCH(KERNEL<5>, dim3(blocks), dim3(256), 0, s, a); break;                   \
    case 6: OAT_LAUNCH(KERNEL<6>, dim3(blocks), dim3(256), 0, s, a); break;                   \
    case 7: OAT_LAUNCH(KERNEL<7>, dim3(blocks), dim3(256), 0, s, a); break;                   \
    case 8: OAT_LAUNCH(KERNEL<8>, dim3(blocks), dim3(256), 0, s, a); break;                   \
    case 12: OAT_LAUNCH(KERNEL<12>, dim3(blocks), dim3(256), 0, s, a); break;                 \
    case 16: OAT_LAUNCH(KERNEL<16>, dim3(blocks), dim3(256), 0, s, a); break;                 \
    default: set_error("attn_time: supported frame counts are 1-8, 12, 16"); return -3;               \
  }

extern "C" int oat_attn_time_fwd(const void* qkv, int ldqkv, void* out, int ldo, float* lse, int B, int T, int N,
                                 int H, int D, float scale, void* stream) {
  if (D != H * 64) { set_error("attn_time: head_dim must be 64"); return -3; }
  TimeArgs a{(const bf16*)qkv, ldqkv, (bf16*)out, ldo, lse, nullptr, 0, nullptr, 0, nullptr, B, T, N, H, D, scale};
  hipStream_t s = (hipStream_t)stream;
  const int waves = B * H * ((N + 7) / 8);
  const int blocks = (waves + 3) / 4;
  OAT_TIME_DISPATCH(attn_time_fwd_kernel)
  return check_launch("attn_time_fwd");
}

static int time_bwd(const void* qkv, int ldqkv, const void* out, int ldo, const float* lse, const void* dout, int lddo,
                    void* dqkv, int lddqkv, float* cls_side, int* done, int B, int T, int N, int H, int D, float scale,
                    void* stream);
extern "C" int oat_attn_cls_finalize(float* cls_side, void* dqkv, int lddqkv, int B, int T, int N, int H, int D, void* stream);
extern "C" int oat_attn_time_bwd(const void* qkv, int ldqkv, const void* out, int ldo, const float* lse,
                                 const void* dout, int lddo, void* dqkv, int lddqkv, float* cls_side, int B, int T,
                                 int N, int H, int D, float scale, void* stream) {
  return time_bwd(qkv, ldqkv, out, ldo, lse, dout, lddo, dqkv, lddqkv, cls_side, nullptr, B, T, N, H, D, scale, stream);
}
// As oat_attn_time_bwd followed by oat_attn_cls_finalize in one launch - the last workgroup that feeds cls_side[b][h] writes
// the CLS row itself (`done` = int [B, H], zero on entry and on exit).
extern "C" int oat_attn_time_bwd_fin(const void* qkv, int ldqkv, const void* out, int ldo, const float* lse,
                                     const void* dout, int lddo, void* dqkv, int lddqkv, float* cls_side, int* done,
                                     int B, int T, int N, int H, int D, float scale, void* stream) {
  if (!done) { set_error("attn_time_bwd_fin: null ticket buffer"); return -4; }
  return time_bwd(qkv, ldqkv, out, ldo, lse, dout, lddo, dqkv, lddqkv, cls_side, done, B, T, N, H, D, scale, stream);
}
static int time_bwd(const void* qkv, int ldqkv, const void* out, int ldo, const float* lse, const void* dout, int lddo,
                    void* dqkv, int lddqkv, float* cls_side, int* done, int B, int T, int N, int H, int D, float scale,
                    void* stream) {
  if (D != H * 64) { set_error("attn_time: head_dim must be 64"); return -3; }
  hipStream_t s = (hipStream_t)stream;
  if (T > 16) { set_error("attn_time: supported frame counts are 1-16 (backward), 1-8, 12, 16 (forward)"); return -3; }
  return attn_time_bwd_mfma(qkv, ldqkv, out, ldo, lse, dout, lddo, dqkv, lddqkv, cls_side, B, T, N, H, D, scale, s, done);
}

extern "C" int oat_attn_cls_fwd(const void* qkv, int ldqkv, void* out, int ldo, float* lse, int B, int T, int N, int H,
                                int D, float scale, void* stream) {
  if (D != H * 64) { set_error("attn_cls: head_dim must be 64"); return -3; }
  TimeArgs a{(const bf16*)qkv, ldqkv, (bf16*)out, ldo, lse, nullptr, 0, nullptr, 0, nullptr, B, T, N, H, D, scale};
  OAT_LAUNCH(attn_cls_fwd_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("attn_cls_fwd");
}

/* CLS query with a second, precise query / output (see attn_cls_fwd_dual_kernel). */
extern "C" int oat_attn_cls_fwd_dual(const void* qkv, int ldqkv, void* out, int ldo, float* lse, const float* q32, int ldq32,
                                     float* o32, int ldo32, int B, int T, int N, int H, int D, float scale, void* stream) {
  if (D != H * 64) { set_error("attn_cls: head_dim must be 64"); return -3; }
  if (!q32 || !o32) { set_error("attn_cls_fwd_dual: null pointer"); return -4; }
  TimeArgs a{(const bf16*)qkv, ldqkv, (bf16*)out, ldo, lse, nullptr, 0, nullptr, 0, nullptr, B, T, N, H, D, scale};
  OAT_LAUNCH(attn_cls_fwd_dual_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, a, q32, ldq32, o32, ldo32);
  return check_launch("attn_cls_fwd_dual");
}
