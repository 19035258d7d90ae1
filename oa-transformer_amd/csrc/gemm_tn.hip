// bf16 MFMA GEMM, "TN" form (weight gradients):  out[N1,N2] = sum_m P[m,N1]^T * Q[m,N2]
//
// This is the reduction-over-rows product autograd runs for every nn.Linear weight of the
// reference (dW = dY^T X; video_transformer.py:102,133,46-50 in backward).  Both operands
// have the reduction index m as their SLOW (row) index, so MFMA fragments (8 consecutive
// k per lane) are column slices of the LDS tile: gfx950's ds_read_b64_tr_b16 transpose-read
// delivers them straight from the row-major image, no transposed copies of activations.
//
// v1: 128x128 output tile, 64 rows of m per stage, 4 waves (2x2) x (4x4 MFMA 16x16x32),
// global_load_lds staging (double-buffered 64 KB), split over m across grid.y with fp32
// slabs + a deterministic reduce kernel (no atomics).
// Contract: rows [M, round_up(M,64)) of P and Q must be READABLE (inside the allocation); their
// contents are ignored (the ragged tail of the last chunk is zeroed in LDS).
#include "common.h"

namespace oat {

constexpr int TB = 128;            // output tile edge
constexpr int TK = 64;             // m rows per stage
constexpr int TSTAGE = 2 * TK * TB * 2;   // P + Q tiles, bytes (32 KB)

OAT_DEV int tn_f(int m) { return ((m & 3) << 1) | (((m >> 3) & 1) << 3); }

struct TnArgs {
  const bf16* P; const bf16* Q;
  int M, N1, N2, ldp, ldq;
  float* slabs;            // [splits][N1][N2]
  int chunks_per_split;    // in units of TK rows
};

__global__ __launch_bounds__(256) void gemm_tn_kernel(TnArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w1 = wave >> 1, w2 = wave & 1;
  const int nt2 = (g.N2 + TB - 1) / TB;
  const int t1 = blockIdx.x / nt2, t2 = blockIdx.x % nt2;
  const int c1 = t1 * TB, c2 = t2 * TB;
  const int split = blockIdx.y;
  const int nchunks_total = (g.M + TK - 1) / TK;
  const int ch0 = split * g.chunks_per_split;
  const int ch1 = min(ch0 + g.chunks_per_split, nchunks_total);

  // staging: one wave instruction = 4 rows x 256 B; wave w stages rows [w*16, w*16+16)
  const int srow = lane >> 4, spc = lane & 15;
  int p_off[4], q_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = wave * 16 + i * 4 + srow;          // tile-local row
    const int lc = spc ^ tn_f(m);
    const int lp = min(lc, (g.N1 - c1) / 8 - 1);     // clamp ragged columns to a valid chunk
    const int lq = min(lc, (g.N2 - c2) / 8 - 1);
    p_off[i] = m * g.ldp + c1 + lp * 8;
    q_off[i] = m * g.ldq + c2 + lq * 8;
  }
  auto stage = [&](int buf, int chunk) {
    char* base = smem + buf * TSTAGE;
    const bf16* Pm = g.P + (size_t)chunk * TK * g.ldp;
    const bf16* Qm = g.Q + (size_t)chunk * TK * g.ldq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(Pm + p_off[i], base + (wave * 16 + i * 4) * 256);
      glds16(Qm + q_off[i], base + TK * 256 + (wave * 16 + i * 4) * 256);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transpose-read addressing: fetch lane s = lane & 15 of 16-lane group gq = lane >> 4
  const int s = lane & 15, gq = lane >> 4;
  const int rsub = s >> 2;                 // m offset 0..3 inside the 4-row block
  const int csub = s & 3;                  // 8-byte column group
  if (ch0 < ch1) stage(0, ch0);
  for (int ch = ch0; ch < ch1; ++ch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ch + 1 < ch1) stage((ch + 1 - ch0) & 1, ch + 1);
    char* sp = smem + ((ch - ch0) & 1) * TSTAGE;
    char* sq = sp + TK * 256;
    if (ch == nchunks_total - 1 && g.M - ch * TK < TK) {
      // ragged tail: rows >= M of the last chunk must not contribute (they are readable, not zero)
      const int valid = g.M - ch * TK;
      for (int idx = tid; idx < (TK - valid) * 16; idx += 256) {
        const int off = (valid + (idx >> 4)) * 256 + ((idx & 15) << 4);
        *reinterpret_cast<f32x4*>(sp + off) = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(sq + off) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      __syncthreads();
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 pf[4], qf[4];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int m = ks * 32 + gq * 8 + half * 4 + rsub;
        const int rowb = m * 256 + ((csub & 1) << 3);
        const int fm = tn_f(m);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int lc1 = ((w1 * 64 + i * 16) >> 3) + (csub >> 1);
          const int lc2 = ((w2 * 64 + i * 16) >> 3) + (csub >> 1);
          const s16x4 a = lds_tr16(sp + rowb + ((lc1 ^ fm) << 4));
          const s16x4 b = lds_tr16(sq + rowb + ((lc2 ^ fm) << 4));
          s16x4* pa = reinterpret_cast<s16x4*>(&pf[i]);
          s16x4* pb = reinterpret_cast<s16x4*>(&qf[i]);
          pa[half] = a;
          pb[half] = b;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[j], pf[i], acc[i][j], 0, 0, 0);
    }
  }

  // lane owns out[n1 = .. + (lane & 15)][n2 = .. + (lane >> 4) * 4 + 0..3]
  float* slab = g.slabs + (size_t)split * g.N1 * g.N2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = c1 + w1 * 64 + i * 16 + (lane & 15);
    if (r >= g.N1) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c2 + w2 * 64 + j * 16 + gq * 4;
      if (c >= g.N2) continue;
      *reinterpret_cast<f32x4*>(slab + (size_t)r * g.N2 + c) = acc[i][j];
    }
  }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slabs[s][i]     (i over N1*N2, vectorised x4)
__global__ void tn_reduce_kernel(const float* slabs, float* out, int n4, int splits, size_t stride4,
                                 int accumulate) {
  const f32x4* s4 = reinterpret_cast<const f32x4*>(slabs);
  f32x4* o4 = reinterpret_cast<f32x4*>(out);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    f32x4 v = accumulate ? o4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s) v += s4[(size_t)s * stride4 + i];
    o4[i] = v;
  }
}

}  // namespace oat

extern "C" size_t oat_gemm_tn_workspace_bytes(int M, int N1, int N2) {
  (void)M;
  // worst case 32 splits
  return (size_t)32 * N1 * N2 * sizeof(float);
}

extern "C" int oat_gemm_tn(const void* P, const void* Q, int M, int N1, int N2, int ldp, int ldq,
                           float* out, int accumulate, void* workspace, size_t workspace_bytes,
                           void* stream) {
  using namespace oat;
  if (M <= 0 || N1 <= 0 || N2 <= 0) { set_error("gemm_tn: empty problem"); return -1; }
  if (N1 % 8 || N2 % 8 || ldp % 8 || ldq % 8) { set_error("gemm_tn: N1,N2,ldp,ldq must be multiples of 8"); return -3; }
  if (!P || !Q || !out || !workspace) { set_error("gemm_tn: null pointer"); return -4; }
  const int tiles = ((N1 + TB - 1) / TB) * ((N2 + TB - 1) / TB);
  const int nchunks = (M + TK - 1) / TK;
  int splits = (768 + tiles - 1) / tiles;          // aim at ~3 workgroups per CU
  if (splits > 32) splits = 32;
  if (splits > nchunks) splits = nchunks;
  const int cps = (nchunks + splits - 1) / splits;
  splits = (nchunks + cps - 1) / cps;
  if ((size_t)splits * N1 * N2 * sizeof(float) > workspace_bytes) { set_error("gemm_tn: workspace too small"); return -6; }
  TnArgs g{(const bf16*)P, (const bf16*)Q, M, N1, N2, ldp, ldq, (float*)workspace, cps};
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TSTAGE);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits), dim3(256), 2 * TSTAGE, s, g);
  int rc = check_launch("gemm_tn");
  if (rc) return rc;
  const int n4 = N1 * N2 / 4;
  int blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(tn_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)workspace, out, n4,
                     splits, (size_t)N1 * N2 / 4, accumulate);
  return check_launch("gemm_tn_reduce");
}
