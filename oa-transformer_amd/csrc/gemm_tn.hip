// bf16 MFMA GEMM, "TN" form (weight gradients):  out[N1,N2] = sum_m P[m,N1]^T * Q[m,N2]
//                               and optionally     bias[N1]   = sum_m P[m,N1]
//
// This is the reduction-over-rows product autograd runs for every nn.Linear weight of the
// reference (dW = dY^T X, db = colsum(dY); video_transformer.py:102,133,46-50 in backward).
// Both operands have the reduction index m as their SLOW (row) index, so MFMA fragments (8
// consecutive k per lane) are column slices of the LDS tile: gfx950's ds_read_b64_tr_b16
// transpose-read delivers them straight from the row-major image - no transposed copies of
// activations ever touch HBM.
//
// Tile <WM,WN,TM,TN>: (WM*TM*16) x (WN*TN*16) outputs, WM*WN waves, 64 rows of m per stage,
// global_load_lds staging (double buffered), XOR chunk swizzle on the source address + read.
//   <2,4,8,4> = 256x256 / 8 waves / 128 KB LDS (big-M weight gradients, half the L2->LDS bytes per
//   FLOP of the 128^2 tile); <2,2,4,4> = 128x128 / 4 waves / 64 KB (small problems).
// The m range is split across grid.y; partial products go to fp32 slabs and a deterministic reduce
// kernel sums them (no atomics).  The bias column sums ride on the matrix pipe: one extra MFMA per
// (n1-tile, k-step) against an all-ones operand in the workgroups of the first N2 tile.
// Contract: rows [M, round_up(M,64)) of P and Q must be READABLE (inside the allocation); their
// contents are ignored (the ragged tail of the last chunk is zeroed in LDS).
#include "gemm.h"

namespace oat {

constexpr int TK = 32;             // m rows per stage (one MFMA k-step)
constexpr int NS = 4;              // LDS ring depth: loads run NS-1 stages ahead of the math

OAT_DEV int tn_f(int m) { return ((m & 3) << 1) | (((m >> 3) & 1) << 3); }

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM * WN * 64) void gemm_tn_kernel(TnArgs g) {
  constexpr int B1 = WM * TM * 16, B2 = WN * TN * 16, NW = WM * WN, NT = NW * 64;
  constexpr int P_BYTES = TK * B1 * 2, Q_BYTES = TK * B2 * 2, STAGE = P_BYTES + Q_BYTES;
  constexpr int CP = B1 / 8, CQ = B2 / 8;                  // 16-byte chunks per tile row
  constexpr int GP = TK * CP / 64 / NW, GQ = TK * CQ / 64 / NW;   // glds instructions per wave per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w1 = wave / WN, w2 = wave % WN;
  // XCD-aware decomposition: hardware places block b on XCD b % 8.  Work items are ordered
  // split-major (all tiles of one m-split are neighbours: they read the SAME rows of P and Q) and
  // each XCD takes a contiguous run of that order, so a split's operand rows are fetched from HBM
  // into one (at most two) XCD L2s and reused by every tile there.  Pure speed choice.
  const int nt2 = (g.N2 + B2 - 1) / B2;
  const int ntiles = ((g.N1 + B1 - 1) / B1) * nt2;
  int wid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = wid & 7, idx = wid >> 3;
    wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective remap
  }
  const int split = wid / ntiles, tile = wid % ntiles;
  const int t1 = tile / nt2, t2 = tile % nt2;
  const int c1 = t1 * B1, c2 = t2 * B2;
  const int nchunks_total = (g.M + TK - 1) / TK;
  const int ch0 = split * g.chunks_per_split;
  const int ch1 = min(ch0 + g.chunks_per_split, nchunks_total);
  // bias column sums (an all-ones MFMA operand) are spread over the nt2 workgroups and WN waves that hold the
  // same P tile, so that no wave carries more than ~2 extra MFMAs per chunk (all 8 on the t2 = 0, w2 = 0 waves
  // made those workgroups - and with them the launch - 10 % slower)
  unsigned bias_own = 0;                     // bit i: this wave sums P tile (w1, i)
  if (g.bias_slabs != nullptr) {
    const int owners = nt2 * WN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
      if ((w1 * TM + i) % owners == t2 * WN + w2) bias_own |= 1u << i;
  }

  // staging: one wave instruction moves 64 chunks = 64/CP rows of the P tile (64/CQ of the Q tile)
  int p_off[GP], q_off[GQ];
#pragma unroll
  for (int i = 0; i < GP; ++i) {
    const int idx = (wave * GP + i) * 64 + lane;
    const int m = idx / CP, pc = idx % CP;
    const int lc = min(pc ^ tn_f(m), (g.N1 - c1) / 8 - 1);      // inverse swizzle; clamp ragged columns
    p_off[i] = m * g.ldp + c1 + lc * 8;
  }
#pragma unroll
  for (int i = 0; i < GQ; ++i) {
    const int idx = (wave * GQ + i) * 64 + lane;
    const int m = idx / CQ, pc = idx % CQ;
    const int lc = min(pc ^ tn_f(m), (g.N2 - c2) / 8 - 1);
    q_off[i] = m * g.ldq + c2 + lc * 8;
  }
  auto stage = [&](int buf, int chunk) {
    char* base = smem + buf * STAGE;
    const bf16* Pm = g.P + (size_t)chunk * TK * g.ldp;
    const bf16* Qm = g.Q + (size_t)chunk * TK * g.ldq;
    // The last chunk of a ragged M holds fewer than TK rows.  Its missing rows are zeroed in LDS after they land, but they must not be
    // FETCHED from beyond row M - 1: the operands of the pruned top block are row slices that start at a clip's CLS rows (3 ... 64 rows
    // before the end of an [Mp, .] buffer), and up to TK - 1 rows past M then lie past the end of the allocation - an illegal access
    // whenever the allocator placed the tensor at the end of a mapping (seen once in four runs of the GPU suite, round 5).  Rows past
    // M - 1 re-read row M - 1 instead.
    const int valid = g.M - chunk * TK;                     // workgroup-uniform
    if (valid >= TK) {
#pragma unroll
      for (int i = 0; i < GP; ++i) glds16_asm_so(Pm, (uint32_t)p_off[i] * 2u, base + (wave * GP + i) * 1024);
#pragma unroll
      for (int i = 0; i < GQ; ++i) glds16_asm_so(Qm, (uint32_t)q_off[i] * 2u, base + P_BYTES + (wave * GQ + i) * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < GP; ++i) {
        const int m = ((wave * GP + i) * 64 + lane) / CP, over = m - min(m, valid - 1);
        glds16_asm_so(Pm, (uint32_t)(p_off[i] - over * g.ldp) * 2u, base + (wave * GP + i) * 1024);
      }
#pragma unroll
      for (int i = 0; i < GQ; ++i) {
        const int m = ((wave * GQ + i) * 64 + lane) / CQ, over = m - min(m, valid - 1);
        glds16_asm_so(Qm, (uint32_t)(q_off[i] - over * g.ldq) * 2u, base + P_BYTES + (wave * GQ + i) * 1024);
      }
    }
  };

  f32x4 acc[TM][TN], accb[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  // transpose-read addressing: fetch lane s = lane & 15 of 16-lane group gq = lane >> 4
  const int s = lane & 15, gq = lane >> 4;
  const int rsub = s >> 2;                 // m offset 0..3 inside the 4-row block
  const int csub = s & 3;                  // 8-byte column group
  // Ring pipeline: the loads of NS-1 stages are in flight while one stage is consumed.  A stage is
  // published by (own counted vmcnt) + (raw s_barrier): every wave waits for ITS loads of stage i, the
  // barrier then makes all of them visible and also proves everyone finished reading stage i-1, whose
  // slot the next prefetch (stage i+NS-1) overwrites.  __syncthreads() is avoided on purpose: with an
  // LDS-DMA in flight the compiler turns it into vmcnt(0) and drains the pipeline.
  constexpr int LPS = GP + GQ;                               // glds instructions per thread per stage
#pragma unroll
  for (int d = 0; d < NS - 1; ++d)
    if (ch0 + d < ch1) stage(d, ch0 + d);
  for (int ch = ch0; ch < ch1; ++ch) {
    const int ahead = min(NS - 2, ch1 - 1 - ch);             // stages issued after this one
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ch + NS - 1 < ch1) stage((ch + NS - 1 - ch0) % NS, ch + NS - 1);
    char* sp = smem + ((ch - ch0) % NS) * STAGE;
    char* sq = sp + P_BYTES;
    if (ch == nchunks_total - 1 && g.M - ch * TK < TK) {
      // ragged tail: rows >= M of the last chunk must not contribute (they are readable, not zero)
      const int valid = g.M - ch * TK;
      for (int idx = tid; idx < (TK - valid) * CP; idx += NT)
        *reinterpret_cast<f32x4*>(sp + (valid + idx / CP) * (CP * 16) + (idx % CP) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int idx = tid; idx < (TK - valid) * CQ; idx += NT)
        *reinterpret_cast<f32x4*>(sq + (valid + idx / CQ) * (CQ * 16) + (idx % CQ) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
      __syncthreads();
    }
    {
      constexpr int ks = 0;
      bf16x8 qf[TN];
      const int sub = (csub & 1) << 3;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int m = ks * 32 + gq * 8 + half * 4 + rsub;
        const int fm = tn_f(m);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int lc = ((w2 * TN * 16 + j * 16) >> 3) + (csub >> 1);
          reinterpret_cast<s16x4*>(&qf[j])[half] = lds_tr16(sq + m * (CQ * 16) + ((lc ^ fm) << 4) + sub);
        }
      }
      // P fragments in groups of 4 tiles (keeps the live fragment set small)
#pragma unroll
      for (int i0 = 0; i0 < TM; i0 += 4) {
        bf16x8 pf[4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int m = ks * 32 + gq * 8 + half * 4 + rsub;
          const int fm = tn_f(m);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int lc = ((w1 * TM * 16 + (i0 + i) * 16) >> 3) + (csub >> 1);
            reinterpret_cast<s16x4*>(&pf[i])[half] = lds_tr16(sp + m * (CP * 16) + ((lc ^ fm) << 4) + sub);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[j], pf[i], acc[i0 + i][j], 0, 0, 0);
        if (bias_own) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (bias_own >> (i0 + i) & 1)
              accb[i0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[i], accb[i0 + i], 0, 0, 0);
        }
      }
    }
  }

  // lane owns out[n1 = .. + (lane & 15)][n2 = .. + (lane >> 4) * 4 + 0..3]
  float* slab = g.slabs + (size_t)split * g.N1 * g.N2;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = c1 + w1 * TM * 16 + i * 16 + (lane & 15);
    if (r >= g.N1) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = c2 + w2 * TN * 16 + j * 16 + gq * 4;
      if (c >= g.N2) continue;
      *reinterpret_cast<f32x4*>(slab + (size_t)r * g.N2 + c) = acc[i][j];
    }
    if ((bias_own >> i & 1) && gq == 0) g.bias_slabs[(size_t)split * g.N1 + r] = accb[i][0];
  }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slabs[s][i] (i over n4 float4s) and, in the same launch, the same
// reduction of the bias slabs (nb4 float4s; the items after n4)
__global__ void tn_reduce_kernel(const float* slabs, float* out, int n4, int splits, size_t stride4,
                                 const float* bslabs, float* bout, int nb4, int accumulate) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4 + nb4; i += gridDim.x * blockDim.x) {
    const bool isb = i >= n4;
    const int k = isb ? i - n4 : i;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(isb ? bslabs : slabs);
    f32x4* o4 = reinterpret_cast<f32x4*>(isb ? bout : out);
    const size_t st = isb ? (size_t)nb4 : stride4;
    f32x4 v = accumulate ? o4[k] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s) v += s4[(size_t)s * st + k];
    o4[k] = v;
  }
}

// tile choice of a launch (the `tune` argument of oat_gemm_tn, per call): 0 auto, 1 force 128^2, 2 force the lockstep 256^2, 4 force ping-pong
static int g_tn_pp_default = 1;   // auto prefers the ping-pong kernel (gemm_tn_pp.hip) for 256^2 launches

static int reduce_slabs(const TnArgs& g, int splits, float* out, float* bias_out, int accumulate, hipStream_t s) {
  const int n4 = g.N1 * g.N2 / 4, nb4 = bias_out ? g.N1 / 4 : 0;
  int blocks = (n4 + nb4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  OAT_LAUNCH(tn_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)g.slabs, out, n4, splits,
                     (size_t)g.N1 * g.N2 / 4, (const float*)g.bias_slabs, bias_out, nb4, accumulate);
  return check_launch("gemm_tn_reduce");
}

template <int WM, int WN, int TM, int TN>
static int launch_tn(TnArgs g, int splits, float* out, float* bias_out, int accumulate, hipStream_t s) {
  constexpr int B1 = WM * TM * 16, B2 = WN * TN * 16;
  constexpr int LDS = NS * TK * (B1 + B2) * 2;
  const int tiles = ((g.N1 + B1 - 1) / B1) * ((g.N2 + B2 - 1) / B2);
  OAT_MAX_LDS((gemm_tn_kernel<WM, WN, TM, TN>), LDS);
  OAT_LAUNCH((gemm_tn_kernel<WM, WN, TM, TN>), dim3(tiles * splits), dim3(WM * WN * 64), LDS, s, g);
  int rc = check_launch("gemm_tn");
  if (rc) return rc;
  return reduce_slabs(g, splits, out, bias_out, accumulate, s);
}

}  // namespace oat

// tile shape and split count of a launch (shared by the launcher and the workspace query)
static void tn_plan(int M, int N1, int N2, int variant, bool* big_, int* splits_, int* cps_, bool* pp_ = nullptr) {
  using namespace oat;
  const bool fits = N1 % 256 == 0 && N2 % 256 == 0;
  const bool big = variant == 2 || (variant == 4 && fits) || (variant == 0 && M >= 4096 && fits);
  const bool pp = big && fits && (variant == 4 || (variant == 0 && g_tn_pp_default));
  const int B = big ? 256 : 128;
  const int tiles = ((N1 + B - 1) / B) * ((N2 + B - 1) / B);
  const int unit = pp ? 64 : TK;                    // rows per chunk (ping-pong: K-tiles of 64 rows, an even number per split)
  const int nchunks = (M + unit - 1) / unit;
  const int slots = big ? 256 : 512;   // one 8-wave workgroup per CU, or two 4-wave ones
  int splits = slots / tiles;                       // largest split count that still fits one round
  if (splits < 1) splits = 1;
  if (splits > 32) splits = 32;
  if (splits > nchunks) splits = nchunks;
  int cps = (nchunks + splits - 1) / splits;
  if (pp) cps += cps & 1;
  splits = (nchunks + cps - 1) / cps;
  *big_ = big; *splits_ = splits; *cps_ = cps;
  if (pp_) *pp_ = pp;
}

// M > 0: exact need of that launch (with the same `tune`); M <= 0: worst case over every M (32 splits)
extern "C" size_t oat_gemm_tn_workspace_bytes(int M, int N1, int N2, int tune) {
  int splits = 32, cps = 0;
  bool big = false;
  if (M > 0) tn_plan(M, N1, N2, tune & 0xff, &big, &splits, &cps);
  return (size_t)splits * ((size_t)N1 * N2 + N1) * sizeof(float);      // weight slabs + bias slabs
}

// out[N1,N2] (+)= P^T Q ; bias_out[N1] (+)= column sums of P (NULL to skip)
extern "C" int oat_gemm_tn(const void* P, const void* Q, int M, int N1, int N2, int ldp, int ldq,
                           float* out, float* bias_out, int accumulate, void* workspace, size_t workspace_bytes,
                           int tune, void* stream) {
  using namespace oat;
  const int variant = tune & 0xff;
  if (variant != 0 && variant != 1 && variant != 2 && variant != 4) { set_error("gemm_tn: unknown tile choice in `tune`"); return -3; }
  if (M <= 0 || N1 <= 0 || N2 <= 0) { set_error("gemm_tn: empty problem"); return -1; }
  if (N1 % 8 || N2 % 8 || ldp % 8 || ldq % 8) { set_error("gemm_tn: N1,N2,ldp,ldq must be multiples of 8"); return -3; }
  if (!P || !Q || !out || !workspace) { set_error("gemm_tn: null pointer"); return -4; }
  bool big, pp; int splits, cps;
  tn_plan(M, N1, N2, variant, &big, &splits, &cps, &pp);
  const size_t need = (size_t)splits * ((size_t)N1 * N2 + N1) * sizeof(float);
  if (need > workspace_bytes) { set_error("gemm_tn: workspace too small"); return -6; }
  float* slabs = (float*)workspace;
  TnArgs g{(const bf16*)P, (const bf16*)Q, M, N1, N2, ldp, ldq, slabs,
           bias_out ? slabs + (size_t)splits * N1 * N2 : nullptr, cps, splits, 0};
  hipStream_t s = (hipStream_t)stream;
  if (pp) {
    int rc = launch_tn_pp(g, 0, s);
    if (rc) return rc;
    return reduce_slabs(g, splits, out, bias_out, accumulate, s);
  }
  if (big) return launch_tn<2, 4, 8, 4>(g, splits, out, bias_out, accumulate, s);
  return launch_tn<2, 2, 4, 4>(g, splits, out, bias_out, accumulate, s);
}
