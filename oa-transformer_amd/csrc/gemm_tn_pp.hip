// bf16 MFMA GEMM, "TN" form (weight gradients), two-wave-group ("ping-pong") 256x256 kernel:
//     slab[split][N1,N2] = sum over the split's rows m of P[m,N1]^T * Q[m,N2]      (+ bias column sums of P)
//
// Same contract and slab format as gemm_tn_kernel<2,4,8,4> (gemm_tn.hip; dW = dY^T X and db = colsum(dY) of every
// nn.Linear of the reference, video_transformer.py:46-50,102,133 in backward); what differs is the loop over m.  See
// gemm_nt_pp.hip for the schedule: waves 0-3 / 4-7 run the same stream one barrier interval apart, so that on every
// SIMD one wave issues MFMAs while the other one reads LDS and issues LDS-DMA; a "K-tile" is 64 rows of m,
// 4 quadrants x 16 MFMAs per wave; quadrant order (a0,b0) (a1,b0) (a1,b1) (a0,b1); region b0 | a0 | a1 | b1 of a
// K-tile buffer is refilled with K-tile s + 2 in L1 | L2 | L3 | L4 of K-tile s; every wait is `s_waitcnt vmcnt(12)`.
//
// Operands have the reduction index m as their ROW index, so fragments come from ds_read_b64_tr_b16 transpose reads
// of row-major LDS images.  For the region-wise refill the 256 tile columns of an operand are split into two CLASSES of
// 128 columns ([64 m][128 cols] = 256-byte rows, 16 KB): class a of P holds the columns a*128 + wm*64 + [0,64) of the
// two wave rows, class b of Q the columns b*128 + wn*32 + [0,32) of the four wave columns.  One LDS-DMA piece = 4 rows
// x 256 B (two full cache lines per row).  16-byte chunk c of row m sits at c ^ tn_f(m) (tn_f as in gemm_tn.hip: the 8
// rows a 32-lane half of a transpose read touches land in 8 different 32-byte bank groups).
//
// Rows: a split covers `chunks_per_split` K-tiles of 64 rows (even, so the steady loop runs whole pairs); what is
// left in the LAST split - an odd full K-tile and / or the ragged one (M % 64 rows, zeroed in LDS beyond M) - is
// computed by a plain, non-staggered tail after the loop from the data the stream already staged.
// Contract (unchanged): rows [M, round_up(M,64)) of P and Q must be READABLE; their contents are ignored.
#include "gemm.h"
#include <type_traits>

namespace oat {

namespace {

constexpr int TP_BUF = 32768, TP_CLS = 16384, TP_Q = 65536;
constexpr int TP_LDS = 131072;
enum : int { TPF_PRIO = 1, TPF_NOSTAGGER = 2, TPF_LGKM = 4, TPF_NOSTORE = 8 };

OAT_DEV int tp_f(int m) { return ((m & 3) << 1) | (((m >> 3) & 1) << 3); }

template <int FL>
__global__ __launch_bounds__(512) void gemm_tn_pp_kernel(TnArgs g) {
  constexpr bool PRIO = FL & TPF_PRIO, STAGGER = !(FL & TPF_NOSTAGGER), LGKM = FL & TPF_LGKM, NOSTORE = FL & TPF_NOSTORE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // work item: XCD-contiguous, split-major (all tiles of one row range are neighbours on one L2)
  const int nt2 = g.N2 >> 8;
  const int ntiles = (g.N1 >> 8) * nt2;
  int wid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = wid & 7, idx = wid >> 3;
    wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = wid / ntiles, tile = wid - split * ntiles;
  const int t1 = tile / nt2, t2 = tile - t1 * nt2;
  const int c1 = t1 << 8, c2 = t2 << 8;
  const int nkt_total = (g.M + 63) >> 6;
  const int kt0 = split * g.chunks_per_split;
  const int n = min(kt0 + g.chunks_per_split, nkt_total) - kt0;          // K-tiles of this split (>= 1)
  const bool ragged = (kt0 + n == nkt_total) && (g.M & 63) != 0;
  const int npairs = (n - (ragged ? 1 : 0)) >> 1;
  const int ntail = n - 2 * npairs;                                      // 0..2 K-tiles for the tail path

  // ---- staging: wave w stages pieces 2w, 2w+1 (rows 8w .. 8w+7) of each class region
  const int prow = lane >> 4;
  const uint32_t c16 = (uint32_t)(((lane & 15) ^ ((prow << 1) | ((wave & 1) << 3))) << 4);
  uint32_t poff[2], qoff[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const uint32_t row = (uint32_t)(4 * (2 * wave + e) + prow);
    poff[e] = row * (uint32_t)g.ldp * 2u + c16;
    qoff[e] = row * (uint32_t)g.ldq * 2u + c16;
  }
  const bf16* cp = g.P + (size_t)kt0 * 64 * g.ldp + c1;
  const bf16* cq = g.Q + (size_t)kt0 * 64 * g.ldq + c2;
  const size_t pstep = (size_t)64 * g.ldp, qstep = (size_t)64 * g.ldq;
  int cnext = 0;                      // K-tile the cursor points at
  uint32_t dmask = ~0u;               // 0xff once the split's rows are exhausted: the pieces degenerate to re-reads of
                                      // one 256-byte line into regions nobody reads any more (uniform op stream)
  // The ragged last K-tile (M % 64 rows) is zeroed in LDS beyond row M - 1 after it lands; its rows past M - 1 must not be FETCHED from
  // beyond the operand (a row slice may end right there: see gemm_tn.hip `stage`): once the cursor stands on that tile the per-lane
  // row offsets are clamped to its last valid row (they are not needed unclamped again: it is the last tile of the walk).
  auto clamp_last = [&]() __attribute__((always_inline)) {
    const uint32_t vlast = (uint32_t)((g.M & 63) - 1);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const uint32_t row = (uint32_t)(4 * (2 * wave + e) + prow), rc = row < vlast ? row : vlast;
      poff[e] = rc * (uint32_t)g.ldp * 2u + c16;
      qoff[e] = rc * (uint32_t)g.ldq * 2u + c16;
    }
  };
  if (ragged && n == 1) clamp_last();
  auto advance = [&]() __attribute__((always_inline)) {
    ++cnext;
    const bool more = cnext < n;
    cp = more ? cp + pstep : cp;
    cq = more ? cq + qstep : cq;
    dmask = more ? dmask : 0xffu;
    if (ragged && cnext == n - 1) clamp_last();          // wave-uniform
  };
  auto stageP = [&](auto cls, int buf) __attribute__((always_inline)) {
    constexpr int A = decltype(cls)::value;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      glds16_asm_lds(cp + A * 128, poff[e] & dmask, lds0 + buf * TP_BUF + A * TP_CLS + (2 * wave + e) * 1024);
  };
  auto stageQ = [&](auto cls, int buf) __attribute__((always_inline)) {
    constexpr int B = decltype(cls)::value;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      glds16_asm_lds(cq + B * 128, qoff[e] & dmask, lds0 + TP_Q + buf * TP_BUF + B * TP_CLS + (2 * wave + e) * 1024);
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;

  // ---- transpose-read addresses.  16-lane group gq = lane >> 4 covers k = gq*8 .. +8 of a 32-row k-step; inside it
  // fetch lane s = lane & 15 reads 4 bf16 of row (s >> 2) [+4 for the second half] at column group s & 3.
  const int sl = lane & 15, gq = lane >> 4;
  const int rsub = sl >> 2, csub = sl & 3;
  const int fx = (rsub << 1) | ((gq & 1) << 3);                          // tp_f of every row this lane reads
  uint32_t pa[4], pb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    pa[i] = lds0 + (gq * 8 + rsub) * 256 + (((wm * 8 + i * 2 + (csub >> 1)) ^ fx) << 4) + ((csub & 1) << 3);
#pragma unroll
  for (int j = 0; j < 2; ++j)
    pb[j] = lds0 + TP_Q + (gq * 8 + rsub) * 256 + (((wn * 4 + j * 2 + (csub >> 1)) ^ fx) << 4) + ((csub & 1) << 3);
  typedef __attribute__((address_space(3))) s16x4* lds_tr;
  auto tr = [&](uint32_t addr) __attribute__((always_inline)) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(uintptr_t)addr);
  };
  auto readA = [&](bf16x8 (&f)[2][4], int cls, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          reinterpret_cast<s16x4*>(&f[kk][i])[h] = tr(pa[i] + buf * TP_BUF + cls * TP_CLS + kk * 8192 + h * 1024);
  };
  auto readB = [&](bf16x8 (&f)[2][2], int cls, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          reinterpret_cast<s16x4*>(&f[kk][j])[h] = tr(pb[j] + buf * TP_BUF + cls * TP_CLS + kk * 8192 + h * 1024);
  };

  f32x4 acc[8][4], accb[2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  accb[0] = accb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias column sums ride on the matrix pipe (all-ones A operand).  The 8 P fragments of a wave row are spread over the
  // nt2 x 4 waves that hold them (workgroups of the same t1, waves of the same wm): at most one per class and wave.
  int own[2] = {-1, -1};
  if (g.bias_slabs != nullptr) {
    const int owners = nt2 * 4, me = t2 * 4 + wn;
#pragma unroll
    for (int f = 0; f < 8; ++f)
      if (f % owners == me) own[f >> 2] = f & 3;
  }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  auto endL = [&]() __attribute__((always_inline)) {
    if constexpr (LGKM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto quad = [&](const bf16x8 (&fa)[2][4], const bf16x8 (&fb)[2][2], int ah, int bh) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[ah * 4 + i][bh * 2 + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
    if (bh == 0 && own[ah] >= 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (own[ah] == i) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
            accb[ah] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[kk][i], accb[ah], 0, 0, 0);
        }
    }
  };
  auto mma = [&](const bf16x8 (&fa)[2][4], const bf16x8 (&fb)[2][2], int ah, int bh) __attribute__((always_inline)) {
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    quad(fa, fb, ah, bh);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: K-tiles 0 and 1 in canonical issue order (b0 a0 a1 b1), then b0(0) into registers
  stageQ(C0{}, 0); stageP(C0{}, 0); stageP(C1{}, 0); stageQ(C1{}, 0);
  advance();
  stageQ(C0{}, 1); stageP(C0{}, 1); stageP(C1{}, 1); stageQ(C1{}, 1);
  advance();
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8 fbx[2][2];
  readB(fbx, 0, 0);
  if (STAGGER && wm == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one interval behind from here on
  __builtin_amdgcn_sched_barrier(0);

  for (int p = 0; p < npairs; ++p) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bf16x8 fa0[2][4], fa1[2][4], fby[2][2];
      readA(fa0, 0, u); stageQ(C0{}, u); endL();
      mma(fa0, fbx, 0, 0);
      readA(fa1, 1, u); stageP(C0{}, u); endL();
      mma(fa1, fbx, 1, 0);
      readB(fby, 1, u); stageP(C1{}, u); endL();
      mma(fa1, fby, 1, 1);
      readB(fbx, 0, u ^ 1); stageQ(C1{}, u); endL();
      mma(fa0, fby, 0, 1);
      advance();
    }
  }
  if (STAGGER && wm == 0) __builtin_amdgcn_s_barrier();  // pairs with group 1's extra barrier: lockstep again
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // everything staged has landed (no LDS-DMA outlives the loop)
  __syncthreads();

  // ---- tail: up to two K-tiles (buffer t), plain schedule
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t < ntail) {
      if (ragged && t == ntail - 1) {
        const int valid = g.M & 63;
        for (int idx = tid; idx < (64 - valid) * 64; idx += 512) {       // 4 regions x 16 chunks per row
          const int row = valid + (idx >> 6), reg = (idx >> 4) & 3, ch = idx & 15;
          *reinterpret_cast<f32x4*>(smem + (reg >> 1) * TP_Q + t * TP_BUF + (reg & 1) * TP_CLS + row * 256 + ch * 16) =
              f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
      }
      bf16x8 fa0[2][4], fa1[2][4], fb0[2][2], fb1[2][2];
      readA(fa0, 0, t); readA(fa1, 1, t); readB(fb0, 0, t); readB(fb1, 1, t);
      quad(fa0, fb0, 0, 0); quad(fa1, fb0, 1, 0); quad(fa1, fb1, 1, 1); quad(fa0, fb1, 0, 1);
    }
  }

  // ---- slab store: lane owns out[n1 = .. + (lane & 15)][n2 = .. + (lane >> 4) * 4 + 0..3]
  float* slab = g.slabs + (size_t)split * g.N1 * g.N2;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = c1 + a * 128 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = c2 + b * 128 + wn * 32 + j * 16 + gq * 4;
          if constexpr (NOSTORE) asm volatile("" ::"v"(acc[a * 4 + i][b * 2 + j]));
          else *reinterpret_cast<f32x4*>(slab + (size_t)r * g.N2 + c) = acc[a * 4 + i][b * 2 + j];
        }
    }
    if (own[a] >= 0 && gq == 0)
      g.bias_slabs[(size_t)split * g.N1 + c1 + a * 128 + wm * 64 + own[a] * 16 + (lane & 15)] = accb[a][0];
  }
}

template <int FL>
int launch_tn_pp_cfg(const TnArgs& g, hipStream_t s) {
  OAT_MAX_LDS(gemm_tn_pp_kernel<FL>, TP_LDS);
  const int tiles = (g.N1 / 256) * (g.N2 / 256);
  OAT_LAUNCH((gemm_tn_pp_kernel<FL>), dim3(tiles * g.splits), dim3(512), TP_LDS, s, g);
  return check_launch("gemm_tn_pp");
}

}  // namespace

int launch_tn_pp(const TnArgs& g, int flags, hipStream_t s) {
  constexpr int DEF = TPF_PRIO | TPF_LGKM;
  switch (DEF ^ flags) {
    case DEF: return launch_tn_pp_cfg<DEF>(g, s);
    case DEF | TPF_NOSTAGGER: return launch_tn_pp_cfg<DEF | TPF_NOSTAGGER>(g, s);
    case DEF | TPF_NOSTORE: return launch_tn_pp_cfg<DEF | TPF_NOSTORE>(g, s);
    default: set_error("gemm_tn_pp: flag combination not built"); return -7;
  }
}

}  // namespace oat
