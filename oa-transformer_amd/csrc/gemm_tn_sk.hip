// Grouped, persistent "stream-K" form of the ping-pong weight-gradient GEMM (gemm_tn_pp.hip):
//     out_p[N1,N2] (+)= P_p[:M]^T Q_p[:M]      bias_p[N1] (+)= column sums of P_p        for a LIST of problems p
// in ONE launch of one workgroup per CU.  Replaces, per ViT block, the six `gemm_tn + tn_reduce` pairs of
// engine/video.py:_block_bwd (dW = dY^T X, db = colsum(dY) of attn.qkv / attn.proj / timeattn.qkv / timeattn.proj /
// mlp.fc1 / mlp.fc2: the autograd of the nn.Linear calls at /root/reference/OATrans/model/video_transformer.py:46-50,102,133)
// and, per DistilBERT pass, the 36 small ones of engine/text.py.
//
// Why.  A single weight gradient has 9-36 output tiles of 256x256, so gemm_tn_pp splits the reduction (M = 50 k rows) 7-28
// ways to fill 256 CUs and every launch writes 252 fp32 partial tiles (64.5 MB) that tn_reduce reads back: ~15 % of the
// launch, 146 launches and kernel boundaries per step.  Here the unit of work is a PAIR of K-tiles (128 rows) of one
// output tile; the units of all tiles of all problems form one sequence that is cut into `grid` equal, contiguous
// shares.  A workgroup walks its share segment by segment (segment = a row range of one tile): tiles that fall entirely
// inside a share are written straight to the gradient (no slab), only the tiles a share boundary cuts leave partial
// tiles (<= 2 per workgroup) that one fix-up launch sums in a fixed order (deterministic).  Six weight gradients: one
// GEMM launch + one fix-up, ~130 MB of partial tiles instead of 6 x 64.5 MB.
//
// The K loop, LDS layout, transpose reads, bias sums on the matrix pipe and the ragged tail are those of
// gemm_tn_pp_kernel, run once per segment (prologue - pairs - tail - store).  Host side: oat_tn_group_plan cuts the
// unit sequence (plain C, no GPU), the caller keeps the tables in device memory (static per shape set).
#include "gemm.h"
#include <type_traits>
#include <vector>

namespace oat {

struct SkProblem {          // 64 bytes, same layout on host and device (include/oatrans_hip.h: OatTnProblem)
  const bf16* P; const bf16* Q; float* out; float* bias_out;
  int M, N1, N2, ldp, ldq, accumulate, pad0, pad1;
};
struct SkSeg {              // one row range of one output tile (OatTnSeg)
  int prob, c1, c2, t2;     // problem, first output row / column of the tile, column-tile index
  int kt0, n;               // first K-tile (64 rows), number of K-tiles (>= 1)
  int slot;                 // partial tile: index of its slab; -1: the segment covers the whole tile -> direct store
  int last;                 // the segment ends with the tile's last K-tile (which may be ragged: M % 64 rows)
};
struct SkFix {              // one split tile (OatTnFix): out tile = sum of slabs [slot0, slot0 + nslots)
  int prob, c1, c2, t2, slot0, nslots, pad0, pad1;
};

namespace {

constexpr int SK_BUF = 32768, SK_CLS = 16384, SK_Q = 65536;
constexpr int SK_LDS = 131072;
constexpr int SK_SLAB = 256 * 256 + 256;        // floats per slab: the tile, then its 256 bias columns

__global__ __launch_bounds__(512) void gemm_tn_sk_kernel(const SkProblem* __restrict__ probs, const SkSeg* __restrict__ segs,
                                                         const int* __restrict__ seg_off, float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // ---- lane constants of the staging and of the transpose reads (gemm_tn_pp.hip)
  const int prow = lane >> 4;
  const uint32_t c16 = (uint32_t)(((lane & 15) ^ ((prow << 1) | ((wave & 1) << 3))) << 4);
  const int sl = lane & 15, gq = lane >> 4;
  const int rsub = sl >> 2, csub = sl & 3;
  const int fx = (rsub << 1) | ((gq & 1) << 3);
  uint32_t pa[4], pb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    pa[i] = lds0 + (gq * 8 + rsub) * 256 + (((wm * 8 + i * 2 + (csub >> 1)) ^ fx) << 4) + ((csub & 1) << 3);
#pragma unroll
  for (int j = 0; j < 2; ++j)
    pb[j] = lds0 + SK_Q + (gq * 8 + rsub) * 256 + (((wn * 4 + j * 2 + (csub >> 1)) ^ fx) << 4) + ((csub & 1) << 3);
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
          reinterpret_cast<s16x4*>(&f[kk][i])[h] = tr(pa[i] + buf * SK_BUF + cls * SK_CLS + kk * 8192 + h * 1024);
  };
  auto readB = [&](bf16x8 (&f)[2][2], int cls, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          reinterpret_cast<s16x4*>(&f[kk][j])[h] = tr(pb[j] + buf * SK_BUF + cls * SK_CLS + kk * 8192 + h * 1024);
  };
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;

  const int s_begin = seg_off[blockIdx.x], s_end = seg_off[blockIdx.x + 1];
  for (int si = s_begin; si < s_end; ++si) {
    const SkSeg sg = segs[si];
    const SkProblem pr = probs[sg.prob];
    const int c1 = sg.c1, c2 = sg.c2, n = sg.n;
    const bool ragged = sg.last != 0 && (pr.M & 63) != 0;
    const int npairs = (n - (ragged ? 1 : 0)) >> 1;
    const int ntail = n - 2 * npairs;                                      // 0..2 K-tiles for the tail path

    uint32_t poff[2], qoff[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const uint32_t row = (uint32_t)(4 * (2 * wave + e) + prow);
      poff[e] = row * (uint32_t)pr.ldp * 2u + c16;
      qoff[e] = row * (uint32_t)pr.ldq * 2u + c16;
    }
    const bf16* cp = pr.P + (size_t)sg.kt0 * 64 * pr.ldp + c1;
    const bf16* cq = pr.Q + (size_t)sg.kt0 * 64 * pr.ldq + c2;
    const size_t pstep = (size_t)64 * pr.ldp, qstep = (size_t)64 * pr.ldq;
    int cnext = 0;
    uint32_t dmask = ~0u;
    auto advance = [&]() __attribute__((always_inline)) {
      ++cnext;
      const bool more = cnext < n;
      cp = more ? cp + pstep : cp;
      cq = more ? cq + qstep : cq;
      dmask = more ? dmask : 0xffu;
    };
    auto stageP = [&](auto cls, int buf) __attribute__((always_inline)) {
      constexpr int A = decltype(cls)::value;
#pragma unroll
      for (int e = 0; e < 2; ++e)
        glds16_asm_lds(cp + A * 128, poff[e] & dmask, lds0 + buf * SK_BUF + A * SK_CLS + (2 * wave + e) * 1024);
    };
    auto stageQ = [&](auto cls, int buf) __attribute__((always_inline)) {
      constexpr int B = decltype(cls)::value;
#pragma unroll
      for (int e = 0; e < 2; ++e)
        glds16_asm_lds(cq + B * 128, qoff[e] & dmask, lds0 + SK_Q + buf * SK_BUF + B * SK_CLS + (2 * wave + e) * 1024);
    };

    f32x4 acc[8][4], accb[2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    accb[0] = accb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias column sums ride on the matrix pipe (all-ones A operand).  The 8 P fragments of a wave row are spread over
    // the nt2 x 4 waves that hold them (tiles of the same row band, waves of the same wm): at most one per class and wave.
    // Tile (t1, t2) owns the same fragments in every one of its segments, so its partial sums add up like its tile does.
    int own[2] = {-1, -1};
    if (pr.bias_out != nullptr) {
      const int owners = (pr.N2 >> 8) * 4, me = sg.t2 * 4 + wn;
#pragma unroll
      for (int f = 0; f < 8; ++f)
        if (f % owners == me) own[f >> 2] = f & 3;
    }

    auto endL = [&]() __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
      __builtin_amdgcn_s_setprio(1);
      quad(fa, fb, ah, bh);
      __builtin_amdgcn_s_setprio(0);
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
    if (wm == 1) __builtin_amdgcn_s_barrier();               // group 1 runs one interval behind from here on
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
    if (wm == 0) __builtin_amdgcn_s_barrier();               // pairs with group 1's extra barrier: lockstep again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // everything staged has landed
    __syncthreads();

    // ---- tail: up to two K-tiles (buffer t), plain schedule
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t < ntail) {
        if (ragged && t == ntail - 1) {
          const int valid = pr.M & 63;
          for (int idx = tid; idx < (64 - valid) * 64; idx += 512) {       // 4 regions x 16 chunks per row
            const int row = valid + (idx >> 6), reg = (idx >> 4) & 3, ch = idx & 15;
            *reinterpret_cast<f32x4*>(smem + (reg >> 1) * SK_Q + t * SK_BUF + (reg & 1) * SK_CLS + row * 256 + ch * 16) =
                f32x4{0.f, 0.f, 0.f, 0.f};
          }
          __syncthreads();
        }
        bf16x8 fa0[2][4], fa1[2][4], fb0[2][2], fb1[2][2];
        readA(fa0, 0, t); readA(fa1, 1, t); readB(fb0, 0, t); readB(fb1, 1, t);
        quad(fa0, fb0, 0, 0); quad(fa1, fb0, 1, 0); quad(fa1, fb1, 1, 1); quad(fa0, fb1, 0, 1);
      }
    }

    // ---- store: lane owns rows .. + (lane & 15), columns .. + (lane >> 4) * 4 + 0..3 of each 16x16 block.
    // Whole tile: straight into the gradient (+= when the problem accumulates); partial tile: its slab (tile-local).
    const bool direct = sg.slot < 0;
    float* const base = direct ? pr.out + (size_t)c1 * pr.N2 + c2 : slabs + (size_t)sg.slot * SK_SLAB;
    const int ld = direct ? pr.N2 : 256;
    const bool rmw = direct && pr.accumulate != 0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = a * 128 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int c = b * 128 + wn * 32 + j * 16 + gq * 4;
            f32x4* dst = reinterpret_cast<f32x4*>(base + (size_t)r * ld + c);
            f32x4 v = acc[a * 4 + i][b * 2 + j];
            if (rmw) v += *dst;
            *dst = v;
          }
      }
      if (own[a] >= 0 && gq == 0) {
        const int col = a * 128 + wm * 64 + own[a] * 16 + (lane & 15);
        if (direct) {
          float* bd = pr.bias_out + c1 + col;
          *bd = rmw ? *bd + accb[a][0] : accb[a][0];
        } else {
          base[256 * 256 + col] = accb[a][0];
        }
      }
    }
    __syncthreads();                                         // the tail's LDS reads are done before the next prologue's DMA
  }
}

// out tile = (accumulate ? out : 0) + slabs of the tile, in slot order; one block per 16 rows of a split tile
__global__ __launch_bounds__(256) void tn_sk_fix_kernel(const SkProblem* __restrict__ probs, const SkFix* __restrict__ fixes,
                                                        const float* __restrict__ slabs) {
  const SkFix fx = fixes[blockIdx.x >> 4];
  const SkProblem pr = probs[fx.prob];
  const int part = blockIdx.x & 15, t = threadIdx.x;
  const int row = part * 16 + (t >> 4), col = (t & 15) * 16;
  float* const dst = pr.out + (size_t)(fx.c1 + row) * pr.N2 + fx.c2 + col;
  f32x4 v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = pr.accumulate ? reinterpret_cast<const f32x4*>(dst)[q] : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < fx.nslots; ++s) {
    const f32x4* src = reinterpret_cast<const f32x4*>(slabs + (size_t)(fx.slot0 + s) * SK_SLAB + row * 256 + col);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += src[q];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) reinterpret_cast<f32x4*>(dst)[q] = v[q];
  if (part == 0 && pr.bias_out != nullptr) {
    // bias column j of the row band belongs to fragment f = (j / 128) * 4 + (j % 64) / 16, summed by the tile with
    // t2 == (f % owners) / 4 (gemm_tn_sk_kernel: `own`)
    const int j = t, f = (j >> 7) * 4 + ((j & 63) >> 4), owners = (pr.N2 >> 8) * 4;
    if ((f % owners) >> 2 == fx.t2) {
      float b = pr.accumulate ? pr.bias_out[fx.c1 + j] : 0.f;
      for (int s = 0; s < fx.nslots; ++s) b += slabs[(size_t)(fx.slot0 + s) * SK_SLAB + 256 * 256 + j];
      pr.bias_out[fx.c1 + j] = b;
    }
  }
}

}  // namespace
}  // namespace oat

using namespace oat;

// ---- host side (plain C ABI; see include/oatrans_hip.h) -----------------------------------------------------------------
// Decompose `n` problems into per-workgroup segment lists.  No GPU involved.
//   splits == 0 ("stream"): the unit sequence (tile-major) is cut into `grid` contiguous shares; counts[3] = grid.  For MANY
//       SMALL tiles (the DistilBERT gradients: whole tiles per workgroup, nothing split).  A big tile cut this way is
//       computed by workgroups that share no operand rows with their neighbours - every workgroup streams its own P / Q
//       rows from HBM - so big problems use:
//   splits >= 1 ("uniform"): every problem must have the same M.  Every tile is split `splits` ways over M (even K-tile
//       counts), one segment per workgroup, tiles x splits workgroups (counts[3]; `grid` is ignored) in split-major order,
//       XCD-contiguous: the ~32 workgroups of one XCD work on the SAME row range of neighbouring tiles, so a row range of
//       P / Q is fetched once per XCD and shared through its L2 (the order of gemm_tn_pp.hip).  splits == 1 writes the
//       gradients directly (no slabs, no fix-up).
// Writes at most seg_cap segments, blocks + 1 offsets (seg_off must hold max(grid, tiles x splits) + 1 ints), at most fix_cap
// fix records; counts = {segments, fix records, slabs, blocks}.
extern "C" int oat_tn_group_plan(const void* problems, int n, int grid, int splits, void* segs_out, int seg_cap, int* seg_off,
                                 void* fix_out, int fix_cap, int* counts) {
  const SkProblem* pr = static_cast<const SkProblem*>(problems);
  SkSeg* segs = static_cast<SkSeg*>(segs_out);
  SkFix* fixes = static_cast<SkFix*>(fix_out);
  if (!pr || n <= 0 || grid <= 0 || splits < 0 || !segs || !seg_off || !fixes || !counts) { set_error("tn_group_plan: bad arguments"); return -4; }
  long long U = 0;
  for (int p = 0; p < n; ++p) {
    if (pr[p].M <= 0 || pr[p].N1 <= 0 || pr[p].N2 <= 0 || pr[p].N1 % 256 || pr[p].N2 % 256 || pr[p].ldp % 8 || pr[p].ldq % 8) {
      set_error("tn_group_plan: every problem needs M > 0, N1 % 256 == 0, N2 % 256 == 0, ldp % 8 == 0, ldq % 8 == 0");
      return -3;
    }
    const long long nkt = (pr[p].M + 63) / 64, npt = (nkt + 1) / 2;
    U += (long long)(pr[p].N1 / 256) * (pr[p].N2 / 256) * npt;
  }
  if (splits >= 1) {
    int T = 0;
    for (int p = 0; p < n; ++p) {
      if (pr[p].M != pr[0].M) { set_error("tn_group_plan: uniform splits need the same M in every problem"); return -3; }
      T += (pr[p].N1 / 256) * (pr[p].N2 / 256);
    }
    const int nkt = (pr[0].M + 63) / 64;
    int cps = (nkt + splits - 1) / splits;
    cps += cps & 1;
    const int S = (nkt + cps - 1) / cps, blocks = T * S;
    if (blocks > seg_cap || (S > 1 && T > fix_cap)) { set_error("tn_group_plan: tables too small"); return -6; }
    std::vector<int> tp(T), tt(T);                            // global tile -> problem, tile inside it
    for (int p = 0, k = 0; p < n; ++p)
      for (int t = 0, nt = (pr[p].N1 / 256) * (pr[p].N2 / 256); t < nt; ++t, ++k) { tp[k] = p; tt[k] = t; }
    const int q = blocks >> 3, r = blocks & 7;
    for (int b = 0; b < blocks; ++b) {
      const int xcd = b & 7, idx = b >> 3;
      const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;     // XCD-contiguous, bijective (gemm_tn_pp.hip)
      const int split = w / T, tg = w - split * T, p = tp[tg], nt2 = pr[p].N2 / 256, t1 = tt[tg] / nt2, t2 = tt[tg] - t1 * nt2;
      SkSeg& s = segs[b];
      s.prob = p; s.c1 = t1 * 256; s.c2 = t2 * 256; s.t2 = t2;
      s.kt0 = split * cps;
      s.n = (s.kt0 + cps < nkt ? s.kt0 + cps : nkt) - s.kt0;
      s.last = s.kt0 + s.n == nkt ? 1 : 0;
      s.slot = S == 1 ? -1 : tg * S + split;
      seg_off[b] = b;
    }
    seg_off[blocks] = blocks;
    int nfix = 0;
    if (S > 1)
      for (int tg = 0; tg < T; ++tg) {
        const int p = tp[tg], nt2 = pr[p].N2 / 256, t1 = tt[tg] / nt2, t2 = tt[tg] - t1 * nt2;
        SkFix& f = fixes[nfix++];
        f.prob = p; f.c1 = t1 * 256; f.c2 = t2 * 256; f.t2 = t2; f.slot0 = tg * S; f.nslots = S; f.pad0 = f.pad1 = 0;
      }
    counts[0] = blocks; counts[1] = nfix; counts[2] = S > 1 ? T * S : 0; counts[3] = blocks;
    return 0;
  }
  int nseg = 0, nfix = 0, nslot = 0;
  // cursor over (problem, tile, unit)
  int p = 0, tile = 0;
  long long u = 0;                                            // unit inside the current tile
  auto tiles_of = [&](int q) { return (pr[q].N1 / 256) * (pr[q].N2 / 256); };
  auto npt_of = [&](int q) { return (((long long)pr[q].M + 63) / 64 + 1) / 2; };
  int open_fix = -1;                                          // fix record of the tile under the cursor (if it is split)
  // Share boundaries, in units.  Where tiles are SMALL against a share (2 * units of the tile <= average share: the
  // DistilBERT gradients, 8 units per tile against ~20 per share) a boundary moves to the nearest tile boundary: shares
  // differ by at most half a tile, and no tile is split (no slab, no fix-up); big tiles are cut exactly.
  std::vector<long long> bound(grid + 1, 0);
  {
    std::vector<long long> tstart, tnpt;                      // first unit of every tile (then U), units of every tile
    long long acc_u = 0;
    for (int q = 0; q < n; ++q)
      for (int t = 0, nt = tiles_of(q); t < nt; ++t) { tstart.push_back(acc_u); tnpt.push_back(npt_of(q)); acc_u += npt_of(q); }
    tstart.push_back(acc_u);
    size_t ti = 0;
    for (int g = 1; g < grid; ++g) {
      long long b = (U * g) / grid;
      while (ti + 1 < tstart.size() - 1 && tstart[ti + 1] <= b) ++ti;          // tile containing unit b
      if (ti < tnpt.size() && 2 * tnpt[ti] * grid <= U) {
        const long long lo = tstart[ti], hi = tstart[ti + 1];
        b = (b - lo) * 2 < (hi - lo) ? lo : hi;
      }
      bound[g] = b < bound[g - 1] ? bound[g - 1] : b;
    }
    bound[grid] = U;
  }
  for (int g = 0; g < grid; ++g) {
    seg_off[g] = nseg;
    long long share = bound[g + 1] - bound[g];
    while (share > 0 && p < n) {
      const long long npt = npt_of(p), take = share < npt - u ? share : npt - u;
      const int nt2 = pr[p].N2 / 256, t1 = tile / nt2, t2 = tile - t1 * nt2;
      const long long nkt = ((long long)pr[p].M + 63) / 64;
      const bool whole = u == 0 && take == npt;
      if (nseg >= seg_cap) { set_error("tn_group_plan: segment table too small"); return -6; }
      SkSeg& s = segs[nseg++];
      s.prob = p; s.c1 = t1 * 256; s.c2 = t2 * 256; s.t2 = t2;
      s.kt0 = (int)(2 * u);
      const long long kend = 2 * (u + take) < nkt ? 2 * (u + take) : nkt;
      s.n = (int)(kend - 2 * u);
      s.last = (u + take == npt) ? 1 : 0;
      if (whole) s.slot = -1;
      else {
        s.slot = nslot++;
        if (open_fix < 0) {
          if (nfix >= fix_cap) { set_error("tn_group_plan: fix table too small"); return -6; }
          open_fix = nfix++;
          SkFix& f = fixes[open_fix];
          f.prob = p; f.c1 = s.c1; f.c2 = s.c2; f.t2 = t2; f.slot0 = s.slot; f.nslots = 0; f.pad0 = f.pad1 = 0;
        }
        fixes[open_fix].nslots++;
      }
      u += take;
      share -= take;
      if (u == npt) {                                         // next tile
        u = 0;
        open_fix = -1;
        if (++tile == tiles_of(p)) { tile = 0; ++p; }
      }
    }
  }
  seg_off[grid] = nseg;
  counts[0] = nseg; counts[1] = nfix; counts[2] = nslot; counts[3] = grid;
  return 0;
}

extern "C" size_t oat_tn_group_slab_bytes(int nslots) { return (size_t)(nslots > 0 ? nslots : 1) * SK_SLAB * sizeof(float); }

// All tables in device memory (layouts above); slabs: oat_tn_group_slab_bytes(counts[2]) bytes of workspace.
extern "C" int oat_tn_group_run(const void* d_problems, const void* d_segs, const void* d_seg_off, int grid,
                                const void* d_fixes, int nfix, void* d_slabs, void* stream) {
  if (!d_problems || !d_segs || !d_seg_off || grid <= 0 || !d_slabs || (nfix > 0 && !d_fixes)) { set_error("tn_group_run: null pointer"); return -4; }
  OAT_MAX_LDS(gemm_tn_sk_kernel, SK_LDS);
  hipStream_t s = (hipStream_t)stream;
  OAT_LAUNCH(gemm_tn_sk_kernel, dim3(grid), dim3(512), SK_LDS, s, static_cast<const SkProblem*>(d_problems),
             static_cast<const SkSeg*>(d_segs), static_cast<const int*>(d_seg_off), static_cast<float*>(d_slabs));
  int rc = check_launch("gemm_tn_sk");
  if (rc || nfix <= 0) return rc;
  OAT_LAUNCH(tn_sk_fix_kernel, dim3(nfix * 16), dim3(256), 0, s, static_cast<const SkProblem*>(d_problems),
             static_cast<const SkFix*>(d_fixes), static_cast<const float*>(d_slabs));
  return check_launch("tn_sk_fix");
}
