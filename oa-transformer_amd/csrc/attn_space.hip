// Divided SPACE attention of the SpaceTimeTransformer, fused (never materialises the
// [B*H*T, N, N+1] score tensor the reference writes: video_transformer.py:99-135 with the
// '(b f) n d' pattern :275-276, attn() :28-32).
//
// One workgroup = one (sample b, frame f, head h).  Keys/values = the frame's N patches + the
// sample's CLS token (key index N); padding keys up to NKP = 16*NKT are masked.
//   forward : S^T = K Q^T by MFMA (lane owns ONE query column -> softmax row reductions are
//             in-lane + 2 shuffles), P stays in registers as the B operand of O^T = V^T P^T,
//             V^T fragments come from the row-major LDS tile through ds_read_b64_tr_b16.
//   backward: flash-style with saved LSE and delta = rowsum(dO * O); phase A (lane = query)
//             produces dQ, phase B (lane = key) produces dK, dV.  The CLS *query* rides along
//             as query index N with its global LSE, so patch-key gradients are complete in one
//             pass; gradients of the shared CLS row are fp32 atomics into a side buffer.
// Token row layout (engine-wide): patch (b,f,n) -> row (b*T+f)*N+n ; CLS(b) -> row B*T*N+b.
#include "common.h"

namespace oat {

// chunk swizzle for [rows][64 bf16] LDS tiles.  The permutation {0,2,4,6,5,7,1,3}[(row>>1)&7] was found
// by exhaustive search over all 8! candidates against the REAL lane groups of both access kinds:
// ds_read_b128 row fragments (4 non-contiguous 16-lane groups whose lanes carry two different
// k-chunks) and ds_read_b64_tr_b16 transpose reads (32-lane halves, 8 rows x 32 B): zero conflicts
// for both (the first hand-derived swizzle measured SQ_LDS_BANK_CONFLICT / IDX_ACTIVE = 0.23-0.30).
OAT_DEV int sw8(int row) { const int rp = (row >> 1) & 7; return ((rp << 1) + (rp >> 2) * 5) & 7; }
OAT_DEV int tile_off(int row, int lc) { return row * 128 + ((lc ^ sw8(row)) << 4); }

// row fragment (A/B operand with k = head dim): 16 rows starting at r0, k-step ks (32 dims)
// rmax: rows beyond it read row rmax instead (TIME mode keeps 17-row tiles: every padding row is the CLS row)
OAT_DEV bf16x8 row_frag(const char* tile, int r0, int ks, int lane, int rmax = 0x7fffffff) {
  const int row = min(r0 + (lane & 15), rmax);
  return *reinterpret_cast<const bf16x8*>(tile + tile_off(row, ks * 4 + (lane >> 4)));
}
// transposed fragment: A operand [i = dim dt*16 + (lane&15)][k-slot (g,e)] = tile[row(g,e)][dim]
// rows of slot (g,e): e < 4 -> r0 + g*4 + e ; e >= 4 -> r0 + 16 + g*4 + (e-4)
OAT_DEV bf16x8 tr_frag(const char* tile, int r0, int dt, int lane, int rmax = 0x7fffffff) {
  const int s = lane & 15, g = lane >> 4;
  bf16x8 out;
  s16x4* o = reinterpret_cast<s16x4*>(&out);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int row = min(r0 + half * 16 + g * 4 + (s >> 2), rmax);
    const int lc = dt * 2 + ((s & 3) >> 1);
    o[half] = lds_tr16(tile + tile_off(row, lc) + ((s & 1) << 3));
  }
  return out;
}

// Output tiles leave the matrix pipe as O^T / dQ^T / dK^T / dV^T accumulators: lane (g, col) holds dims 16 dt + 4 g .. + 3 of
// local row `col`.  Stored from there, one instruction writes 16 rows x 32 contiguous bytes and every 128-byte line of
// the output is written four times over as a 32-byte fragment; measured (profiles/round4b_attention_ablation.md) the
// 231 MB of dqkv stores of the TIME backward cost 77 us of its 190 - more than its 385 MB of loads.  So the tile takes
// one trip through a wave-private LDS scratch (16 rows x (128 + 16) bytes: ds_write_b64 per lane and dt, ds_read_b128
// back) and leaves as whole 128-byte lines, 16 bytes per lane, 8 rows per store instruction.
// rowptr(j) -> address of local row j's 64-element head slice, or nullptr (row not stored).  Wave-uniform call.
// Scratch layout (round 5): 16 rows x 128 bytes, no padding; the 8-byte slot a lane writes is XOR-ed with a row key so that both sides are
// free of bank conflicts - slot' = slot ^ ((row & 7) << 1 | row >> 3).  Writes (ds_write_b64, groups of 16 consecutive lanes = 16 rows, one
// logical slot): 16 distinct slots = all 32 banks once.  Reads (ds_read_b128, 8 lanes per row): the two halves of a 16-byte chunk stay
// together (the key's upper bits permute chunks, its lowest bit swaps the halves of rows 8..15, undone in registers), and the four rows a
// 16-lane read group touches land on disjoint quarters of the 64 banks.  The 144-byte pitch this replaces measured
// SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = 0.64 in the TIME backward (two-way on every write, partial overlaps on the reads).
constexpr int SCR_PITCH = 128, SCR_BYTES = 16 * SCR_PITCH;
template <class F>
OAT_DEV void store_rows16(char* scr, const f32x4 (&acc)[4], float mul, int lane, F&& rowptr) {
  const int col = lane & 15, g = lane >> 4;
  const int key = ((col & 7) << 1) | (col >> 3);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const bf16x4 o = {f2bf(acc[dt][0] * mul), f2bf(acc[dt][1] * mul), f2bf(acc[dt][2] * mul), f2bf(acc[dt][3] * mul)};
    *reinterpret_cast<bf16x4*>(scr + col * SCR_PITCH + (((dt * 4 + g) ^ key) << 3)) = o;       // logical 8-byte slot dt * 4 + g
  }
  __builtin_amdgcn_wave_barrier();                       // DS instructions of a wave execute in order: the reads see the writes
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int j = p * 8 + (lane >> 3), ch = lane & 7;
    uint4 v = *reinterpret_cast<const uint4*>(scr + j * SCR_PITCH + ((ch ^ (j & 7)) << 4));
    if (p == 1) v = uint4{v.z, v.w, v.x, v.y};           // rows 8..15: the key's lowest bit swapped the two halves of every chunk
    bf16* dst = rowptr(j);
    if (dst) *reinterpret_cast<uint4*>(dst + ch * 8) = v;
  }
  __builtin_amdgcn_wave_barrier();                       // ... and the next tile's writes come after these reads
}

struct SpaceArgs {
  const bf16* qkv; int ldqkv;
  bf16* out; int ldo;              // fwd: attention output ; bwd: saved attention output (read)
  float* lse;                      // [M, H]
  const bf16* dout; int lddo;      // bwd
  bf16* dqkv; int lddqkv;          // bwd
  float* cls_side;                 // bwd: [B, H, 3, 64] fp32 (dq, dk, dv of the CLS row)
  int B, T, N, H, D;
  float scale;
  int gpw;                         // TIME backward: position groups per workgroup (0 = 1)
  int* done;                       // bwd, optional: [B, H] tickets (zero on entry, left zero) - the LAST workgroup that feeds
                                   // cls_side[b][h] writes the CLS row of dqkv itself (no oat_attn_cls_finalize launch)
};

// Two clips in ONE launch (the object frame + the video clip of the OA models: a one-frame clip alone is B x H problems, a third
// of the GPU): workgroups [0, n0) work on s[0], the rest on s[1].  Single-clip launches set n0 to the grid size.
struct SpaceArgs2 { SpaceArgs s[2]; int n0; int nclips; };

// forward: 4 waves, 56 KB LDS -> two workgroups per CU overlap each other's prologue;
// backward: 116 KB LDS pins one workgroup per CU, so it runs 8 waves (two per SIMD) to hide the
// MFMA / LDS / exp latency chains (measured 899 -> 526 us at B=32, T=8).
constexpr int FWD_THREADS = 512, BWD_THREADS = 512;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// stage a [NKP][64] tile from token rows with LDS-DMA: j < N -> patch row, j == N -> CLS row, j > N ->
// alias of the CLS row (padding rows are never used un-masked: their scores are -inf / their P is 0, so they
// only need to be finite).  All slabs of a tile are in flight at once (the earlier load->wait->ds_write loop
// serialised ~7 HBM latencies per tile); the bank swizzle is applied on the per-lane SOURCE address.
// Which token row a workgroup-local row j is.  SPACE: the workgroup owns frame (b, f): j < N -> patch j of that frame,
// j == N -> the sample's CLS row.  TIME (backward only): the workgroup owns G = 16 / T consecutive patch POSITIONS of one
// sample across all T frames - local row j = (position n0 + j / T, frame j % T), CLS at index N = G * T - and the
// attention mask is block-diagonal: a patch query sees the keys of its own position plus CLS.  Positions past the end of
// the frame (ragged last group) read a clamped row and are masked out.
struct RowMap {
  size_t base_row, cls_row;
  int N;                     // CLS index = number of local patch rows
  int T, Nf, n0;             // TIME: frames, patches per frame, first position of the group
  template <bool TIME> OAT_DEV size_t row(int j) const {
    if (j >= N) return cls_row;
    if (!TIME) return base_row + j;
    return base_row + (size_t)(j % T) * Nf + min(n0 + j / T, Nf - 1);     // base_row = b * T * Nf
  }
  template <bool TIME> OAT_DEV bool live(int j) const { return !TIME || j >= N || n0 + j / T < Nf; }
  // may query q see key k (both <= N)?
  template <bool TIME> OAT_DEV bool sees(int q, int k) const {
    if (!TIME) return true;
    if (q >= N) return live<TIME>(k);
    if (k >= N) return live<TIME>(q);
    return q / T == k / T && live<TIME>(q);
  }
};

// ASM: the LDS-DMA is issued from inline asm, invisible to hipcc's wait-count pass (which drains every DMA it knows of
// with vmcnt(0) before the next LDS read): the caller waits itself - used to stage the NEXT problem under the current one.
template <int NKT, int NTHR, bool TIME = false, bool ASM = false>
OAT_DEV void load_tile(char* tile, const bf16* src, int ld, int col, const RowMap& rm) {
  constexpr int NSLAB = NKT * 2;                    // 8 rows (1 KB) per wave instruction
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int slab = wave; slab < NSLAB; slab += NTHR / 64) {
    const int j = slab * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ sw8(j);
    if (TIME && j > 16) continue;                   // 17-row tiles: rows past the CLS slot are never staged (inactive lanes do not write)
    if (ASM) glds16_asm(src + rm.row<TIME>(j) * ld + col + lc * 8, tile + slab * 1024);
    else glds16(src + rm.row<TIME>(j) * ld + col + lc * 8, tile + slab * 1024);
  }
}

// NFIX: patches per frame known at compile time (196 = 224^2 / 16^2, the shape every BASELINE config but the 336^2 one runs): every
// "does key tile kt hold a real / a padding key" test of the score pass is wave-uniform AND loop-invariant, and with a run-time N hipcc
// hoists them as 112 lane masks that it spills to VGPR lanes and reads back (8 v_readlane per key tile) behind four taken branches per
// key tile - the 28 score MFMAs of a query tile end up in 14 two-instruction basic blocks.  With N a constant the pass is straight-line.
template <int NKT, int NFIX = 0>
__global__ __launch_bounds__(FWD_THREADS, NKT <= 14 ? 4 : 2) void attn_space_fwd_kernel(SpaceArgs2 aa) {
  const bool second = (int)blockIdx.x >= aa.n0;            // workgroup-uniform
  const SpaceArgs& a = aa.s[second ? 1 : 0];
  const int bid = (int)blockIdx.x - (second ? aa.n0 : 0);
  constexpr int NKP = NKT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kt = smem;
  char* Vt = smem + NKP * 128;
  char* const scr = smem + 2 * NKP * 128 + (threadIdx.x >> 6) * SCR_BYTES;      // this wave's output scratch
  const int h = bid % a.H;
  const int bf = bid / a.H;                   // b * T + f
  const int b = bf / a.T;
  const int N = NFIX > 0 ? NFIX : a.N;
  const size_t base_row = (size_t)bf * N;
  const size_t cls_row = (size_t)a.B * a.T * N + b;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4;
  const int nqt = (N + 15) / 16;
  // this wave's first Q fragment is requested before the K/V tiles so its latency hides behind them
  auto load_q = [&](int qt, bf16x8* dst) {
    const size_t qrow = base_row + min(qt * 16 + (lane & 15), N - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      dst[ks] = *reinterpret_cast<const bf16x8*>(a.qkv + qrow * a.ldqkv + h * 64 + ks * 32 + g * 8);
  };
  bf16x8 qnext[2];
  load_q(min(wave, nqt - 1), qnext);
  const RowMap rm{base_row, cls_row, N, 1, N, 0};
  load_tile<NKT, FWD_THREADS>(Kt, a.qkv, a.ldqkv, a.D + h * 64, rm);
  load_tile<NKT, FWD_THREADS>(Vt, a.qkv, a.ldqkv, 2 * a.D + h * 64, rm);
  __syncthreads();

  const float c2 = a.scale * LOG2E;
  for (int qt = wave; qt < nqt; qt += FWD_THREADS / 64) {
    const int qi = qt * 16 + (lane & 15);
    bf16x8 qf[2] = {qnext[0], qnext[1]};
    if (qt + FWD_THREADS / 64 < nqt) load_q(qt + FWD_THREADS / 64, qnext);     // prefetch the next tile's Q
    // raw scores: the softmax scale goes into the exponent's fma (max commutes with a positive scale), and only the key
    // tiles that reach past key N are masked per element - the kernel is VALU-bound in this stretch (10 -> 7.5 issue
    // slots per score); a key tile that is pure padding (NKT is even: 14 tiles for 197 keys) skips its MFMAs as well
    f32x4 st[NKT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4 acc = {0, 0, 0, 0};
      if (kt * 16 <= N) {                                  // wave-uniform
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Kt, kt * 16, ks, lane), qf[ks], acc, 0, 0, 0);
      }
      if (kt * 16 + 15 <= N) {                             // every key of the tile is real
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + g * 4 + r;
          acc[r] = key <= N ? acc[r] : -INFINITY;
          m = fmaxf(m, acc[r]);
        }
      }
      st[kt] = acc;
      if (kt & 1) __builtin_amdgcn_sched_barrier(0);     // keeps the K fragments of at most two tiles live (3 waves/SIMD)
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mc = m * c2;
    m = mc;                                                // log2-domain maximum, as the LSE below expects
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { st[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], c2, -mc)); l += st[kt][r]; }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    f32x4 ot[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < NKT / 2; ++u) {
      const bf16x8 pb = {f2bf(st[2 * u][0]), f2bf(st[2 * u][1]), f2bf(st[2 * u][2]), f2bf(st[2 * u][3]),
                         f2bf(st[2 * u + 1][0]), f2bf(st[2 * u + 1][1]), f2bf(st[2 * u + 1][2]), f2bf(st[2 * u + 1][3])};
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Vt, u * 32, dt, lane), pb, ot[dt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    store_rows16(scr, ot, 1.0f / l, lane, [&](int j) -> bf16* {
      const int q = qt * 16 + j;
      return q < N ? a.out + (base_row + q) * a.ldo + h * 64 : nullptr;
    });
    if (qi < N && g == 0) a.lse[(base_row + qi) * a.H + h] = (m + log2f(l)) * LN2;
  }
}

// BIG (frames of 224..447 patches, e.g. 336^2 / 16 -> 441): four [NKP][64] tiles no longer fit the LDS, but
// neither phase needs all four.  Phase A (dQ) streams over K and V and touches Q / dO only as the per-wave row
// fragments; phase B (dK, dV) is the mirror image.  So LDS holds TWO tiles: K, V during phase A, then - after a
// barrier - Q, dO for phase B, and the per-wave row fragments come straight from global memory.
// WIDE: 16 waves (four per SIMD, <= 128 VGPRs) that each own ONE 16-row tile instead of 8 waves with a tile pair: the
// kernel is bound by its dependent LDS -> MFMA -> exp -> MFMA chains, not by LDS or MFMA throughput, so twice the
// resident waves hide twice the latency.
// WIDE == 2: single tiles on 8 waves, registers capped at 128: with the two-tile BIG layout (59 KB at 196 patches) TWO
// workgroups share a CU and one's loads / stores overlap the other's MFMA phases.
// TIME (see RowMap): the same kernel as a ONE-wave workgroup (WIDE == 3) on a 16-row mini problem; the time-attention
// backward of attn_time.hip evaluates every score twice on the VALU (8-lane dot products), this one spends two dozen
// mostly-masked MFMAs per problem and is bound by its loads and stores.
// A TIME workgroup walks `gpw` consecutive position groups of its (sample, head): the CLS row is row N of every one of
// them, so its three gradients stay in registers across the walk and cost one set of atomics per workgroup (one per
// group put 7 M contended atomics on 4.6 K addresses).  TT = frame count at compile time (row <-> (position, frame)
// is a division per row otherwise).
// CLSQ (round 6, the pruned top block: engine/video.py _top_block_bwd_pruned): only the CLS query carries a gradient - dO of every
// patch query is exactly zero and its lse is +inf, so P = dS = 0 there and everything those queries contribute is an exact zero
// (the kernel run in full adds 12 of 13 query tiles of zeros).  The instance skips them: phase A computes the one tile that holds
// query N and stores zeros for the other dQ rows, phase B walks only the query pair that holds it, and delta / lse are fetched
// for row N alone.  Skipped terms are +0.0 added to fp32 accumulators: the gradients are bit-identical to the full launch.
template <int NKT, bool BIG, int WIDE, bool TIME = false, int TT = 0, bool CLSQ = false>
OAT_DEV void attn_space_bwd_body(const SpaceArgs& a, const int bid) {
  static_assert(!CLSQ || !TIME, "CLS-query-only instance: SPACE mode");
  constexpr int NKP = NKT * 16;
  constexpr int THR = WIDE == 1 ? 1024 : WIDE == 3 ? 64 : BWD_THREADS, STEP = (WIDE == 1 || WIDE == 2) ? 1 : 2;
  static_assert(!TIME || (!BIG && NKT == 2), "time mode = 16 local rows + CLS");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // TIME: rows 0..16 of a tile are real (16 patch rows + CLS), every padding row reads row 16: 17-row tiles, 8.8 KB per workgroup
  constexpr int TROWS = TIME ? 17 : NKP, RMAX = TIME ? 16 : 0x7fffffff;
  constexpr int BUF = (BIG ? 2 : 4) * TROWS * 128 + 2 * NKP * 4;     // one problem's tiles + (lse, delta); TIME keeps two
  const int h = bid % a.H;
  const int bf = bid / a.H;
  const int Tc = TT > 0 ? TT : a.T;
  const int G = TIME ? 16 / Tc : 1, ngrp = TIME ? (a.N + G - 1) / G : a.T;
  const int gpw = TIME ? max(a.gpw, 1) : 1, nchunk = (ngrp + gpw - 1) / gpw;      // bf = b * nchunk + chunk
  const int b = bf / nchunk, f_lo = (bf % nchunk) * gpw, f_hi = min(f_lo + gpw, ngrp);
  const int N = TIME ? G * Tc : a.N;
  const size_t cls_row = (size_t)a.B * a.T * a.N + b;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4;
  const float c2 = a.scale * LOG2E;
  float* side = a.cls_side + ((size_t)b * a.H + h) * 3 * 64;
  f32x4 cls_dq[4], cls_dk[4], cls_dv[4];                // TIME: the CLS row's gradients, summed over the walk
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { cls_dq[dt] = f32x4{0, 0, 0, 0}; cls_dk[dt] = f32x4{0, 0, 0, 0}; cls_dv[dt] = f32x4{0, 0, 0, 0}; }
  // delta = rowsum(dO * O) and lse (log2 units), 8 lanes per row: request (registers) and finish (LDS) are separate so
  // that TIME can request the NEXT group's rows before the current group's phases
  constexpr int ITER = (NKP * 8 + THR - 1) / THR;
  bf16x8 gv[ITER], ov[ITER];
  float lv[ITER];
  // TIME: delta of a PATCH query is formed inside phase A from the scores themselves (delta_q = sum_k P_qk dP_qk over the 9 keys
  // the query sees - the same number as rowsum(dO * O), from fp32 P instead of the bf16-rounded O): the saved attention output is
  // not read at all and dO only as the tile (77 of 616 MB per launch).  The CLS query's delta spans every key of the sample, so it
  // alone still comes from its dO / O rows, once per workgroup; a group requests the lse of its 17 rows (one lane per row).
  float cls_delta = 0.f;
  if constexpr (TIME) {
    const int c = lane & 7;
    const bf16x8 gc = *reinterpret_cast<const bf16x8*>(a.dout + cls_row * a.lddo + h * 64 + c * 8);
    const bf16x8 oc = *reinterpret_cast<const bf16x8*>(a.out + cls_row * a.ldo + h * 64 + c * 8);
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d += bf2f(gc[e]) * bf2f(oc[e]);
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    cls_delta = d;                                     // every lane holds the sum of its 8-lane group = the row's delta
  }
  auto delta_request = [&](const RowMap& rq) {
    if constexpr (TIME) {
      lv[0] = a.lse[rq.row<TIME>(min(lane, N)) * a.H + h];
      return;
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = it * THR + threadIdx.x;
      const int j = min(idx >> 3, N), c = idx & 7;
      const size_t r = rq.row<TIME>(j);
      if (CLSQ && j != N) {                              // a patch query of the pruned top block: dO = 0, lse = +inf
        gv[it] = bf16x8{}; ov[it] = bf16x8{}; lv[it] = INFINITY;
        continue;
      }
      gv[it] = *reinterpret_cast<const bf16x8*>(a.dout + r * a.lddo + h * 64 + c * 8);
      ov[it] = *reinterpret_cast<const bf16x8*>(a.out + r * a.ldo + h * 64 + c * 8);
      lv[it] = a.lse[r * a.H + h];
    }
  };
  auto stage = [&](const RowMap& rq, char* base) {      // K, V, Q, dO tiles of one problem
    if (!BIG) load_tile<NKT, THR, TIME, TIME>(base + 2 * TROWS * 128, a.qkv, a.ldqkv, h * 64, rq);
    load_tile<NKT, THR, TIME, TIME>(base, a.qkv, a.ldqkv, a.D + h * 64, rq);
    load_tile<NKT, THR, TIME, TIME>(base + TROWS * 128, a.qkv, a.ldqkv, 2 * a.D + h * 64, rq);
    if (!BIG) load_tile<NKT, THR, TIME, TIME>(base + 3 * TROWS * 128, a.dout, a.lddo, h * 64, rq);
  };
  auto group_map = [&](int f) { return RowMap{TIME ? (size_t)b * Tc * a.N : (size_t)bf * a.N, cls_row, N, Tc, a.N, f * G}; };
  if constexpr (TIME) {                                  // pipeline prologue: the first group's tiles and delta rows
    const RowMap r0 = group_map(f_lo);
    stage(r0, smem);
    delta_request(r0);
  }
  for (int f = f_lo; f < f_hi; ++f) {                   // f: frame (SPACE, one pass) / position group (TIME); 0 owns the CLS->CLS pair
  const RowMap rm = group_map(f);
  const size_t base_row = rm.base_row;
  char* const buf = smem + (TIME ? ((f - f_lo) & 1) * BUF : 0);
  char* Kt = buf;
  char* Vt = buf + TROWS * 128;
  char* Qt = BIG ? Kt : buf + 2 * TROWS * 128;           // BIG: aliases, valid in phase B only
  char* Dt = BIG ? Vt : buf + 3 * TROWS * 128;           // dO tile
  float* lse_s = reinterpret_cast<float*>(buf + (BIG ? 2 : 4) * TROWS * 128);
  float* del_s = lse_s + NKP;
  // output scratch (store_rows16).  SPACE: a region per wave behind the tiles.  TIME (one wave, two problem buffers): the
  // problem's own tile buffer, once phase B is done with it - the dQ tile waits in registers until then.
  char* const scr = TIME ? buf : smem + BUF + wave * SCR_BYTES;
  f32x4 dq_keep[4];
  if constexpr (TIME) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this group's tiles (staged one group ago) have landed
  } else {
    stage(rm, buf);
    delta_request(rm);
  }
  // row fragment of a token-row matrix straight from global memory (same element order as row_frag on a tile)
  auto grow_frag = [&](const bf16* src, int ld, int col, int r0, int ks, int lane_) {
    const int j = r0 + (lane_ & 15);
    const size_t r = j < N ? base_row + j : cls_row;
    return *reinterpret_cast<const bf16x8*>(src + r * ld + col + (ks * 4 + (lane_ >> 4)) * 8);
  };
  if constexpr (TIME) {
    if (lane < NKP) {
      del_s[lane] = lane == N ? cls_delta : 0.f;         // patch rows: written by phase A below
      lse_s[lane] = lane <= N ? lv[0] * LOG2E : INFINITY;
    }
  } else {
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = it * THR + threadIdx.x;
    const int j = idx >> 3, c = idx & 7;
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d += bf2f(gv[it][e]) * bf2f(ov[it][e]);
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0xB1, 0xF, 0xF, true));
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x4E, 0xF, 0xF, true));
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x141, 0xF, 0xF, true));
    if (c == 0 && j < NKP) {
      del_s[j] = j <= N ? d : 0.f;
      lse_s[j] = j <= N ? lv[it] * LOG2E : INFINITY;     // a padding QUERY's exp2(s c2 - lse) is exactly 0: no select in the passes below
    }
  }
  }
  __syncthreads();
  if constexpr (TIME) {
    if (f + 1 < f_hi) {                                  // the next group's tiles and delta rows fly under this group's phases
      const RowMap rn = group_map(f + 1);
      stage(rn, smem + (((f + 1 - f_lo) & 1) * BUF));
      delta_request(rn);
    }
  }

  // Both phases process TWO 16-wide tiles per wave so that every LDS fragment (row fragments and
  // transpose-read fragments) feeds two MFMA chains: half the LDS traffic per MFMA and two independent
  // dependency chains per wave (the kernel is latency-bound at two waves per SIMD).
  auto pack8 = [](const f32x4& a, const f32x4& b) {
    return bf16x8{f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
  };
  const int ntile = N / 16 + 1;                         // tiles that contain a real row (index <= N)
  // 32-row key (phase A) / query (phase B) pairs that contain a real row.  The pair loops below run this trip count and have
  // NO early exit: with `if (u * 32 > N) break;` hipcc kept the loop-carried accumulators in a second register block and
  // copied all 16 (dQ) / 32 (dK, dV) of them at every loop end - a quarter of the loop's VALU instructions (ISA count).
  const int nu = min(NKT / 2, N / 32 + 1);

  // ------------------------------------------------ phase A: lane = query column, produces dQ
  // streamed over key pairs: dS of keys [32u, 32u+32) is consumed by the dQ MFMAs right away
  for (int pr = wave; pr * STEP < ntile; pr += THR / 64) {
    const int qt0 = pr * STEP;
    if constexpr (CLSQ) {
      static_assert(!CLSQ || STEP == 1, "CLS-query-only instance: one tile per wave");
      if (qt0 != N / 16) {                                 // wave-uniform: a tile of patch queries only - dQ is exactly zero
        const f32x4 z[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
        store_rows16(scr, z, 0.f, lane, [&](int j) -> bf16* {
          const int q = qt0 * 16 + j;
          return q < N ? a.dqkv + rm.row<TIME>(q) * a.lddqkv + h * 64 : nullptr;
        });
        continue;
      }
    }
    const bool two = STEP == 2 && qt0 + 1 < ntile;                   // wave-uniform
    const int qiA = qt0 * 16 + (lane & 15), qiB = qiA + 16;
    const float lqA = lse_s[qiA], dlA = del_s[qiA], lqB = lse_s[qiB], dlB = del_s[qiB];
    bf16x8 qfA[2], dfA[2], qfB[2], dfB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (BIG) {
        qfA[ks] = grow_frag(a.qkv, a.ldqkv, h * 64, qt0 * 16, ks, lane); dfA[ks] = grow_frag(a.dout, a.lddo, h * 64, qt0 * 16, ks, lane);
        qfB[ks] = grow_frag(a.qkv, a.ldqkv, h * 64, qt0 * 16 + 16, ks, lane); dfB[ks] = grow_frag(a.dout, a.lddo, h * 64, qt0 * 16 + 16, ks, lane);
      } else {
        qfA[ks] = row_frag(Qt, qt0 * 16, ks, lane, RMAX); dfA[ks] = row_frag(Dt, qt0 * 16, ks, lane, RMAX);
        qfB[ks] = row_frag(Qt, qt0 * 16 + 16, ks, lane, RMAX); dfB[ks] = row_frag(Dt, qt0 * 16 + 16, ks, lane, RMAX);
      }
    }
    f32x4 dqA[4], dqB[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dqA[dt] = f32x4{0, 0, 0, 0}; dqB[dt] = f32x4{0, 0, 0, 0}; }
    // SPACE: no masked path.  A padding query has lse = +inf (above); what depends on the KEY - padding keys and the CLS -> CLS pair that
    // only frame 0 counts - lives in the last pass alone and is folded into the lse OPERAND of its eight rows: lqv[hf][r] = ok ? lse : +inf,
    // so exp2(s c2 - lqv) is exactly zero where the mask was (no compare / select per element; 21 % of the passes took the masked path)
    f32x4 lqvA[2] = {f32x4{lqA, lqA, lqA, lqA}, f32x4{lqA, lqA, lqA, lqA}}, lqvB[2] = {f32x4{lqB, lqB, lqB, lqB}, f32x4{lqB, lqB, lqB, lqB}};
    auto last_pass_lse = [&]() {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = (2 * (nu - 1) + hf) * 16 + g * 4 + r;
          const bool clsdup = key == N && f != 0;
          lqvA[hf][r] = (key <= N && !(qiA == N && clsdup)) ? lqA : INFINITY;
          lqvB[hf][r] = (key <= N && !(qiB == N && clsdup)) ? lqB : INFINITY;
        }
    };
    if (!TIME && nu == 1) last_pass_lse();
#pragma unroll 1
    for (int u = 0; u < nu; ++u) {
      f32x4 dsA[2], dsB[2];
      const bool plain = !TIME;                                                               // (TIME keeps its block-diagonal masks)
      f32x4 dpA[2];                                      // TIME: dP of tile A (patch queries), whose delta is formed below
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int kt = 2 * u + hf;
        if (hf == 1 && kt * 16 > N) {      // wave-uniform: the pass's second key tile is pure padding (197 keys: tile 13 of 14) - its dS is zero
          dsA[1] = f32x4{0, 0, 0, 0}; dsB[1] = f32x4{0, 0, 0, 0}; dpA[1] = f32x4{0, 0, 0, 0};
          continue;
        }
        // dP - delta comes out of the matrix pipe: the accumulator starts at -delta (this lane's query) instead of zero
        const float dA0 = TIME ? 0.f : -dlA;             // TIME: tile A holds the patch queries, delta not known yet
        f32x4 sA = {0, 0, 0, 0}, pA = {dA0, dA0, dA0, dA0}, sB = {0, 0, 0, 0}, pB = {-dlB, -dlB, -dlB, -dlB};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 kf = row_frag(Kt, kt * 16, ks, lane, RMAX), vf = row_frag(Vt, kt * 16, ks, lane, RMAX);
          sA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qfA[ks], sA, 0, 0, 0);
          pA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dfA[ks], pA, 0, 0, 0);
          if (two) {
            sB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qfB[ks], sB, 0, 0, 0);
            pB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dfB[ks], pB, 0, 0, 0);
          }
        }
        if (plain) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dsA[hf][r] = __builtin_amdgcn_exp2f(sA[r] * c2 - lqvA[hf][r]) * pA[r];
            dsB[hf][r] = two ? __builtin_amdgcn_exp2f(sB[r] * c2 - lqvB[hf][r]) * pB[r] : 0.f;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + g * 4 + r;
            const bool clsdup = key == N && f != 0;       // CLS->CLS pair is counted once (frame 0)
            const bool okA = key <= N && qiA <= N && !(qiA == N && clsdup) && rm.sees<TIME>(qiA, key);
            const bool okB = key <= N && qiB <= N && !(qiB == N && clsdup) && rm.sees<TIME>(qiB, key);
            if constexpr (TIME) {
              dsA[hf][r] = okA ? __builtin_amdgcn_exp2f(sA[r] * c2 - lqA) : 0.f;      // P for now; dS after the delta below
              dpA[hf][r] = pA[r];
            } else {
              dsA[hf][r] = okA ? __builtin_amdgcn_exp2f(sA[r] * c2 - lqA) * pA[r] : 0.f;
            }
            dsB[hf][r] = (two && okB) ? __builtin_amdgcn_exp2f(sB[r] * c2 - lqB) * pB[r] : 0.f;
          }
        }
      }
      if constexpr (TIME) {
        // delta of this lane's patch query: sum over its keys of P dP - 8 accumulator rows in this lane, the other 24 key rows in the
        // three lanes that share the query column (g = lane >> 4); then dS = P (dP - delta).  Lanes g == 0 leave it for phase B.
        float dl = 0.f;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int r = 0; r < 4; ++r) dl = __builtin_fmaf(dsA[hf][r], dpA[hf][r], dl);
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (qiA == N) dl = cls_delta;                    // frame counts that do not divide 16: the CLS query sits in tile A (N < 16)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int r = 0; r < 4; ++r) dsA[hf][r] *= dpA[hf][r] - dl;
        if (g == 0 && qiA < N) del_s[qiA] = dl;
      }
      const bf16x8 sbA = pack8(dsA[0], dsA[1]), sbB = pack8(dsB[0], dsB[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 tf = tr_frag(Kt, u * 32, dt, lane, RMAX);
        dqA[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, sbA, dqA[dt], 0, 0, 0);
        if (two) dqB[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, sbB, dqB[dt], 0, 0, 0);
      }
      if (!TIME && u == nu - 2) last_pass_lse();              // wave-uniform
    }
    // local rows [t0, t0 + 16) of an output tile -> columns `col0` of their dqkv rows
    auto put_rows = [&](int t0, const f32x4 (&acc)[4], float mul, int col0) {
      store_rows16(scr, acc, mul, lane, [&](int j) -> bf16* {
        const int q = t0 + j;
        return (q < N && rm.live<TIME>(q)) ? a.dqkv + rm.row<TIME>(q) * a.lddqkv + col0 + h * 64 : nullptr;
      });
    };
    auto put_q = [&](int qi, const f32x4 (&dq)[4]) {
      if (qi - (lane & 15) < N) {                          // wave-uniform: the tile holds a patch row
        if constexpr (TIME) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) dq_keep[dt] = dq[dt];
        } else {
          put_rows(qi - (lane & 15), dq, a.scale, 0);
        }
      }
      if (qi == N) {
        if constexpr (TIME) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) cls_dq[dt] += dq[dt];
        } else {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(side + dt * 16 + g * 4 + r, dq[dt][r] * a.scale);
        }
      }
    };
    put_q(qiA, dqA);
    if (two) put_q(qiB, dqB);
  }

  // ------------------------------------------------ phase B: lane = key column, produces dK, dV
  if constexpr (BIG) {
    __syncthreads();                                      // every wave is done with K, V in LDS
    load_tile<NKT, THR, TIME>(Qt, a.qkv, a.ldqkv, h * 64, rm);
    load_tile<NKT, THR, TIME>(Dt, a.dout, a.lddo, h * 64, rm);
    __syncthreads();
  }
  for (int pr = wave; pr * STEP < ntile; pr += THR / 64) {
    const int kt0 = pr * STEP;
    const bool two = STEP == 2 && kt0 + 1 < ntile;
    const int keyA = kt0 * 16 + (lane & 15), keyB = keyA + 16;
    bf16x8 kfA[2], vfA[2], kfB[2], vfB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (BIG) {
        kfA[ks] = grow_frag(a.qkv, a.ldqkv, a.D + h * 64, kt0 * 16, ks, lane); vfA[ks] = grow_frag(a.qkv, a.ldqkv, 2 * a.D + h * 64, kt0 * 16, ks, lane);
        kfB[ks] = grow_frag(a.qkv, a.ldqkv, a.D + h * 64, kt0 * 16 + 16, ks, lane); vfB[ks] = grow_frag(a.qkv, a.ldqkv, 2 * a.D + h * 64, kt0 * 16 + 16, ks, lane);
      } else {
        kfA[ks] = row_frag(Kt, kt0 * 16, ks, lane, RMAX); vfA[ks] = row_frag(Vt, kt0 * 16, ks, lane, RMAX);
        kfB[ks] = row_frag(Kt, kt0 * 16 + 16, ks, lane, RMAX); vfB[ks] = row_frag(Vt, kt0 * 16 + 16, ks, lane, RMAX);
      }
    }
    f32x4 dkA[4], dvA[4], dkB[4], dvB[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dkA[dt] = f32x4{0, 0, 0, 0}; dvA[dt] = f32x4{0, 0, 0, 0}; dkB[dt] = f32x4{0, 0, 0, 0}; dvB[dt] = f32x4{0, 0, 0, 0};
    }
#pragma unroll 1
    for (int u = CLSQ ? N / 32 : 0; u < nu; ++u) {       // CLSQ: only the query pair that holds query N has P != 0
      f32x4 pvA[2], svA[2], pvB[2], svB[2];
      // wave-uniform.  Patch keys only: every pass is plain - a padding query has lse = +inf (P = 0 without a select), the CLS query sees
      // every patch key.  Only the key tile that holds the CLS key and the padding keys keeps the masked form.
      const bool plain = !TIME && (kt0 + (two ? 2 : 1)) * 16 <= N;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int q0 = u * 32 + hf * 16;
        if (hf == 1 && q0 > N) {           // wave-uniform: the pass's second query tile is pure padding - P and dS are zero
          pvA[1] = f32x4{0, 0, 0, 0}; svA[1] = f32x4{0, 0, 0, 0}; pvB[1] = f32x4{0, 0, 0, 0}; svB[1] = f32x4{0, 0, 0, 0};
          continue;
        }
        // this lane's four query rows q0 + 4g .. + 3 are contiguous: one 16-byte read each for lse and delta
        const f32x4 lq4 = *reinterpret_cast<const f32x4*>(lse_s + q0 + g * 4), dl4 = *reinterpret_cast<const f32x4*>(del_s + q0 + g * 4);
        f32x4 sA = {0, 0, 0, 0}, pA = -dl4, sB = {0, 0, 0, 0}, pB = -dl4;      // accumulators start at -delta (row r = query q0 + 4g + r)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 qf = row_frag(Qt, q0, ks, lane, RMAX), df = row_frag(Dt, q0, ks, lane, RMAX);
          sA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kfA[ks], sA, 0, 0, 0);
          pA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vfA[ks], pA, 0, 0, 0);
          if (two) {
            sB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kfB[ks], sB, 0, 0, 0);
            pB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vfB[ks], pB, 0, 0, 0);
          }
        }
        if (plain) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float lq = lq4[r];
            const float a_ = __builtin_amdgcn_exp2f(sA[r] * c2 - lq);
            const float b_ = two ? __builtin_amdgcn_exp2f(sB[r] * c2 - lq) : 0.f;
            pvA[hf][r] = a_; svA[hf][r] = a_ * pA[r];
            pvB[hf][r] = b_; svB[hf][r] = b_ * pB[r];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = q0 + g * 4 + r;
            const float lq = lq4[r];
            const bool clsq = qi == N && f != 0;
            const bool okA = keyA <= N && qi <= N && !(clsq && keyA == N) && rm.sees<TIME>(qi, keyA);
            const bool okB = two && keyB <= N && qi <= N && !(clsq && keyB == N) && rm.sees<TIME>(qi, keyB);
            const float a_ = okA ? __builtin_amdgcn_exp2f(sA[r] * c2 - lq) : 0.f;
            const float b_ = okB ? __builtin_amdgcn_exp2f(sB[r] * c2 - lq) : 0.f;
            pvA[hf][r] = a_; svA[hf][r] = a_ * pA[r];
            pvB[hf][r] = b_; svB[hf][r] = b_ * pB[r];
          }
        }
      }
      const bf16x8 pbA = pack8(pvA[0], pvA[1]), sbA = pack8(svA[0], svA[1]);
      const bf16x8 pbB = pack8(pvB[0], pvB[1]), sbB = pack8(svB[0], svB[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 td = tr_frag(Dt, u * 32, dt, lane, RMAX), tq = tr_frag(Qt, u * 32, dt, lane, RMAX);
        dvA[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(td, pbA, dvA[dt], 0, 0, 0);
        dkA[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tq, sbA, dkA[dt], 0, 0, 0);
        if (two) {
          dvB[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(td, pbB, dvB[dt], 0, 0, 0);
          dkB[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tq, sbB, dkB[dt], 0, 0, 0);
        }
      }
    }
    auto put_rows = [&](int t0, const f32x4 (&acc)[4], float mul, int col0) {
      store_rows16(scr, acc, mul, lane, [&](int j) -> bf16* {
        const int q = t0 + j;
        return (q < N && rm.live<TIME>(q)) ? a.dqkv + rm.row<TIME>(q) * a.lddqkv + col0 + h * 64 : nullptr;
      });
    };
    auto put_kv = [&](int key, const f32x4 (&dk)[4], const f32x4 (&dv)[4]) {
      const int t0 = key - (lane & 15);
      if (t0 < N) {                                        // wave-uniform: the tile holds a patch row
        if constexpr (TIME) {
          if (t0 == 0) put_rows(0, dq_keep, a.scale, 0);   // the patch tile of the mini problem: its dQ waited for the tile buffer
        }
        put_rows(t0, dk, a.scale, a.D);
        put_rows(t0, dv, 1.0f, 2 * a.D);
      }
      if (key == N) {
        if constexpr (TIME) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) { cls_dk[dt] += dk[dt]; cls_dv[dt] += dv[dt]; }
        } else {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              atomicAdd(side + 64 + dt * 16 + g * 4 + r, dk[dt][r] * a.scale);
              atomicAdd(side + 128 + dt * 16 + g * 4 + r, dv[dt][r]);
            }
        }
      }
    };
    put_kv(keyA, dkA, dvA);
    if (two) put_kv(keyB, dkB, dvB);
  }
  }   // group walk
  if constexpr (TIME) {
    if ((lane & 15) == (N & 15)) {                       // the lanes that own row N of its tile
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          atomicAdd(side + dt * 16 + g * 4 + r, cls_dq[dt][r] * a.scale);
          atomicAdd(side + 64 + dt * 16 + g * 4 + r, cls_dk[dt][r] * a.scale);
          atomicAdd(side + 128 + dt * 16 + g * 4 + r, cls_dv[dt][r]);
        }
    }
  }
  // Fused finalize: the CLS row's three gradients are complete once every workgroup that feeds side[b][h] has had its
  // atomics acknowledged (vmcnt(0): they are performed at the memory side, where all XCDs meet).  Each workgroup then
  // draws a ticket; the last one swaps the 192 sums out (atomic exchange: a coherent read that also leaves the buffer
  // zero for the next launch), writes the bf16 row and resets the ticket.  No release fence is needed - nothing but
  // atomics carries data between the workgroups - so no L2 write-back is paid per workgroup.
  if (a.done != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last;
    if constexpr (THR > 64) {
      __syncthreads();
      int* flag = reinterpret_cast<int*>(smem);             // every wave is done with the tiles
      if (threadIdx.x == 0) *flag = atomicAdd(a.done + b * a.H + h, 1) == (TIME ? nchunk : a.T) - 1;
      __syncthreads();
      last = *flag;
    } else {
      int t = 0;
      if (threadIdx.x == 0) t = atomicAdd(a.done + b * a.H + h, 1) == (TIME ? nchunk : a.T) - 1;
      last = __shfl(t, 0, 64);
    }
    if (last) {
      for (int t = threadIdx.x; t < 192; t += THR) {
        const float v = atomicExch(side + t, 0.f);
        a.dqkv[cls_row * a.lddqkv + (t >> 6) * a.D + h * 64 + (t & 63)] = f2bf(v);
      }
      if (threadIdx.x == 0) atomicExch(a.done + b * a.H + h, 0);
    }
  }
}

template <int NKT, bool BIG, int WIDE, bool TIME = false, int TT = 0, bool CLSQ = false>
__global__ __launch_bounds__(WIDE == 1 ? 1024 : WIDE == 3 ? 64 : 512, WIDE == 2 ? 4 : 2) void attn_space_bwd_kernel(SpaceArgs2 aa) {
  const bool second = (int)blockIdx.x >= aa.n0;            // workgroup-uniform
  attn_space_bwd_body<NKT, BIG, WIDE, TIME, TT, CLSQ>(aa.s[second ? 1 : 0], (int)blockIdx.x - (second ? aa.n0 : 0));
}
// TIME backward of two clips with DIFFERENT frame counts in one launch (the one-frame object clip of the OA models beside the T-frame
// clip): each clip runs the body compiled for its own frame count (a run-time T costs the kernel 16 spilled registers).
template <int TTA, int TTB>
__global__ __launch_bounds__(64, 2) void attn_time_bwd2_kernel(SpaceArgs2 aa) {
  if ((int)blockIdx.x >= aa.n0) attn_space_bwd_body<2, false, 3, true, TTB>(aa.s[1], (int)blockIdx.x - aa.n0);
  else attn_space_bwd_body<2, false, 3, true, TTA>(aa.s[0], (int)blockIdx.x);
}

// dqkv[cls row(b)][which*D + h*64 + d] = bf16(side[b][h][which][d]); side is left ZERO again, ready for the next
// backward launch (saves a memset launch per attention on the backward chain)
__global__ void attn_cls_finalize_kernel(float* side, bf16* dqkv, int lddqkv, int B, int H, int D, size_t cls_row0) {
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int t = threadIdx.x;                       // 192 threads: which = t / 64, d = t % 64
  if (t >= 192) return;
  const float v = side[((size_t)b * H + h) * 192 + t];
  side[((size_t)b * H + h) * 192 + t] = 0.f;
  dqkv[(cls_row0 + b) * lddqkv + (t >> 6) * D + h * 64 + (t & 63)] = f2bf(v);
}

static SpaceArgs2 one_clip(const SpaceArgs& a, int blocks) { return SpaceArgs2{{a, a}, blocks, 1}; }
static SpaceArgs2 two_clips(const SpaceArgs& a, const SpaceArgs& b) { return SpaceArgs2{{a, b}, a.B * a.T * a.H, 2}; }
// the clip count is carried explicitly (two clips may share a qkv base pointer)
static int space_blocks(const SpaceArgs2& aa) { return aa.n0 + (aa.nclips == 2 ? aa.s[1].B * aa.s[1].T * aa.s[1].H : 0); }
// frames of 196 / 441 patches (224^2 / 336^2) run the compile-time-N forward kernels (straight-line score pass, round 5)
static bool space_nfix(const SpaceArgs2& aa, int n) { return aa.s[0].N == n && (aa.nclips < 2 || aa.s[1].N == n); }
template <int NKT>
static int launch_fwd(const SpaceArgs2& aa, hipStream_t s) {
  const int lds = 2 * NKT * 16 * 128 + (FWD_THREADS / 64) * SCR_BYTES;
  if constexpr (NKT == 14) {
    if (space_nfix(aa, 196)) {
      OAT_MAX_LDS((attn_space_fwd_kernel<NKT, 196>), lds);
      OAT_LAUNCH((attn_space_fwd_kernel<NKT, 196>), dim3(space_blocks(aa)), dim3(FWD_THREADS), lds, s, aa);
      return check_launch("attn_space_fwd");
    }
  }
  if constexpr (NKT == 28) {
    if (space_nfix(aa, 441)) {                     // 336^2 / 16^2: config 5's frames
      OAT_MAX_LDS((attn_space_fwd_kernel<NKT, 441>), lds);
      OAT_LAUNCH((attn_space_fwd_kernel<NKT, 441>), dim3(space_blocks(aa)), dim3(FWD_THREADS), lds, s, aa);
      return check_launch("attn_space_fwd");
    }
  }
  OAT_MAX_LDS(attn_space_fwd_kernel<NKT>, lds);
  OAT_LAUNCH(attn_space_fwd_kernel<NKT>, dim3(space_blocks(aa)), dim3(FWD_THREADS), lds, s, aa);
  return check_launch("attn_space_fwd");
}
template <int NKT, bool BIG = false, int WIDE = 0, bool CLSQ = false>
static int launch_bwd(const SpaceArgs2& aa, hipStream_t s) {
  const int lds = (BIG ? 2 : 4) * NKT * 16 * 128 + 2 * NKT * 16 * 4 + (WIDE == 1 ? 16 : BWD_THREADS / 64) * SCR_BYTES;
  OAT_MAX_LDS((attn_space_bwd_kernel<NKT, BIG, WIDE, false, 0, CLSQ>), lds);
  OAT_LAUNCH((attn_space_bwd_kernel<NKT, BIG, WIDE, false, 0, CLSQ>), dim3(space_blocks(aa)), dim3(WIDE == 1 ? 1024 : BWD_THREADS), lds, s, aa);
  return check_launch("attn_space_bwd");
}
// 97..223 patches: two 8-wave workgroups per CU on the two-tile layout, one tile per wave; 224..447 patches: 16 waves x one tile.
// (The 8-wave tile-pair kernel and the 16-wave four-tile layout of rounds 3-5 - measured slower, DESIGN section 4 - left the library
// in round 6.)

// time-attention backward through the MFMA kernel: one single-wave workgroup per (sample, group of 16 / T positions, head)
template <int TT>
static int launch_time_bwd(const SpaceArgs& a, int blocks, int lds, hipStream_t s) {
  OAT_LAUNCH((attn_space_bwd_kernel<2, false, 3, true, TT>), dim3(blocks), dim3(64), lds, s, one_clip(a, blocks));
  return check_launch("attn_time_bwd_mfma");
}
constexpr int g_time_gpw = 4;        // position groups per workgroup
int attn_time_bwd_mfma(const void* qkv, int ldqkv, const void* out, int ldo, const float* lse, const void* dout, int lddo,
                       void* dqkv, int lddqkv, float* cls_side, int B, int T, int N, int H, int D, float scale, hipStream_t s,
                       int* done) {
  SpaceArgs a{(const bf16*)qkv, ldqkv, (bf16*)out, ldo, (float*)lse, (const bf16*)dout, lddo, (bf16*)dqkv, lddqkv,
              cls_side, B, T, N, H, D, scale, g_time_gpw > 0 ? g_time_gpw : 4, done};
  const int G = 16 / T, ngrp = (N + G - 1) / G, nchunk = (ngrp + a.gpw - 1) / a.gpw;
  const int lds = 2 * (4 * 17 * 128 + 2 * 32 * 4), blocks = B * nchunk * H;      // two problem buffers
  switch (T) {
    case 1: return launch_time_bwd<1>(a, blocks, lds, s);
    case 2: return launch_time_bwd<2>(a, blocks, lds, s);
    case 4: return launch_time_bwd<4>(a, blocks, lds, s);
    case 8: return launch_time_bwd<8>(a, blocks, lds, s);
    case 16: return launch_time_bwd<16>(a, blocks, lds, s);
    default: return launch_time_bwd<0>(a, blocks, lds, s);      // any other T <= 16: runtime row arithmetic
  }
}

// Two clips of different frame counts in ONE launch: workgroups [0, n0) walk clip 0, the rest clip 1 (each clip with its own rows, CLS
// side buffer and tickets).  Built for clip 0 = ONE frame (the object frame: alone a 32 us launch on a fraction of the GPU, twelve times
// per step) and clip 1 = 2, 4, 8 or 16 frames; anything else runs as two launches.
int attn_time_bwd_mfma_clips(const SpaceArgs& c0, const SpaceArgs& c1, hipStream_t s) {
  SpaceArgs a[2] = {c0, c1};
  int blocks[2];
  for (int i = 0; i < 2; ++i) {
    a[i].gpw = g_time_gpw > 0 ? g_time_gpw : 4;
    const int G = 16 / a[i].T, ngrp = (a[i].N + G - 1) / G, nchunk = (ngrp + a[i].gpw - 1) / a[i].gpw;
    blocks[i] = a[i].B * nchunk * a[i].H;
  }
  const int lds = 2 * (4 * 17 * 128 + 2 * 32 * 4);
  const SpaceArgs2 aa{{a[0], a[1]}, blocks[0], 2};
  const dim3 grid(blocks[0] + blocks[1]);
  if (a[0].T == 1 && a[1].T == 2) OAT_LAUNCH((attn_time_bwd2_kernel<1, 2>), grid, dim3(64), lds, s, aa);
  else if (a[0].T == 1 && a[1].T == 4) OAT_LAUNCH((attn_time_bwd2_kernel<1, 4>), grid, dim3(64), lds, s, aa);
  else if (a[0].T == 1 && a[1].T == 8) OAT_LAUNCH((attn_time_bwd2_kernel<1, 8>), grid, dim3(64), lds, s, aa);
  else if (a[0].T == 1 && a[1].T == 16) OAT_LAUNCH((attn_time_bwd2_kernel<1, 16>), grid, dim3(64), lds, s, aa);
  else {
    for (int i = 0; i < 2; ++i) {
      const int rc = attn_time_bwd_mfma(a[i].qkv, a[i].ldqkv, a[i].out, a[i].ldo, a[i].lse, a[i].dout, a[i].lddo, a[i].dqkv, a[i].lddqkv, a[i].cls_side,
                                        a[i].B, a[i].T, a[i].N, a[i].H, a[i].D, a[i].scale, s, a[i].done);
      if (rc) return rc;
    }
    return 0;
  }
  return check_launch("attn_time_bwd_mfma_clips");
}

static int pick_nkt(int N) {
  const int need = (N + 1 + 31) / 32 * 2;     // even number of 16-key tiles
  const int opts[] = {2, 4, 8, 14, 28};
  for (int o : opts) if (o >= need) return o;
  return -1;
}

}  // namespace oat

using namespace oat;

static int space_fwd(const SpaceArgs2& aa, int N, int H, int D, void* stream) {
  if (D != H * 64) { set_error("attn_space: head_dim must be 64"); return -3; }
  const int nkt = pick_nkt(N);
  if (nkt < 0) { set_error("attn_space: patches per frame > 447 not supported by this build"); return -3; }
  hipStream_t s = (hipStream_t)stream;
  switch (nkt) {
    case 2: return launch_fwd<2>(aa, s);
    case 4: return launch_fwd<4>(aa, s);
    case 8: return launch_fwd<8>(aa, s);
    case 14: return launch_fwd<14>(aa, s);
    default: return launch_fwd<28>(aa, s);
  }
}
extern "C" int oat_attn_space_fwd(const void* qkv, int ldqkv, void* out, int ldo, float* lse, int B, int T, int N,
                                  int H, int D, float scale, void* stream) {
  SpaceArgs a{(const bf16*)qkv, ldqkv, (bf16*)out, ldo, lse, nullptr, 0, nullptr, 0, nullptr, B, T, N, H, D, scale};
  return space_fwd(one_clip(a, B * T * H), N, H, D, stream);
}

// cls_side: fp32 [B, H, 3, 64], must be ZERO on entry (caller memsets); it receives the CLS row's
// dq/dk/dv partial sums.  Call oat_attn_cls_finalize afterwards to write them into dqkv (it zeroes cls_side again).
static int space_bwd2(const SpaceArgs2& aa, int N, int H, int D, void* stream, bool clsq = false) {
  if (D != H * 64) { set_error("attn_space: head_dim must be 64"); return -3; }
  const int nkt = pick_nkt(N);
  if (nkt < 0) { set_error("attn_space: patches per frame > 447 not supported by this build"); return -3; }
  hipStream_t s = (hipStream_t)stream;
  switch (nkt) {
    case 2: return launch_bwd<2>(aa, s);
    case 4: return launch_bwd<4>(aa, s);
    case 8: return launch_bwd<8>(aa, s);
    // clsq: the CLS-query-only instance (bit-identical to the full launch when every patch query has dO = 0 and lse = +inf; frames
    // of fewer than 97 patches - test sizes - run the full kernel, which adds the same exact zeros)
    case 14: return clsq ? launch_bwd<14, true, 2, true>(aa, s) : launch_bwd<14, true, 2>(aa, s);
    default: return clsq ? launch_bwd<28, true, 1, true>(aa, s) : launch_bwd<28, true, 1>(aa, s);
  }
}
static int space_bwd(const void* qkv, int ldqkv, const void* out, int ldo, const float* lse, const void* dout, int lddo,
                     void* dqkv, int lddqkv, float* cls_side, int* done, int B, int T, int N, int H, int D, float scale,
                     void* stream) {
  SpaceArgs a{(const bf16*)qkv, ldqkv, (bf16*)out, ldo, (float*)lse, (const bf16*)dout, lddo, (bf16*)dqkv, lddqkv,
              cls_side, B, T, N, H, D, scale, 0, done};
  return space_bwd2(one_clip(a, B * T * H), N, H, D, stream);
}
// Two clips of the same geometry (N, H, D, leading dimensions) in one launch each way - the object frame and the video clip
// of the OA models (oa_model_global_local.py:170, oa_model_region_mem.py:120).  OatAttnClip: see include/oatrans_hip.h.
struct OatAttnClip { const void* qkv; void* out; float* lse; const void* dout; void* dqkv; float* cls_side; int* done; int B, T; };
extern "C" int oat_attn_space_fwd_clips(const OatAttnClip* c, int n_clips, int ldqkv, int ldo, int N, int H, int D, float scale,
                                        void* stream) {
  if (!c || n_clips < 1 || n_clips > 2) { set_error("attn_space_fwd_clips: one or two clips"); return -4; }
  SpaceArgs a[2];
  for (int i = 0; i < n_clips; ++i)
    a[i] = SpaceArgs{(const bf16*)c[i].qkv, ldqkv, (bf16*)c[i].out, ldo, c[i].lse, nullptr, 0, nullptr, 0, nullptr, c[i].B, c[i].T, N, H, D, scale};
  return space_fwd(n_clips == 2 ? two_clips(a[0], a[1]) : one_clip(a[0], a[0].B * a[0].T * H), N, H, D, stream);
}
extern "C" int oat_attn_space_bwd_clips(const OatAttnClip* c, int n_clips, int ldqkv, int ldo, int lddo, int lddqkv, int N, int H,
                                        int D, float scale, int cls_query_only, void* stream) {
  if (!c || n_clips < 1 || n_clips > 2) { set_error("attn_space_bwd_clips: one or two clips"); return -4; }
  SpaceArgs a[2];
  for (int i = 0; i < n_clips; ++i) {
    if (!c[i].done || !c[i].cls_side) { set_error("attn_space_bwd_clips: every clip needs its cls_side and ticket buffers"); return -4; }
    a[i] = SpaceArgs{(const bf16*)c[i].qkv, ldqkv, (bf16*)c[i].out, ldo, c[i].lse, (const bf16*)c[i].dout, lddo, (bf16*)c[i].dqkv, lddqkv,
                     c[i].cls_side, c[i].B, c[i].T, N, H, D, scale, 0, c[i].done};
  }
  return space_bwd2(n_clips == 2 ? two_clips(a[0], a[1]) : one_clip(a[0], a[0].B * a[0].T * H), N, H, D, stream, cls_query_only != 0);
}
// TIME attention backward (with the fused CLS-row finalize) of two clips in one launch (1 + {2, 4, 8, 16} frames; other pairs: two launches)
extern "C" int oat_attn_time_bwd_clips(const OatAttnClip* c, int n_clips, int ldqkv, int ldo, int lddo, int lddqkv, int N, int H,
                                       int D, float scale, void* stream) {
  if (!c || n_clips != 2) { set_error("attn_time_bwd_clips: two clips"); return -4; }
  if (D != H * 64) { set_error("attn_time: head_dim must be 64"); return -3; }
  SpaceArgs a[2];
  for (int i = 0; i < 2; ++i) {
    if (!c[i].done || !c[i].cls_side) { set_error("attn_time_bwd_clips: every clip needs its cls_side and ticket buffers"); return -4; }
    if (c[i].T < 1 || c[i].T > 16) { set_error("attn_time_bwd_clips: 1 <= T <= 16"); return -3; }
    a[i] = SpaceArgs{(const bf16*)c[i].qkv, ldqkv, (bf16*)c[i].out, ldo, c[i].lse, (const bf16*)c[i].dout, lddo, (bf16*)c[i].dqkv, lddqkv,
                     c[i].cls_side, c[i].B, c[i].T, N, H, D, scale, 0, c[i].done};
  }
  return attn_time_bwd_mfma_clips(a[0], a[1], (hipStream_t)stream);
}
extern "C" int oat_attn_space_bwd(const void* qkv, int ldqkv, const void* out, int ldo, const float* lse,
                                  const void* dout, int lddo, void* dqkv, int lddqkv, float* cls_side, int B, int T,
                                  int N, int H, int D, float scale, void* stream) {
  return space_bwd(qkv, ldqkv, out, ldo, lse, dout, lddo, dqkv, lddqkv, cls_side, nullptr, B, T, N, H, D, scale, stream);
}
// As oat_attn_space_bwd followed by oat_attn_cls_finalize, in ONE launch: `done` = int [B, H], zero on entry, zero on exit.
extern "C" int oat_attn_space_bwd_fin(const void* qkv, int ldqkv, const void* out, int ldo, const float* lse,
                                      const void* dout, int lddo, void* dqkv, int lddqkv, float* cls_side, int* done,
                                      int B, int T, int N, int H, int D, float scale, void* stream) {
  if (!done) { set_error("attn_space_bwd_fin: null ticket buffer"); return -4; }
  return space_bwd(qkv, ldqkv, out, ldo, lse, dout, lddo, dqkv, lddqkv, cls_side, done, B, T, N, H, D, scale, stream);
}

extern "C" int oat_attn_cls_finalize(float* cls_side, void* dqkv, int lddqkv, int B, int T, int N, int H, int D,
                                     void* stream) {
  OAT_LAUNCH(attn_cls_finalize_kernel, dim3(B * H), dim3(192), 0, (hipStream_t)stream, cls_side, (bf16*)dqkv,
                     lddqkv, B, H, D, (size_t)B * T * N);
  return check_launch("attn_cls_finalize");
}
