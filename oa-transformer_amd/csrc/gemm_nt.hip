// bf16 MFMA GEMM, "NT" form:  C[M,N] = A[M,K] * B[N,K]^T  (+ fused epilogue)
//
// Replaces the ATen linear calls on the hot path of the reference:
//   qkv / proj   /root/reference/OATrans/model/video_transformer.py:102,133
//   fc1 / fc2    /root/reference/OATrans/model/video_transformer.py:46-50
//   vid_proj/txt_proj  /root/reference/OATrans/model/oa_model.py:68-74
// and their data-gradients in backward (B = W^T shadow).
//
// gfx950 design: 256x256x64 tile, 512 threads = 8 waves (2x4), each wave a 128x64 sub-tile = 8x4 MFMA
// 16x16x32 tiles (128 fp32 accumulators / lane); a 128x128x64 / 4-wave configuration serves small
// problems.  Both operands are K-contiguous, so A and B fragments are one ds_read_b128 each.  Tiles are
// staged with 16-byte global_load_lds (no VGPR round trip) into an LDS ring (see the kernel comment); the
// LDS image is lane-linear, so the bank-conflict swizzle (chunk ^= (row >> 1) & 7 inside a
// 128-byte row) is applied to the per-lane GLOBAL source address and again on the read.
// Epilogues.  fp32 outputs (residual / row-modulo table / bf16 copy) and the two legacy GELU forms: MFMA operands
// swapped (acc = mfma(Bfrag, Afrag)) so a lane owns 4 consecutive output columns of one row, then a transpose through
// per-wave LDS scratch to full 128-byte lines.  bf16 outputs (EPI_BF16 / EPI_GELU_GRAD / EPI_MUL_AUX, 98 % of the
// launches of a training step): DIRECT - operands not swapped, B rows permuted on their way into LDS so that a lane's
// four tiles hold 4 consecutive columns and 16 consecutive lanes store one 128-byte line straight from the
// accumulators (see DIRECT in the kernel).
#include "gemm.h"
#include <type_traits>
#include <atomic>

namespace oat {


// MFMA with the accumulator pinned to AGPRs.  With 256 accumulator registers per lane (128x128 wave tile) hipcc
// otherwise selects the VGPR form and shuttles every result through v_accvgpr_write / _read.
OAT_DEV void mfma_agpr(f32x4& c, const bf16x8 a, const bf16x8 b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// physical byte offset of logical 16B-chunk `lc` (0..7) of row r in a [rows][64 bf16] tile
OAT_DEV int swz(int r, int lc) { return r * 128 + ((lc ^ ((r >> 1) & 7)) << 4); }

// WM x WN waves, each owning TM x TN MFMA tiles of 16x16:  <2,2,4,4> = 128x128 / 4 waves,
// <2,4,8,4> = 256x256 / 8 waves (one workgroup per CU, 160 KB LDS).
//
// Staging pipeline: B (weights: re-read by every workgroup, L2-resident) is double-buffered; A (activations:
// streamed once from HBM / Infinity Cache, ~2 us away under load) gets NSA = 3 stages in the 256x256
// configuration (3 x 32 KB + 2 x 32 KB = all 160 KB of LDS), so its loads are issued TWO K-steps ahead.
// The LDS-DMA loads are issued from inline asm (the compiler would otherwise drain them with vmcnt(0)
// before the first ds_read) and published by a counted s_waitcnt + raw s_barrier.
template <int EPI, int WM, int WN, int TM, int TN, int NSA, bool SPREAD>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_kernel(GemmArgs g) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NW = WM * WN;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;       // glds instructions per wave per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware tile order: consecutive ids on one XCD share the A row-panel (M-major walk of N).
  const int ntn = (g.N + BN - 1) / BN;
  const int ntm = (g.M + BM - 1) / BM;
  // Persistent launch (gridDim.x < number of tiles): workgroup w walks tiles w, w + gridDim.x, ... - exactly the
  // tiles the hardware would have dispatched to that slot - so the store drain of one tile overlaps the
  // prologue loads of the next instead of an idle CU waiting for the old workgroup to retire.
  const int nwg = ntm * ntn;
  auto remap = [&](int t) {
    const int q = nwg >> 3, r = nwg & 7, xcd = t & 7, idx = t >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective
  };
  // PREF: while a tile's epilogue runs (scratch in the upper 96 KB of the ring), the first two A stages of the
  // workgroup's NEXT tile are already streaming into A slots 0 and 1.
  constexpr bool PREF = NSA == 3 && EPI != EPI_DGELU;
  // DIRECT: bf16 outputs are stored straight from the accumulators, with no transpose through LDS, no scratch and no
  // wave barriers.  The MFMA operands are NOT swapped here (lane l holds rows 4*(l>>4)+r, column l&15 of a 16x16 tile)
  // and the B rows (= output columns) are PERMUTED on their way into LDS: tile j, MFMA column c of a wave's 64-column
  // group is output column 4*c + j.  A lane then owns 4 consecutive columns of a row across its TN = 4 tiles (one 8-B
  // store) and the 16 consecutive lanes of a row write one full 128-byte line.
  constexpr bool DIRECT = TN == 4 && (EPI == EPI_BF16 || EPI == EPI_GELU_GRAD || EPI == EPI_MUL_AUX);
  bool prefetched = false;
  // Tile scheduling: static - workgroup w walks tiles w, w + gridDim.x, ... (remapped so that the column tiles of one A panel share
  // an XCD's L2).  (The dynamic, counter-driven walk of round 1 - a late-starting workgroup takes fewer tiles - lost 5-15 % to the
  // counter's round trip and left the library in round 6.)
  int static_tile = blockIdx.x;
  auto fetch = [&]() -> int {
    const int t = static_tile;
    static_tile += gridDim.x;
    return t < nwg ? remap(t) : -1;
  };
  int bid = fetch();
  while (bid >= 0) {
  const int tm = bid / ntn, tn = bid % ntn;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging addresses: wave w stages GA (GB) slabs of 8 rows of A (B), one glds each
  const int srow = lane >> 3;                   // row within an 8-row slab
  uint32_t a_src[GA], b_src[GB];                // per-lane BYTE offsets into A / B (the K offset rides in the scalar base)
#pragma unroll
  for (int i = 0; i < GA; ++i) {
    const int r = (wave * GA + i) * 8 + srow;
    const int lc = (lane & 7) ^ ((r >> 1) & 7);            // inverse swizzle on the source side
    a_src[i] = (uint32_t)(((size_t)min(m0 + r, g.M - 1) * g.lda + lc * 8) * 2);   // clamp: ragged M reads a valid row
  }
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    const int r = (wave * GB + i) * 8 + srow;
    const int lc = (lane & 7) ^ ((r >> 1) & 7);
    const int rg = DIRECT ? ((r & ~63) | ((r & 15) << 2) | ((r >> 4) & 3)) : r;   // LDS row r <- B row rg
    b_src[i] = (uint32_t)(((size_t)min(n0 + rg, g.N - 1) * g.ldb + lc * 8) * 2);
  }
  char* const sA0 = smem;
  char* const sB0 = smem + NSA * A_BYTES;
  auto stageA = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < GA; ++i) glds16_asm_so(g.A + k0, a_src[i], sA0 + buf * A_BYTES + (wave * GA + i) * 1024);
  };
  auto stageB = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < GB; ++i) glds16_asm_so(g.B + k0, b_src[i], sB0 + buf * B_BYTES + (wave * GB + i) * 1024);
  };

  const int nk = g.K / BK;
  const int frow = lane & 15, fk = lane >> 4;
  // bias of this lane's output columns, requested before the main loop: fetched in the epilogue it costs one
  // exposed global-load latency per tile (22 of 193 us at N = 2304, K = 768)
  f32x4 bias_v[TN];
  if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU_DUAL || EPI == EPI_GELU_GRAD || EPI == EPI_MUL_AUX) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 16 + (DIRECT ? frow * 4 : j * 16 + fk * 4);     // DIRECT: only bias_v[0] is used
      bias_v[j] = (g.bias && col < g.N) ? *reinterpret_cast<const f32x4*>(g.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // DIRECT: the bias is the accumulators' initial value (lane column 4 * frow + j is the same for every row it owns),
  // so the epilogue is conversions and stores only
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float b0 = DIRECT ? bias_v[0][j & 3] : 0.f;
      acc[i][j] = f32x4{b0, b0, b0, b0};
    }
  const bool have_a = PREF && prefetched;         // A(0), A(1) were issued during the previous tile's epilogue
  if (!have_a) stageA(0, 0);
  stageB(0, 0);
  if (NSA == 3 && nk > 1 && !have_a) stageA(1, BK);
  int abuf = 0;                                   // kt % NSA
  for (int kt = 0; kt < nk; ++kt) {
    // in flight after this wait: only A(kt+1) (issued last), everything older - A(kt), B(kt) - has landed
    // (after a prefetch B(0) is the youngest load, so step 0 waits for everything)
    if (NSA == 3 && kt + 1 < nk && !(have_a && kt == 0)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // stage kt visible; everyone is done reading stage kt-1
    const bool moreB = kt + 1 < nk, moreA = NSA == 3 ? kt + 2 < nk : kt + 1 < nk;
    const int bbuf_n = (kt + 1) & 1;
    const int abuf_n = NSA == 3 ? (abuf == 0 ? 2 : abuf - 1) : (kt + 1) & 1;      // (kt + 2) % 3 | (kt + 1) % 2
    const int ka_n = (NSA == 3 ? kt + 2 : kt + 1) * BK;
    if (!SPREAD) {
      if (moreB) stageB(bbuf_n, (kt + 1) * BK);
      if (moreA) stageA(abuf_n, ka_n);
    }
    const char* sa = sA0 + abuf * A_BYTES;
    const char* sb = sB0 + (kt & 1) * B_BYTES;
    abuf = abuf + 1 == NSA ? 0 : abuf + 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(sb + swz(wn * TN * 16 + j * 16 + frow, kk * 4 + fk));
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(sa + swz(wm * TM * 16 + i * 16 + frow, kk * 4 + fk));
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        if constexpr (SPREAD) {
          // one LDS-DMA piece per SLOT MFMA groups: a piece blocks its wave's issue port for 60-180 cycles;
          // spread through the MFMA stream that time is hidden behind matrix work instead of stalling both
          // waves of a SIMD together right after the barrier
          constexpr int SLOTS = 2 * TM, NP = GA + GB, EVERY = SLOTS / NP;
          const int slot = kk * TM + i;
          if (EVERY > 0 && slot % EVERY == EVERY - 1 && slot / EVERY < NP) {
            const int pc = slot / EVERY;
            __builtin_amdgcn_sched_barrier(0);
            if (pc < GB) {
              if (moreB) glds16_asm_so(g.B + (kt + 1) * BK, b_src[pc], sB0 + bbuf_n * B_BYTES + (wave * GB + pc) * 1024);
            } else {
              if (moreA) glds16_asm_so(g.A + ka_n, a_src[pc - GB], sA0 + abuf_n * A_BYTES + (wave * GA + pc - GB) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  }

  // ---- epilogue.  Non-DIRECT forms: each lane owns C[row = .. + (lane & 15)][col = .. + (lane >> 4) * 4 + 0..3]
  // (4 columns = 8..16 B): storing that directly gives 32-64 B fragments per row and an
  // issue-bound store tail (measured: 40 % of a K=768 GEMM).  Instead every wave transposes
  // 64-row chunks through its own LDS scratch (the staging buffers are free now) and writes
  // FULL 128-byte lines, 16 B per lane; residual / aux loads use the same coalesced shape.
  // DIRECT forms need none of that (their lane layout already is line-shaped) and use no LDS here.
  __syncthreads();
  prefetched = false;
  const int bid_next = fetch();            // the LDS staging data is dead here: the broadcast word is safe
  if constexpr (PREF) {
    if (bid_next >= 0 && nk > 1) {
      const int m0n = (bid_next / ntn) * BM;
#pragma unroll
      for (int i = 0; i < GA; ++i) {
        const int r = (wave * GA + i) * 8 + srow;
        const int lc = (lane & 7) ^ ((r >> 1) & 7);
        a_src[i] = (uint32_t)(((size_t)min(m0n + r, g.M - 1) * g.lda + lc * 8) * 2);
      }
      stageA(0, 0);
      stageA(1, BK);
      prefetched = true;
    }
  }
  // per-wave scratch: 64 rows x 128 B, 16-byte chunk c of row r at ((c ^ (r & 7)) << 4).  The earlier padded layout
  // (144-B pitch) measured SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = 0.31: its ds_read_b128 lane groups (rows r..r+3,
  // chunk halves 0-3 / 4-7) collided; with an unpadded pitch they do not, and the XOR keeps the column-strided writes
  // at their 2-pass minimum.
  constexpr int EP = 128;
  char* ep = smem + (PREF ? 2 * A_BYTES : 0) + wave * (64 * EP);
  auto sc = [](int r, int byte_off) { return r * 128 + ((((byte_off >> 4) ^ (r & 7)) << 4) | (byte_off & 15)); };
  // 8-byte writes (ds_write_b64: 16-lane groups = 16 rows of one column): rows r and r + 8 share a chunk under `sc`,
  // so rows with bit 3 set use the OTHER half of the chunk; the reader swaps the halves back (compile-time per pass)
  auto sc8 = [](int r, int byte_off) {
    return r * 128 + ((((byte_off >> 4) ^ (r & 7)) << 4) | ((((byte_off >> 3) ^ (r >> 3)) & 1) << 3));
  };
  const int rrow = lane >> 3, rch = lane & 7;     // read phase: 8 rows x 8 chunks of 16 B
  const int wrow0 = m0 + wm * TM * 16;
  static_assert(TN % 4 == 0, "epilogue works on 64-column groups of the wave tile");
  const int wcol00 = n0 + wn * TN * 16;
  if constexpr (DIRECT) {
    const int col = wcol00 + frow * 4;
    auto finish = [&](const f32x4 v, const bf16x4 a, bf16x4& o, bf16x4& o2) {
      if constexpr (EPI == EPI_GELU_GRAD) {
        f32x4 gl, dg;                       // the arithmetic of the ping-pong kernel's epilogue, bit for bit (common.h: gelu_pair)
        gelu_quad(v, gl, dg);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = f2bf(dg[e]);
          o2[e] = f2bf(gl[e]);
        }
      } else if constexpr (EPI == EPI_MUL_AUX) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e] * bf2f(a[e]));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
      }
    };
    if (m0 + BM <= g.M && n0 + BN <= g.N) {
      // interior tile (all but the last row panel): no bounds checks, and every address is a wave-uniform 64-bit base
      // (scalar registers, advanced with scalar adds) plus ONE 32-bit lane offset per matrix
      const size_t t0 = (size_t)wrow0;
      char* const ob = reinterpret_cast<char*>(g.out) + (t0 * g.ldc + wcol00) * 2;
      char* const ob2 = EPI == EPI_GELU_GRAD ? reinterpret_cast<char*>(g.out2) + (t0 * g.ld2 + wcol00) * 2 : nullptr;
      const char* const ab = EPI == EPI_MUL_AUX ? reinterpret_cast<const char*>(g.aux) + (t0 * g.ldaux + wcol00) * 2 : nullptr;
      const uint32_t lo = (uint32_t)(fk * 4 * g.ldc + frow * 4) * 2;
      const uint32_t lo2 = EPI == EPI_GELU_GRAD ? (uint32_t)(fk * 4 * g.ld2 + frow * 4) * 2 : 0;
      const uint32_t la = EPI == EPI_MUL_AUX ? (uint32_t)(fk * 4 * g.ldaux + frow * 4) * 2 : 0;
      bf16x4 an[4] = {}, ac[4] = {};
      if constexpr (EPI == EPI_MUL_AUX) {
#pragma unroll
        for (int r = 0; r < 4; ++r) an[r] = *reinterpret_cast<const bf16x4*>(ab + (size_t)(uint32_t)(r * g.ldaux * 2) + la);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (EPI == EPI_MUL_AUX) {        // the saved derivative of row group i + 1 is requested one group ahead
#pragma unroll
          for (int r = 0; r < 4; ++r) ac[r] = an[r];
          if (i + 1 < TM) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              an[r] = *reinterpret_cast<const bf16x4*>(ab + (size_t)(uint32_t)(((i + 1) * 16 + r) * g.ldaux * 2) + la);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 v = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
          bf16x4 o, o2;
          finish(v, ac[r], o, o2);
          const uint32_t rr = (uint32_t)(i * 16 + r);
          *reinterpret_cast<bf16x4*>(ob + (size_t)(rr * (uint32_t)g.ldc * 2) + lo) = o;
          if constexpr (EPI == EPI_GELU_GRAD) *reinterpret_cast<bf16x4*>(ob2 + (size_t)(rr * (uint32_t)g.ld2 * 2) + lo2) = o2;
        }
      }
    } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wrow0 + i * 16 + fk * 4 + r;
        const f32x4 v = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
        if (row < g.M && col < g.N) {
          bf16x4 o, o2, a = {};
          if constexpr (EPI == EPI_MUL_AUX) a = *reinterpret_cast<const bf16x4*>(g.aux + (size_t)row * g.ldaux + col);
          finish(v, a, o, o2);
          if constexpr (EPI == EPI_GELU_GRAD) *reinterpret_cast<bf16x4*>((bf16*)g.out2 + (size_t)row * g.ld2 + col) = o2;
          *reinterpret_cast<bf16x4*>((bf16*)g.out + (size_t)row * g.ldc + col) = o;
        }
      }
    }
  } else {
#pragma unroll
  for (int rh = 0; rh < TM / 4; ++rh)
#pragma unroll
  for (int cg = 0; cg < TN / 4; ++cg) {
    const int wcol0 = wcol00 + cg * 64;
    if constexpr (EPI == EPI_GELU_GRAD) {
      // GELU and its derivative share one erf / exp evaluation on the fp32 pre-activation.  gelu'(h) goes through the
      // wave's bf16 scratch first; gelu(h) waits, packed, in 32 registers and takes the same route afterwards.
      bf16x4 glv[4][4];
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bf16x4 o;
            if (pass == 0) {
              const f32x4 v = acc[rh * 4 + i][cg * 4 + j] + bias_v[cg * 4 + j];
              f32x4 gl, dg;
              gelu_quad(v, gl, dg);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                o[e] = f2bf(dg[e]);
                glv[i][j][e] = f2bf(gl[e]);
              }
            } else {
              o = glv[i][j];
            }
            *reinterpret_cast<bf16x4*>(ep + sc8(i * 16 + frow, (j * 16 + fk * 4) * 2)) = o;
            if (pass == 0) __builtin_amdgcn_sched_barrier(0);     // one tile's erf / exp chains at a time: no spills
          }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int lr = it * 8 + rrow;
          const int row = wrow0 + rh * 64 + lr, col = wcol0 + rch * 8;
          typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2;
          const u64x2 raw = *reinterpret_cast<const u64x2*>(ep + sc(lr, rch * 16));
          const u64x2 fixed = (it & 1) ? u64x2{raw[1], raw[0]} : raw;      // see the bf16 path below
          if (row < g.M && col < g.N) {
            if (pass == 0) *reinterpret_cast<u64x2*>((bf16*)g.out + (size_t)row * g.ldc + col) = fixed;
            else *reinterpret_cast<u64x2*>((bf16*)g.out2 + (size_t)row * g.ld2 + col) = fixed;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU_DUAL || EPI == EPI_MUL_AUX) {
      // bf16 scratch: 64 rows x 64 cols
      bf16x8 auxv[8];
      auto aux_load = [&](int it) {
        const int row = wrow0 + rh * 64 + it * 8 + rrow, col = wcol0 + rch * 8;
        if (row < g.M && col < g.N) auxv[it] = *reinterpret_cast<const bf16x8*>(g.aux + (size_t)row * g.ldaux + col);
      };
      if constexpr (EPI == EPI_MUL_AUX) {         // first half requested before the transpose: its latency hides behind it
#pragma unroll
        for (int it = 0; it < 2; ++it) aux_load(it);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = acc[rh * 4 + i][cg * 4 + j] + bias_v[cg * 4 + j];
          const bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
          *reinterpret_cast<bf16x4*>(ep + sc8(i * 16 + frow, (j * 16 + fk * 4) * 2)) = o;
        }
      __builtin_amdgcn_wave_barrier();
      if constexpr (EPI == EPI_MUL_AUX) {         // second half: this chunk's accumulators are dead now
#pragma unroll
        for (int it = 2; it < 8; ++it) aux_load(it);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int lr = it * 8 + rrow;
        const int row = wrow0 + rh * 64 + lr, col = wcol0 + rch * 8;
        bf16x8 hv;
        {
          typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2;
          const u64x2 raw = *reinterpret_cast<const u64x2*>(ep + sc(lr, rch * 16));
          const u64x2 fixed = (it & 1) ? u64x2{raw[1], raw[0]} : raw;      // rows with bit 3 set: halves swapped (renaming)
          hv = __builtin_bit_cast(bf16x8, fixed);
        }
        if (row < g.M && col < g.N) {
          if constexpr (EPI == EPI_MUL_AUX) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = f2bf(bf2f(hv[e]) * bf2f(auxv[it][e]));
          }
          *reinterpret_cast<bf16x8*>((bf16*)g.out + (size_t)row * g.ldc + col) = hv;
          if constexpr (EPI == EPI_GELU_DUAL) {
            // GELU is evaluated on the bf16-rounded pre-activation so that backward (which only
            // has the saved bf16 h) differentiates exactly the function that forward applied.
            bf16x8 a;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = f2bf(gelu_f(bf2f(hv[e])));
            *reinterpret_cast<bf16x8*>((bf16*)g.out2 + (size_t)row * g.ld2 + col) = a;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    } else if constexpr (EPI == EPI_DGELU && BM == 256) {
      // fp32 scratch, 64 rows x 64 cols (8 waves x 16 KB fit the 160 KB ring): a lane then owns 8
      // consecutive columns, so the saved pre-activation is read and dH written in full 128-byte lines
      constexpr int EW = 256;                           // unpadded; 16-byte chunk c of row r at (c ^ (r & 7)): conflict-free
      char* ew = smem + wave * (64 * EW);                // for the b128 writes (8 rows x 1 column) and the b128 reads
      auto sw = [](int r, int byte_off) { return r * 256 + ((((byte_off >> 4) ^ (r & 7))) << 4); };
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(ew + sw(i * 16 + frow, (j * 16 + fk * 4) * 4)) = acc[rh * 4 + i][cg * 4 + j];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int lr = it * 8 + rrow;
        const int row = wrow0 + rh * 64 + lr, col = wcol0 + rch * 8;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(ew + sw(lr, rch * 32));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(ew + sw(lr, rch * 32 + 16));
        if (row < g.M && col < g.N) {
          const bf16x8 h = *reinterpret_cast<const bf16x8*>(g.aux + (size_t)row * g.ldaux + col);
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = f2bf((v0[e] + (g.bias ? g.bias[col + e] : 0.f)) * dgelu_f(bf2f(h[e])));
            o[4 + e] = f2bf((v1[e] + (g.bias ? g.bias[col + 4 + e] : 0.f)) * dgelu_f(bf2f(h[4 + e])));
          }
          *reinterpret_cast<bf16x8*>((bf16*)g.out + (size_t)row * g.ldc + col) = o;
        }
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      // fp32 scratch: 64 rows x 32 cols, two column halves
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            *reinterpret_cast<f32x4*>(ep + sc(i * 16 + frow, (jj * 16 + fk * 4) * 4)) = acc[rh * 4 + i][cg * 4 + ch * 2 + jj];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int lr = it * 8 + rrow;
          const int row = wrow0 + rh * 64 + lr, col = wcol0 + ch * 32 + rch * 4;
          f32x4 v = *reinterpret_cast<const f32x4*>(ep + sc(lr, rch * 16));
          if (row < g.M && col < g.N) {
            if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + col);
            if constexpr (EPI == EPI_F32 || EPI == EPI_F32_BF16) {
              if (g.resid) {
                const int rr = g.resid_mod > 0 ? (row + g.row0) % g.resid_mod : row;
                v += *reinterpret_cast<const f32x4*>(g.resid + (size_t)rr * g.ldr + col);
              }
              *reinterpret_cast<f32x4*>((float*)g.out + (size_t)row * g.ldc + col) = v;
              if constexpr (EPI == EPI_F32_BF16) {
                const bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                *reinterpret_cast<bf16x4*>((bf16*)g.out2 + (size_t)row * g.ld2 + col) = o;
              }
            } else {   // EPI_DGELU
              const bf16x4 h = *reinterpret_cast<const bf16x4*>(g.aux + (size_t)row * g.ldaux + col);
              const bf16x4 o = {f2bf(v[0] * dgelu_f(bf2f(h[0]))), f2bf(v[1] * dgelu_f(bf2f(h[1]))),
                                f2bf(v[2] * dgelu_f(bf2f(h[2]))), f2bf(v[3] * dgelu_f(bf2f(h[3])))};
              *reinterpret_cast<bf16x4*>((bf16*)g.out + (size_t)row * g.ldc + col) = o;
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  }
  if (!DIRECT && bid_next >= 0) {          // the epilogue scratch lives in the staging ring the next tile refills
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  bid = bid_next;
  }   // tile loop
}

static int cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

template <int EPI, int WM, int WN, int TM, int TN, int NSA, bool SPREAD>
static int launch_cfg(const GemmArgs& g, const GemmTune& t, hipStream_t s) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int LDS = (NSA * BM + 2 * BN) * BK * 2;
  const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
  OAT_MAX_LDS((gemm_nt_kernel<EPI, WM, WN, TM, TN, NSA, SPREAD>), LDS);
  int grid = ntm * ntn;
  if (BM == 256 && t.grid != 0xffff) {        // persistent: one workgroup per CU walks the tiles (+1.8 % per step)
    const int slots = t.grid > 0 ? t.grid : cu_count();
    if (grid > slots) grid = slots;
  }
  OAT_LAUNCH((gemm_nt_kernel<EPI, WM, WN, TM, TN, NSA, SPREAD>), dim3(grid), dim3(WM * WN * 64), LDS, s, g);
  return check_launch("gemm_nt");
}

static int pp_slots(const GemmTune& t) { return t.grid == 0xffff ? 0x7fffffff : t.grid > 0 ? t.grid : cu_count(); }

// the 256x256 configuration of a launch: ping-pong kernel where it applies, else the lockstep kernel
template <int EPI>
static int launch_big(const GemmArgs& g, const GemmTune& t, hipStream_t s) {
  if ((t.variant == 4 || t.variant == 0) && pp_supported(EPI, g)) return launch_pp(EPI, g, pp_slots(t), t, s);
  return launch_cfg<EPI, 2, 4, 8, 4, 3, true>(g, t, s);
}

template <int EPI>
static int launch(const GemmArgs& g, const GemmTune& t, hipStream_t s) {
  // 256x256 tiles need enough of them: below ~100 tiles (the object clip of the OA variants, 6304 rows x 768 columns = 75
  // tiles on 256 CUs) the 128x128 configuration's 300 quarter tiles finish 20-30 % sooner (scripts/dev/small_m_gemm.py)
  const bool big = t.variant >= 2 || (t.variant == 0 && g.M >= 4096 && g.N % 256 == 0 &&
                                      ((g.M + 255) / 256) * (g.N / 256) * 5 >= cu_count() * 2);
  // (Round 6: the tail split - the last, mostly empty round of 256x256 tiles re-tiled as 128x128, measured equal with the ping-pong
  // kernel - the 4-wave pipelined 256x256 configuration, the dynamic tile walk and the split-K workspace left the library.)
  if (big) return launch_big<EPI>(g, t, s);
  return launch_cfg<EPI, 2, 2, 4, 4, 2, false>(g, t, s);
}

}  // namespace oat

// tune: bits 0-7 kernel choice (GemmTune::variant), bits 16-17 the 224-row-tile mode (0 = default: auto, 1 = never, 2 = always),
// bits 18-25 the band walk (0 = default: auto, 1 = off, n + 1 = n column tiles per group).  grid: GemmTune::grid.
static oat::GemmTune decode_tune(int tune, int grid) {
  oat::GemmTune t;
  t.variant = tune & 0xff;
  t.grid = grid;
  const int m = (tune >> 16) & 3, b = (tune >> 18) & 0xff;
  t.m224 = m == 0 ? 1 : m == 1 ? 0 : 2;
  t.band = b == 0 ? -1 : b - 1;
  return t;
}

extern "C" int oat_gemm_nt(const void* A, const void* B, int M, int N, int K, int lda, int ldb,
                           int epi, void* out, int ldc, void* out2, int ld2,
                           const float* bias, const float* resid, int ldr, int resid_mod,
                           const void* aux, int ldaux, int tune, int grid, void* stream) {
  using namespace oat;
  if (grid < 0 || grid > 0xffff) { set_error("gemm_nt: grid must be 0 (one workgroup per CU), a workgroup count, or 0xffff (one per tile)"); return -3; }
  const GemmTune t = decode_tune(tune, grid);
  if (t.variant != 0 && t.variant != 1 && t.variant != 2 && t.variant != 4) { set_error("gemm_nt: unknown kernel choice in `tune`"); return -3; }
  const int h_u8 = (epi >> 8) & 1;          // epi | 0x100: the GELU-derivative tensor is 8-bit fixed point (gemm.h: h_u8)
  epi &= 0xff;
  if (M <= 0 || N <= 0 || K <= 0) { set_error("gemm_nt: empty problem"); return -1; }
  if (K % BK != 0) { set_error("gemm_nt: K must be a multiple of 64"); return -2; }
  if (N % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0) {
    set_error("gemm_nt: N, lda, ldb, ldc must be multiples of 8"); return -3;
  }
  if (!A || !B || !out) { set_error("gemm_nt: null pointer"); return -4; }
  GemmArgs g{(const bf16*)A, (const bf16*)B, M, N, K, lda, ldb, out, ldc, out2, ld2,
             bias, resid, ldr, resid_mod, (const bf16*)aux, ldaux, 0, 0, nullptr, nullptr};
  hipStream_t s = (hipStream_t)stream;
  if (h_u8) {
    g.h_u8 = 1;
    if ((epi != EPI_GELU_GRAD && epi != EPI_MUL_AUX) || !pp_supported(epi, g)) {
      set_error("gemm_nt: the 8-bit GELU derivative needs EPI_GELU_GRAD / EPI_MUL_AUX on a shape the ping-pong kernel covers");
      return -3;
    }
    if ((epi == EPI_GELU_GRAD && !out2) || (epi == EPI_MUL_AUX && !aux)) { set_error("gemm_nt: missing out2 / aux"); return -4; }
    // the 8-bit derivative tensor is blocked ([row / 16][col / 64][lane][16 B], gemm_nt_pp.hip): a dense [round_up(M, 16), N] byte array
    if ((epi == EPI_GELU_GRAD ? ldc : ldaux) != N) { set_error("gemm_nt: the 8-bit GELU derivative is a dense blocked tensor (ld == N)"); return -3; }
    return launch_pp(epi, g, pp_slots(t), t, s);
  }
  switch (epi) {
    case EPI_BF16: return launch<EPI_BF16>(g, t, s);
    case EPI_F32: return launch<EPI_F32>(g, t, s);
    case EPI_GELU_DUAL:
      if (!out2) { set_error("gemm_nt: EPI_GELU_DUAL needs out2"); return -4; }
      return launch<EPI_GELU_DUAL>(g, t, s);
    case EPI_DGELU:
      if (!aux) { set_error("gemm_nt: EPI_DGELU needs aux"); return -4; }
      return launch<EPI_DGELU>(g, t, s);
    case EPI_F32_BF16:
      if (!out2) { set_error("gemm_nt: EPI_F32_BF16 needs out2"); return -4; }
      return launch<EPI_F32_BF16>(g, t, s);
    case EPI_GELU_GRAD:
      if (!out2) { set_error("gemm_nt: EPI_GELU_GRAD needs out2"); return -4; }
      return launch<EPI_GELU_GRAD>(g, t, s);
    case EPI_MUL_AUX:
      if (!aux) { set_error("gemm_nt: EPI_MUL_AUX needs aux"); return -4; }
      return launch<EPI_MUL_AUX>(g, t, s);
    default: set_error("gemm_nt: unknown epilogue"); return -5;
  }
}
