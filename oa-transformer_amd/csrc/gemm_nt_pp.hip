// bf16 MFMA GEMM, "NT" form, two-wave-group ("ping-pong") 256x256x64 kernel:  C[M,N] = A[M,K] * B[N,K]^T (+ epilogue)
//
// Same contract and same arithmetic as gemm_nt_kernel<EPI,2,4,8,4,...> (gemm_nt.hip) for the three bf16-output
// epilogues that make up 98 % of the GEMM launches of a training step (EPI_BF16, EPI_GELU_GRAD, EPI_MUL_AUX;
// replaces the nn.Linear calls at /root/reference/OATrans/model/video_transformer.py:46-50,102,133 and their
// data gradients).  What differs is the K loop.
//
// The lockstep kernel runs all 8 waves of a workgroup through "barrier - 24 ds_read_b128 - 64 MFMA" together, so the
// two waves that share a SIMD want the matrix pipe at the same time and wait for the LDS at the same time (PMC: MFMA
// pipe busy 58 % of the K loop).  Here the 8 waves form two groups (waves 0-3 / 4-7: wave w and w + 4 share a SIMD)
// that run the SAME instruction stream ONE barrier interval apart:
//
//     interval      t        t+1       t+2       t+3    ...
//     group 0      L1        M1        L2        M2            L = LDS -> register fragments of the next quadrant
//     group 1      M4'       L1        M1        L2                + 2 LDS-DMA pieces of a later K-tile + counted vmcnt
//                                                              M = 16 MFMAs (one 64x32 quadrant x K = 64), s_setprio 1
//
// so on every SIMD one wave feeds the matrix pipe while the other one talks to the LDS and the memory pipeline.
// A K-tile (64 k) is 4 quadrants = 8 intervals per wave; quadrant order (a0,b0) (a1,b0) (a1,b1) (a0,b1) keeps both A
// half-fragments (2 x 32 VGPRs) and two B half-fragments (2 x 16 VGPRs) live and reads 8 / 8 / 4 / 4 ds_read_b128 in
// L1..L4 (L4 pre-reads b0 of the NEXT K-tile).
//
// LDS: two K-tile buffers of A (2 x 32 KB) and B (2 x 32 KB) + the bias vector = 144 KB.  A buffer is not recycled as a
// whole: the 64-row / 32-row region a quadrant read last is dead one interval later (the trailing group has read it
// too), so the region b0 | a0 | a1 | b1 of K-tile s is refilled with K-tile s + 2 in L1 | L2 | L3 | L4 of K-tile s
// itself.  Every piece is therefore issued ~1.75 K-tiles (14 intervals) before its first read, and after the prologue
// every wait is the same `s_waitcnt vmcnt(12)`: the 12 youngest LDS-DMA pieces may still fly (derivation at `endL`).
// The stream of K-tiles is continuous across the tiles a persistent workgroup walks, so the first two K-tiles of the
// next tile are already in flight or landed when the epilogue of a tile stores its accumulators, and those stores
// drain under the next tile's K loop (vmcnt retires loads and stores in issue order on gfx950; the first six waits
// after an interior epilogue allow its NST stores on top of the 12 pieces).
#include "gemm.h"
#include "fp8.h"
#include <type_traits>

namespace oat {

namespace {

constexpr int PP_A1 = 32768, PP_B0 = 65536, PP_BIAS = 131072;
constexpr int PP_MAXN = 4096;                       // bias vector kept in LDS
constexpr int PP_FLAG = PP_BIAS + PP_MAXN * 4;      // one word: the split-K arrival count, broadcast to the workgroup
constexpr int PP_LDS = PP_FLAG + 16;

// compile-time configuration of the kernel.  PRIO | LGKM | BONUS are the shipped schedule (their ablation builds, the two-phase
// K-tile kernel, wide stores and the split-K of the last round were measured in rounds 2-5 and left the tree in round 6).
enum : int { PPF_PRIO = 1, PPF_LGKM = 4, PPF_BONUS = 8,
             PPF_F8 = 128,                      // OCP e4m3 operands (see below)
             PPF_A_BF8 = 256,                   // (unused since round 6: the A operand as e5m2)
             PPF_HU8 = 512,                     // the saved GELU derivative travels as 8-bit fixed point (see HU8_*)
             PPF_M224 = 2048,                   // 224-row tiles (see TM in the kernel)
             PPF_BAND = 4096 };                 // band-grouped per-XCD tile walk (see `tile_of` in the kernel)

// gelu'(h) lies in [-0.129, 1.129].  As bf16 it costs 2 bytes per element to write (fc1 forward) and to read back (fc2 data
// gradient) - 308 MB per launch each way, all of it on top of a GEMM that is otherwise MFMA-bound.  Stored as
// q = round((g' + 0.135) * 255 / 1.27) in ONE byte the absolute error is <= 0.0025, the size of bf16's own rounding
// error for values near 1 (2^-9 = 0.002), at half the traffic.
constexpr float HU8_OFF = 0.135f, HU8_SCALE = 255.f / 1.27f, HU8_INV = 1.27f / 255.f;
OAT_DEV f32x4 hu8_unpack(uint32_t w) {
  return f32x4{(float)(w & 255u) * HU8_INV - HU8_OFF, (float)((w >> 8) & 255u) * HU8_INV - HU8_OFF,
               (float)((w >> 16) & 255u) * HU8_INV - HU8_OFF, (float)(w >> 24) * HU8_INV - HU8_OFF};
}

// PPF_F8: OCP fp8 (e4m3) operands, per-tensor scaled.  A K-tile is still 128 BYTES of every row - now 128 k - so the
// staging stream, the LDS layout, the swizzle, the region refill and every wait count are those of the bf16 kernel; a
// lane's MFMA operand is 32 bytes of its row (lane l of v_mfma_f32_16x16x128_f8f6f4 holds row l & 15 and the k block
// l >> 4; scripts/dev/fp8_probe - WHICH 32 k a block holds is free as long as A and B agree) and one MFMA does the
// work of four bf16 ones in twice the time.  The accumulators hold the product of the QUANTISED operands; the epilogue
// multiplies by the two dequantisation scales (device scalars, GemmArgs::dq_a / dq_b), the bias enters pre-divided.
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
OAT_DEV i32x8 f8_operand(const bf16x8 lo, const bf16x8 hi) {
  return __builtin_shufflevector(__builtin_bit_cast(i32x4, lo), __builtin_bit_cast(i32x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

// In-place MFMA (D = C register block).  hipcc otherwise gives the second k-half's MFMAs fresh destination registers
// (it renames around the dependent pair), which costs up to 32 VGPRs in a 32-MFMA interval and spills at the 256 limit.
// Operands come from ds_read (the compiler still places the lgkmcnt wait in front of this statement); the first
// non-MFMA reader of an accumulator is the epilogue, an `s_nop` block and a barrier later.
OAT_DEV void mfma_inplace(f32x4& c, const bf16x8 a, const bf16x8 b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// Output tiles leave with the NON-TEMPORAL hint (`global_store ... nt`): a persistent launch's 256 CUs store 32 MB per round into 32 MB of
// L2 that also holds the A / B panels the K loops stream; written as ordinary (write-back, retained) lines the tiles wait in the L2 and
// leave in eviction bursts, and - vmcnt retiring in issue order - every later LDS-DMA wait of the K loop stands behind them.  Streamed
// out, the K = 768 launches run 8-11 % faster (N768 63 -> 56.5, N2304 157 -> 145, N3072 214 -> 192 us), K >= 2304 unchanged
// (scripts/dev/store_policy_time.py; sc1 / sc0 sc1 write-through measured equal to plain).
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
template <class T>
OAT_DEV void st_nt(T* p, const T v) { __builtin_nontemporal_store(v, p); }

// Epilogue of one 256x256 tile (no LDS, no barriers): lane (fk, frow) owns rows 16 i + 4 fk + r and the 4 consecutive
// columns 4 frow + j of its wave tile, so 16 consecutive lanes store one 128-byte line per row.  Returns whether the
// tile was an interior one (then exactly NST store instructions were issued per lane, see the kernel).
template <int EPI, bool F8 = false, bool HU8 = false, int TM = 256>
OAT_DEV bool pp_epilogue(const GemmArgs& g, const f32x4 (&acc)[8][4], int m0, int n0, int wm, int wn, int lane, float dq = 1.f) {
  // TM: rows of the tile (256, or 224: wave rows of 112 = 7 row groups of 16, acc[7] unused)
  constexpr int WR = TM / 2, NI = WR / 16;
  // HU8: the derivative tensor (out of EPI_GELU_GRAD, aux of EPI_MUL_AUX) is one byte per element in a blocked layout (below)
  uint32_t d8 = 0;                                                // HU8 + EPI_GELU_GRAD: the 4 derivative bytes of the last finish()
  // lane-constant store offsets are derived from an opaque copy of the lane id: hoisted out of the tile loop they would
  // sit in VGPRs through the K loop, which has none to spare
  asm volatile("" : "+v"(lane));
  const int frow = lane & 15, fk = lane >> 4;
  const int wrow0 = m0 + wm * WR, wcol00 = n0 + wn * 64;
  constexpr bool Q8 = F8 && (EPI == EPI_GELU_GRAD || EPI == EPI_MUL_AUX);   // optional fp8 copy for the next GEMM:
                                                       // e4m3 of gelu(h) (-> fc2) or e5m2 of the fc2 data gradient (-> fc1 dgrad)
  uint8_t* const o8 = Q8 ? reinterpret_cast<uint8_t*>(g.out8) : nullptr;
  const float q8 = (Q8 && o8) ? g.q_out[0] : 0.f;
  float m8 = 0.f;
  uint32_t w8 = 0;                                     // the 4 quantised values of the last finish() call
  // EPI_GELU_GRAD: gelu / gelu' of a row group are formed COLUMN-wise before its rows are stored - an accumulator's four registers
  // (rows r = 0..3 of one column) are two aligned register pairs, so the packed fp32 math of gelu_pair (common.h) runs on them in
  // place; taken row-wise (four columns = four different accumulators) every packed operand would first be gathered by a v_mov.
  // Gc[j][r] = gelu, Hc[j][r] = the derivative: 8-bit code as a float (HU8: affine map applied, packed) or gelu' itself.
  f32x4 Gc[4], Hc[4];
  auto gelu_group = [&](int i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v = acc[i][j];
      if constexpr (F8) v *= dq;
      f32x2 g01, g23, h01, h23;
      gelu_pair(f32x2{v[0], v[1]}, g01, h01);
      gelu_pair(f32x2{v[2], v[3]}, g23, h23);
      if constexpr (HU8) {
        constexpr float B = (HU8_OFF + 0.5f) * HU8_SCALE;
        h01 = pk_fma(h01, pk_bc(HU8_SCALE), pk_bc(B));
        h23 = pk_fma(h23, pk_bc(HU8_SCALE), pk_bc(B));
      } else {
        h01 += 0.5f;
        h23 += 0.5f;
      }
      Gc[j] = f32x4{g01[0], g01[1], g23[0], g23[1]};
      Hc[j] = f32x4{h01[0], h01[1], h23[0], h23[1]};
    }
  };
  auto finish = [&](f32x4 v, const f32x4 a, bf16x4& o, bf16x4& o2, int r) {
    if constexpr (F8 && EPI != EPI_GELU_GRAD) v *= dq;
    if constexpr (EPI == EPI_GELU_GRAD) {
      const float gq[4] = {Gc[0][r], Gc[1][r], Gc[2][r], Gc[3][r]};
#pragma unroll
      for (int e = 0; e < 4; ++e) o2[e] = f2bf(gq[e]);
      if constexpr (HU8) {
        // v_cvt_pk_u8_f32 rounds to nearest, saturates to [0, 255] and inserts the byte (the code of a finite h lies in [1.2, 253.8])
        uint32_t w = 0;
        w = __builtin_amdgcn_cvt_pk_u8_f32(Hc[0][r], 0, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(Hc[1][r], 1, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(Hc[2][r], 2, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(Hc[3][r], 3, w);
        d8 = w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(Hc[e][r]);
      }
      if constexpr (Q8) {
        if (o8) {
          m8 = fmaxf(fmaxf(m8, fmaxf(fabsf(gq[0]), fabsf(gq[1]))), fmaxf(fabsf(gq[2]), fabsf(gq[3])));
          w8 = pack_fp8x4(gq[0] * q8, gq[1] * q8, gq[2] * q8, gq[3] * q8);
        }
      }
    } else if constexpr (EPI == EPI_MUL_AUX) {
      float gq[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gq[e] = v[e] * a[e];
        o[e] = f2bf(gq[e]);
      }
      if constexpr (Q8) {
        if (o8) {
          m8 = fmaxf(fmaxf(m8, fmaxf(fabsf(gq[0]), fabsf(gq[1]))), fmaxf(fabsf(gq[2]), fabsf(gq[3])));
          w8 = pack_bf8x4(gq[0] * q8, gq[1] * q8, gq[2] * q8, gq[3] * q8);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
    }
  };
  const bool interior = m0 + TM <= g.M;
  // HU8: the derivative tensor is BLOCKED - [row / 16][col / 64][lane][16 bytes]: the 16 rows x 64 columns a wave's row group i
  // covers are one 1 KB block in which lane (fk, frow) owns 16 consecutive bytes, its rows 4 fk + r times its columns 4 frow + j
  // in [r][j] order.  Producer (EPI_GELU_GRAD) and consumer (EPI_MUL_AUX) are the only readers of this tensor and both hold exactly
  // these 16 elements per lane and row group, whatever the tile height: one 16-byte access per lane and row group, 1 KB contiguous per
  // wave instruction, instead of four 4-byte accesses that each touched four 64-byte row segments.
  constexpr bool DBLK = HU8 && EPI == EPI_GELU_GRAD, ABLK = HU8 && EPI == EPI_MUL_AUX;
  const uint32_t blks = (DBLK || ABLK) ? ((uint32_t)g.N >> 6) << 10 : 0;                  // bytes from a row group's block to the next
  const size_t blk0 = (DBLK || ABLK) ? (((size_t)(wrow0 >> 4) * ((uint32_t)g.N >> 6) + (uint32_t)(wcol00 >> 6)) << 10) + ((uint32_t)lane << 4) : 0;
  char* const dblk = DBLK ? reinterpret_cast<char*>(g.out) + blk0 : nullptr;
  const char* const ablk = ABLK ? reinterpret_cast<const char*>(g.aux) + blk0 : nullptr;
  if (interior) {
    const size_t t0 = (size_t)wrow0;
    char* const ob = DBLK ? nullptr : reinterpret_cast<char*>(g.out) + (t0 * g.ldc + wcol00) * 2;
    char* const ob2 = EPI == EPI_GELU_GRAD ? reinterpret_cast<char*>(g.out2) + (t0 * g.ld2 + wcol00) * 2 : nullptr;
    const char* const ab = (EPI == EPI_MUL_AUX && !ABLK) ? reinterpret_cast<const char*>(g.aux) + (t0 * g.ldaux + wcol00) * 2 : nullptr;
    const uint32_t lo = DBLK ? 0 : (uint32_t)(fk * 4 * g.ldc + frow * 4) * 2;
    const uint32_t lo2 = EPI == EPI_GELU_GRAD ? (uint32_t)(fk * 4 * g.ld2 + frow * 4) * 2 : 0;
    const uint32_t la = (EPI == EPI_MUL_AUX && !ABLK) ? (uint32_t)(fk * 4 * g.ldaux + frow * 4) * 2 : 0;
    auto load_aux = [&](const char* ptr) -> f32x4 {
      const bf16x4 t = *reinterpret_cast<const bf16x4*>(ptr);
      return f32x4{bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])};
    };
    // EPI_MUL_AUX + HU8: ALL derivative blocks of the tile (NI x 16 bytes per lane) are requested before the first store.
    // vmcnt retires in issue order: with the loads of row group i + 1 issued behind the stores of group i - 1 (the former
    // one-group-ahead scheme) every wait for a derivative word also waited for those stores to be WRITTEN - eight
    // dependent store round trips per tile (259 -> 237 us per launch).  The fragment registers of the K loop are dead
    // here, so 32 VGPRs are free.
    uint4 aw[NI] = {};
    f32x4 an[4] = {}, ac[4] = {};
    if constexpr (ABLK) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {           // read once: non-temporal (−3 % alone, equal in the step)
        const u32x4 t_ = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ablk + (size_t)((uint32_t)i * blks)));
        aw[i] = uint4{t_[0], t_[1], t_[2], t_[3]};
      }
      asm volatile("" ::: "memory");
    } else if constexpr (EPI == EPI_MUL_AUX) {
#pragma unroll
      for (int r = 0; r < 4; ++r) an[r] = load_aux(ab + (size_t)(uint32_t)(r * g.ldaux * 2) + la);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if constexpr (ABLK) {
        ac[0] = hu8_unpack(aw[i].x); ac[1] = hu8_unpack(aw[i].y); ac[2] = hu8_unpack(aw[i].z); ac[3] = hu8_unpack(aw[i].w);
      } else if constexpr (EPI == EPI_MUL_AUX) {        // bf16 derivative (non-default): row group i + 1 is requested one group ahead
#pragma unroll
        for (int r = 0; r < 4; ++r) ac[r] = an[r];
        if (i + 1 < NI) {
#pragma unroll
          for (int r = 0; r < 4; ++r) an[r] = load_aux(ab + (size_t)(uint32_t)(((i + 1) * 16 + r) * g.ldaux * 2) + la);
        }
      }
      uint32_t dw[4] = {};
      if constexpr (EPI == EPI_GELU_GRAD) gelu_group(i);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 v = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
        bf16x4 o, o2;
        finish(v, ac[r], o, o2, r);
        const uint32_t rr = (uint32_t)(i * 16 + r);
        if constexpr (DBLK) dw[r] = d8;
        else st_nt(reinterpret_cast<bf16x4*>(ob + (size_t)(rr * (uint32_t)g.ldc * 2) + lo), o);
        if constexpr (EPI == EPI_GELU_GRAD) st_nt(reinterpret_cast<bf16x4*>(ob2 + (size_t)(rr * (uint32_t)g.ld2 * 2) + lo2), o2);
        if constexpr (Q8) {
          if (o8) *reinterpret_cast<uint32_t*>(o8 + (size_t)(wrow0 + i * 16 + fk * 4 + r) * g.ld8 + wcol00 + frow * 4) = w8;
        }
      }
      if constexpr (DBLK) st_nt(reinterpret_cast<u32x4*>(dblk + (size_t)((uint32_t)i * blks)), u32x4{dw[0], dw[1], dw[2], dw[3]});
    }
  } else {
    const int col = wcol00 + frow * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (wrow0 + i * 16 >= g.M) continue;               // wave-uniform: no row of this group exists (its derivative block is not touched)
      uint4 awi = {};
      if constexpr (ABLK) awi = *reinterpret_cast<const uint4*>(ablk + (size_t)((uint32_t)i * blks));
      uint32_t dw[4] = {};
      if constexpr (EPI == EPI_GELU_GRAD) gelu_group(i);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wrow0 + i * 16 + fk * 4 + r;
        const f32x4 v = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
        const bool live = row < g.M;
        bf16x4 o, o2;
        f32x4 a = {};
        if constexpr (ABLK) a = hu8_unpack(r == 0 ? awi.x : r == 1 ? awi.y : r == 2 ? awi.z : awi.w);
        else if constexpr (EPI == EPI_MUL_AUX) {
          if (live) {
            const bf16x4 t = *reinterpret_cast<const bf16x4*>(g.aux + (size_t)row * g.ldaux + col);
            a = f32x4{bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])};
          }
        }
        finish(v, a, o, o2, r);                           // rows >= M of a straddling group: finite values of a re-read row, stored nowhere but in the block
        if constexpr (DBLK) dw[r] = d8;
        if (live) {
          if constexpr (EPI == EPI_GELU_GRAD) *reinterpret_cast<bf16x4*>((bf16*)g.out2 + (size_t)row * g.ld2 + col) = o2;
          if constexpr (!DBLK) *reinterpret_cast<bf16x4*>((bf16*)g.out + (size_t)row * g.ldc + col) = o;
          if constexpr (Q8) {
            if (o8) *reinterpret_cast<uint32_t*>(o8 + (size_t)row * g.ld8 + col) = w8;
          }
        }
      }
      if constexpr (DBLK) *reinterpret_cast<uint4*>(dblk + (size_t)((uint32_t)i * blks)) = uint4{dw[0], dw[1], dw[2], dw[3]};
    }
  }
  if constexpr (Q8) {
    if (o8) amax_commit_wave(m8, g.amax_out);
  }
  return interior;
}

template <int EPI, int FL>
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(GemmArgs g) {
  constexpr bool PRIO = FL & PPF_PRIO, LGKM = FL & PPF_LGKM, BONUS = FL & PPF_BONUS;
  constexpr bool F8 = FL & PPF_F8;
  constexpr int CBSZ = (FL & PPF_A_BF8) ? 1 : 0;                // MFMA format code of A: 0 = e4m3, 1 = e5m2
  constexpr int ESH = F8 ? 0 : 1;                               // log2(bytes per operand element)
  // PPF_M224: 224-row tiles.  A persistent launch takes ceil(tiles / CUs) rounds of one tile time each; at M = 50208, N = 768
  // 256-row tiles are 591 tiles = 2.31 -> 3 rounds, 224-row tiles 675 tiles = 2.64 -> 3 rounds of a tile that is 12.5 %
  // smaller.  The LDS layout stays that of the 256-row tile (wave row wm at LDS rows wm * 128 ..): the second half of a wave
  // row simply has 3 row groups of 16 instead of 4 - 12 MFMAs in two of the four intervals - and the 16 unused LDS rows of
  // each wave row are staged from the wave row's last real row.  Same K order per element: results are bit-identical.
  constexpr int TM = (FL & PPF_M224) ? 224 : 256, WR = TM / 2, NI = WR / 16, NI1 = NI - 4;
  // stores per lane of an interior epilogue (EXACT or an under-count: the head waits of the next tile allow this many ops on top of
  // the 12 pieces): 4 bf16x4 rows per row group, + 4 more (bf16 derivative) or + 1 (the 16-byte block of the 8-bit one) for EPI_GELU_GRAD
  constexpr int NST = (EPI == EPI_GELU_GRAD ? ((FL & PPF_HU8) ? 5 : 8) : 4) * NI;
  constexpr int WB = 12 + NST > 63 ? 63 : 12 + NST;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                      // wave tile: rows wm*128.., columns wn*64..
  const int ntn = g.N >> 8, ntm = (g.M + TM - 1) / TM, nwg = ntm * ntn;
  const int nk = g.K >> (7 - ESH);                                      // K-tiles of 128 bytes per row
  // (ntl: tiles this workgroup walks.  The split-K of a less-than-half-full last round - rounds 2-5, measured equal - left in round 6.)
  const int grid = (int)gridDim.x;
  const int ntl_rm = (nwg - 1 - (int)blockIdx.x) / grid + 1;
  float dq = 1.f, inv_dq = 1.f;
  if constexpr (F8) {
    dq = g.dq_a[0] * g.dq_b[0];
    inv_dq = 1.f / dq;
  }
  // PPF_BAND: band-grouped walk.  The row-major walk gives an XCD (32 CUs, one 4 MB L2) 32 CONSECUTIVE tiles per round:
  // 32 / ntn row panels x all ntn column tiles, i.e. the whole of B every round - at N = 3072, K = 768 that is 4.7 MB of
  // B + 0.9 MB of A per XCD and round, more than the L2 holds, so B is re-fetched from the Infinity Cache / HBM in every
  // round (PMC: 549 MB read per launch for 82 MB of operands).  Here the XCD's chunk of tiles is the same, but the WHOLE
  // row panels inside it are walked band group by band group: a round is (32 / GW) panels x GW column tiles, so per round
  // the XCD needs GW column bands of B (which stay for the next rounds of the same group) and streams 32 / GW row panels of
  // A, each used by GW tiles at the same time.  The partial panels at the two ends of the chunk keep their row-major
  // places.  Same tiles, same per-tile arithmetic: results are bit-identical to the row-major walk.
  // Host guarantees grid % 8 == 0 (a workgroup then stays on one XCD chunk for its whole walk).
  constexpr bool BAND = (FL & PPF_BAND) != 0;
  int bw_s0 = 0, bw_f = 0, bw_mid = 0, bw_pm = 0, bw_P0 = 0, bw_gw = 1;
  if constexpr (BAND) {
    const int q = nwg >> 3, r = nwg & 7, xcd = (int)blockIdx.x & 7;
    bw_s0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;     // first tile (row-major id) of this XCD's chunk
    const int n = q + (xcd < r ? 1 : 0);
    const int c0 = bw_s0 % ntn;
    bw_f = min(c0 == 0 ? 0 : ntn - c0, n);                             // tiles of the partial first panel
    bw_pm = (n - bw_f) / ntn;                                          // whole panels
    bw_mid = bw_pm * ntn;
    bw_P0 = (bw_s0 + bw_f) / ntn;
    bw_gw = g.band;
  }
  struct Tile { int m0, n0; };
  auto tile_of = [&](int t) __attribute__((always_inline)) {
    if constexpr (BAND) {
      const int idx = ((int)blockIdx.x >> 3) + t * (grid >> 3);
      const int L = idx - bw_f;
      const bool mid = L >= 0 && L < bw_mid;
      // row-major place (ends of the chunk)
      const int bid = bw_s0 + idx, tm_r = bid / ntn, tn_r = bid - tm_r * ntn;
      // band-grouped place (whole panels)
      const int Lm = mid ? L : 0, gsz = max(bw_pm * bw_gw, 1);
      const int gi = Lm / gsz, rem = Lm - gi * gsz;
      const int wg = max(min(bw_gw, ntn - gi * bw_gw), 1);
      const int pn = rem / wg;
      const int tm = mid ? bw_P0 + pn : tm_r, tn = mid ? gi * bw_gw + rem - pn * wg : tn_r;
      return Tile{tm * TM, tn << 8};
    } else {
      const int w = (int)blockIdx.x + t * grid;
      const int q = nwg >> 3, r = nwg & 7, xcd = w & 7, idx = w >> 3;
      const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // XCD-contiguous, bijective
      const int tm = bid / ntn;
      return Tile{tm * TM, (bid - tm * ntn) << 8};
    }
  };
  const int ntl = ntl_rm;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  float* const sbias = reinterpret_cast<float*>(smem + PP_BIAS);
  for (int i = tid; i < g.N; i += 512) sbias[i] = g.bias ? g.bias[i] * inv_dq : 0.f;

  // ---- staging.  One LDS-DMA piece = 8 tile rows x 128 B.  Per K-tile a wave stages two pieces of each class:
  //   a0 = A rows {0..63, 128..191} (first halves of the two wave rows), a1 = the other A rows,
  //   b0 = B rows with (row >> 5) even (first halves of the four wave columns), b1 = the others.
  const int srow = lane >> 3;
  const uint32_t c16_0 = (uint32_t)(((lane & 7) ^ (srow >> 1)) << 4);          // 16-byte chunk, source-side swizzle
  const uint32_t lda2 = (uint32_t)g.lda << ESH, ldb2 = (uint32_t)g.ldb << ESH;  // row strides in bytes
  int pa[2], pb[2];                                                             // piece index of (class 0, e)
  uint32_t boff[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int idx = wave * 2 + e;
    pa[e] = idx < 8 ? idx : idx + 8;
    pb[e] = ((idx >> 2) << 3) | (idx & 3);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int r = (pb[e] + c * 4) * 8 + srow;                                 // LDS row <- B row rg (DIRECT epilogue)
      const int rg = (r & ~63) | ((r & 15) << 2) | ((r >> 4) & 3);
      boff[c][e] = (uint32_t)rg * ldb2 + (c16_0 ^ (e << 6));
    }
  }
  // staging cursor: K-tile cs = s + 2 of the stream
  int ckt = 0, ctl = 0, crmax, cnk = nk;       // cnk: K-tiles of the cursor's tile
  const char *ca, *cb;                         // byte cursors: a K-tile is 128 bytes of every row in both formats
  const char* const A0 = reinterpret_cast<const char*>(g.A);
  const char* const B0 = reinterpret_cast<const char*>(g.B);
  uint32_t lda2c = lda2, bmask = ~0u;          // zeroed once the stream is exhausted (see `advance`)
  {
    const Tile t = tile_of(0);
    ca = A0 + (size_t)t.m0 * lda2;
    cb = B0 + (size_t)t.n0 * ldb2;
    crmax = g.M - 1 - t.m0;
  }
  // Past the last K-tile of the stream the cursor stays where it is and the pieces degenerate to re-reads of ONE
  // 128-byte line per operand into regions nobody reads any more: the op stream - and with it every vmcnt count -
  // stays uniform to the end, for 2 K-tiles of dummy pieces per workgroup and launch.
  auto advance = [&]() __attribute__((always_inline)) {
    ++ckt;
    ca += 128;
    cb += 128;
    if (ckt == cnk) {                           // selects, not branches: every path assigns every cursor variable
      const bool more = ctl + 1 < ntl;
      ctl += more ? 1 : 0;
      const Tile t = tile_of(ctl);
      ckt = more ? 0 : cnk - 1;
      ca = more ? A0 + (size_t)t.m0 * lda2 : ca - 128;
      cb = more ? B0 + (size_t)t.n0 * ldb2 : cb - 128;
      crmax = g.M - 1 - t.m0;
      lda2c = more ? lda2c : 0u;
      bmask = more ? bmask : 0x7fu;
    }
  };
  auto stageA = [&](int cls, int buf) __attribute__((always_inline)) {        // cls 0 = a0, 1 = a1
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int p = pa[e] + cls * 8;
      const int lr = p * 8 + srow;                                              // LDS row; tile row = wave row * WR + row in it
      const int tr = TM == 256 ? lr : (lr >> 7) * WR + min(lr & 127, WR - 1);
      const uint32_t r = (uint32_t)min(tr, crmax);                              // ragged M: re-read a valid row
      const uint32_t off = __umul24(r, lda2c) + (c16_0 ^ (e << 6));
      glds16_asm_lds(ca, off, lds0 + buf * PP_A1 + p * 1024);
    }
  };
  auto stageB = [&](int cls, int buf) __attribute__((always_inline)) {        // cls 0 = b0, 1 = b1
#pragma unroll
    for (int e = 0; e < 2; ++e)
      glds16_asm_lds(cb, boff[cls][e] & bmask, lds0 + PP_B0 + buf * PP_A1 + (pb[e] + cls * 4) * 1024);
  };

  // ---- fragment addresses: [rows][64 k] bf16 tiles, 16-byte chunk c of row r at (c ^ ((r >> 1) & 7))
  const int frow = lane & 15, fk = lane >> 4;
  const int sw = (frow >> 1) & 7;
  typedef const __attribute__((address_space(3))) bf16x8* lds_frag;          // 32-bit LDS addresses, immediates fold
  uint32_t pA[2], pB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    // bf16: k = 32 kk + 8 fk ..  fp8: the SAME chunks - lane group fk then holds bytes [16 fk, 16 fk + 16) and [64 + 16 fk, ..)
    // of the 128-k tile instead of 32 consecutive k, for A and B alike, so every product still meets its partner (the MFMA
    // only sums over k); the consecutive assignment (chunks 2 fk, 2 fk + 1) cost 4 LDS bank conflicts per read (PMC)
    const int ch = ((kk * 4 + fk) ^ sw) << 4;
    pA[kk] = lds0 + (wm * 128 + frow) * 128 + ch;
    pB[kk] = lds0 + PP_B0 + (wn * 64 + frow) * 128 + ch;
  }
  bf16x8 fbx[2][2];                          // b0 of the current K-tile (pre-read in L4 of the previous one)
  f32x4 acc[8][4];
  auto readA = [&](bf16x8 (&f)[2][4], int half, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (half == 1 && i >= NI1) continue;              // 224-row tiles: the second half of a wave row has 3 row groups
        f[kk][i] = *(lds_frag)(uintptr_t)(pA[kk] + buf * PP_A1 + (half * 64 + i * 16) * 128);
      }
  };
  auto readB = [&](bf16x8 (&f)[2][2], int half, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        f[kk][j] = *(lds_frag)(uintptr_t)(pB[kk] + buf * PP_A1 + (half * 32 + j * 16) * 128);
  };
  // end of an L interval.  Counted wait: the LDS-DMA stream of a wave is  ... b0(s+2) a0(s+2) a1(s+2) b1(s+2) b0(s+3) ...
  // (2 pieces each, issued in L1..L4 of K-tile s); the region read NEXT was issued six groups before the group this
  // interval just issued, so 12 younger pieces may stay in flight.  Each wave waits for ITS pieces, the barrier
  // publishes them; the first reader comes one interval (other group) or two (same group) later.
  // `bonus` (head pair of a tile only): the NST stores of the previous, INTERIOR tile's epilogue are younger too.
  auto endL = [&](bool bonus) __attribute__((always_inline)) {
    if constexpr (LGKM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (bonus) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WB) : "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma = [&](const bf16x8 (&fa)[2][4], const bf16x8 (&fb)[2][2], int ah, int bh) __attribute__((always_inline)) {
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    if constexpr (F8) {
      // constant zero scale operands select the UNSCALED v_mfma_f32_16x16x128_f8f6f4 (cbsz = blgp = 0: e4m3 x e4m3)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (ah == 1 && i >= NI1) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[ah * 4 + i][bh * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
              f8_operand(fa[0][i], fa[1][i]), f8_operand(fb[0][j], fb[1][j]), acc[ah * 4 + i][bh * 2 + j], CBSZ, 0, 0, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (ah == 1 && i >= NI1) continue;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[ah * 4 + i][bh * 2 + j] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][i], fb[kk][j], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
        }
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: K-tiles 0 and 1 in canonical issue order, then b0(0) into registers
  stageB(0, 0); stageA(0, 0); stageA(1, 0); stageB(1, 0);
  advance();
  stageB(0, 1); stageA(0, 1); stageA(1, 1); stageB(1, 1);
  advance();
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // b0(0), a0(0) landed (K-tile 0's a1, b1 and K-tile 1 may fly)
  __syncthreads();                                       // ... for every wave; also publishes the bias vector
  readB(fbx, 0, 0);
  if (wm == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one interval behind from here on
  __builtin_amdgcn_sched_barrier(0);

  // Two K-tiles (LDS buffers 0 and 1).  HEAD = first pair of a tile: when the previous tile's epilogue was an interior
  // one (`bon`), its NST stores are younger than the awaited pieces in all four waits of the first K-tile and in L1 / L2
  // of the second (the later targets were issued after the stores).
  auto pair = [&](auto head, const bool bon) __attribute__((always_inline)) {
    constexpr bool HEAD = decltype(head)::value;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // buffer u holds K-tile s; its dead regions are refilled with K-tile s + 2 (the cursor)
      bf16x8 fa0[2][4], fa1[2][4], fby[2][2];
      const bool bh = HEAD && bon, bf = HEAD && bon && u == 0;
      readA(fa0, 0, u); stageB(0, u); endL(bh);
      mma(fa0, fbx, 0, 0);
      readA(fa1, 1, u); stageA(0, u); endL(bh);
      mma(fa1, fbx, 1, 0);
      readB(fby, 1, u); stageA(1, u); endL(bf);
      mma(fa1, fby, 1, 1);
      readB(fbx, 0, u ^ 1); stageB(1, u); endL(bf);
      mma(fa0, fby, 0, 1);
      advance();
    }
  };

  bool prev_interior = false;
  for (int tl = 0; tl < ntl; ++tl) {
    const Tile tile = tile_of(tl);
    const int m0 = tile.m0, n0 = tile.n0;
    {
      const f32x4 b = *reinterpret_cast<const f32x4*>(sbias + n0 + wn * 64 + frow * 4);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{b[j], b[j], b[j], b[j]};
    }
    pair(std::true_type{}, BONUS && prev_interior);
    for (int kt = 2; kt < nk; kt += 2) pair(std::false_type{}, false);
    // ---- epilogue (no LDS, no barriers): lane (fk, frow) owns rows 16 i + 4 fk + r and the 4 consecutive columns
    // 4 frow + j of its wave tile, so 16 consecutive lanes store one 128-byte line per row
    prev_interior = pp_epilogue<EPI, F8, (FL & PPF_HU8) != 0, TM>(g, acc, m0, n0, wm, wn, lane, dq);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();  // pairs with group 1's extra barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may outlive the workgroup's LDS allocation
}

template <int EPI, int FL>
int launch_pp_cfg(const GemmArgs& g, int grid_slots, hipStream_t s) {
  OAT_MAX_LDS((gemm_nt_pp_kernel<EPI, FL>), PP_LDS);
  constexpr int TMH = (FL & PPF_M224) ? 224 : 256;
  const int nwg = ((g.M + TMH - 1) / TMH) * (g.N / 256);
  const int grid = nwg < grid_slots ? nwg : grid_slots;
  OAT_LAUNCH((gemm_nt_pp_kernel<EPI, FL>), dim3(grid), dim3(512), PP_LDS, s, g);
  return check_launch("gemm_nt_pp");
}

}  // namespace

// 224- or 256-row tiles?  Time of a persistent launch ~ rounds x tile rows; 224 wins where it saves a whole round's worth.
// mode (GemmTune::m224): 0 never, 1 where they save time (default), 2 always (tests)
static bool pp_prefers_224(const GemmArgs& g, int slots, int mode = 1) {
  if (mode == 0) return false;
  if (mode == 2) return g.M >= 224;
  const long long ntn = g.N / 256;
  auto cost = [&](int tm) {
    const long long tiles = ((g.M + tm - 1) / tm) * ntn, grid = tiles < slots ? tiles : slots;
    return ((tiles + grid - 1) / grid) * tm;
  };
  return g.M >= 224 && cost(224) * 100 < cost(256) * 97;
}

bool pp_f8_supported(int epi, const GemmArgs& g) {
  if (epi != EPI_BF16 && epi != EPI_GELU_GRAD) return false;
  const int nk = g.K / 128;
  return g.K % 128 == 0 && g.N % 256 == 0 && g.N <= PP_MAXN && nk >= 2 && nk % 2 == 0 && g.M >= 256 && g.lda % 16 == 0 &&
         g.ldb % 16 == 0 && g.lda < (1 << 23) && g.ldb < (1 << 23) && g.dq_a && g.dq_b;
}

// fp8 (e4m3 x e4m3) operands: A [M, K] and B [N, K] one byte per element, lda / ldb in elements (= bytes); forward linears only
int launch_pp_f8(int epi, const GemmArgs& g, int grid_slots, hipStream_t s) {
  constexpr int FL = PPF_PRIO | PPF_BONUS | PPF_LGKM | PPF_F8;
  if (!pp_f8_supported(epi, g)) { set_error("gemm_nt_f8: shape / epilogue not covered (K % 256, N % 256, N <= 4096, M >= 256, EPI_BF16 | EPI_GELU_GRAD)"); return -3; }
  if (epi == EPI_GELU_GRAD)
    return g.h_u8 ? launch_pp_cfg<EPI_GELU_GRAD, FL | PPF_HU8>(g, grid_slots, s) : launch_pp_cfg<EPI_GELU_GRAD, FL>(g, grid_slots, s);
  if (pp_prefers_224(g, grid_slots)) return launch_pp_cfg<EPI_BF16, FL | PPF_M224>(g, grid_slots, s);
  return launch_pp_cfg<EPI_BF16, FL>(g, grid_slots, s);
}

// Band-grouped tile walk (PPF_BAND): column tiles per band group.  0 = off (row-major walk), > 0 = that many where the
// launch qualifies, -1 = auto (default).  Measured at M = 50208 (scripts/dev/band_probe.py, us per launch, row-major ->
// groups of 4): N 3072 / K 768 plain bf16 217 -> 202, GELU epilogue (fc1 forward) 287 -> 277, x derivative epilogue (fc2
// data gradient, which also streams the 154 MB derivative tensor) 249 -> 256; N 2304 / K 768 161 -> 163; N 768 shapes
// slower.  The kernel tolerates the L2 misses of the row-major walk well (its LDS-DMA runs 14 intervals ahead), so auto
// only takes the one case that pays: the fc1 forward launch (EPI_GELU_GRAD, N >= 3072, K <= 1024), groups of 4.
// (GemmTune::band carries the request per call.)
static int pp_band_for(const GemmArgs& g, int slots, int tm, int epi, int band) {
  if (band == 0) return 0;
  const long long ntn = g.N / 256, nwg = ((g.M + tm - 1) / tm) * ntn, grid = nwg < slots ? nwg : slots;
  if (grid % 8 != 0 || nwg < 2 * grid) return 0;           // a workgroup must stay on one XCD chunk; >= 2 rounds to gain anything
  int gw = band;
  if (gw < 0) {
    if (epi != EPI_GELU_GRAD || ntn < 12 || ntn % 4 != 0 || g.K > 1024) return 0;
    gw = 4;
  }
  return gw < ntn ? gw : 0;
}

bool pp_supported(int epi, const GemmArgs& g) {
  if (epi != EPI_BF16 && epi != EPI_GELU_GRAD && epi != EPI_MUL_AUX) return false;
  const int nk = g.K / 64;
  return g.N % 256 == 0 && g.N <= PP_MAXN && nk >= 2 && nk % 2 == 0 && g.M >= 256 && g.lda < (1 << 22) &&
         g.ldb < (1 << 22);
}

// The shipped configuration of the ping-pong kernel (PRIO | BONUS | LGKM; the ablation builds of rounds 2-5 - no priority, no stagger,
// the two-phase K-tile kernel, wide stores, split-K of the last round - were measured and left the library in round 6).
int launch_pp(int epi, const GemmArgs& g, int grid_slots, const GemmTune& t, hipStream_t s) {
  constexpr int DEF = PPF_PRIO | PPF_BONUS | PPF_LGKM;   // LGKM: measured free, and it makes the WAR spacing strict
  const bool m224 = pp_prefers_224(g, grid_slots, t.m224);
  GemmArgs a = g;
  a.band = pp_band_for(g, grid_slots, m224 ? 224 : 256, epi, t.band);
  if (epi == EPI_GELU_GRAD || epi == EPI_MUL_AUX) {
    if (g.h_u8) {
      if (epi == EPI_GELU_GRAD) {
        if (a.band) return m224 ? launch_pp_cfg<EPI_GELU_GRAD, DEF | PPF_HU8 | PPF_M224 | PPF_BAND>(a, grid_slots, s)
                                : launch_pp_cfg<EPI_GELU_GRAD, DEF | PPF_HU8 | PPF_BAND>(a, grid_slots, s);
        return m224 ? launch_pp_cfg<EPI_GELU_GRAD, DEF | PPF_HU8 | PPF_M224>(g, grid_slots, s)
                    : launch_pp_cfg<EPI_GELU_GRAD, DEF | PPF_HU8>(g, grid_slots, s);
      }
      if (a.band) return m224 ? launch_pp_cfg<EPI_MUL_AUX, DEF | PPF_HU8 | PPF_M224 | PPF_BAND>(a, grid_slots, s)
                              : launch_pp_cfg<EPI_MUL_AUX, DEF | PPF_HU8 | PPF_BAND>(a, grid_slots, s);
      return m224 ? launch_pp_cfg<EPI_MUL_AUX, DEF | PPF_HU8 | PPF_M224>(g, grid_slots, s)
                  : launch_pp_cfg<EPI_MUL_AUX, DEF | PPF_HU8>(g, grid_slots, s);
    }
    return epi == EPI_GELU_GRAD ? launch_pp_cfg<EPI_GELU_GRAD, DEF>(g, grid_slots, s) : launch_pp_cfg<EPI_MUL_AUX, DEF>(g, grid_slots, s);
  }
  if (a.band) return m224 ? launch_pp_cfg<EPI_BF16, DEF | PPF_M224 | PPF_BAND>(a, grid_slots, s)
                          : launch_pp_cfg<EPI_BF16, DEF | PPF_BAND>(a, grid_slots, s);
  return m224 ? launch_pp_cfg<EPI_BF16, DEF | PPF_M224>(g, grid_slots, s) : launch_pp_cfg<EPI_BF16, DEF>(g, grid_slots, s);
}

}  // namespace oat
