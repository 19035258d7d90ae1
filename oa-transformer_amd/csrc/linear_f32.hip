// fp32 linear layer for SMALL row counts on the exact-f32 matrix instruction (v_mfma_f32_16x16x4_f32):
//     out[M,N] = act( in(A)[M,K] * W[N,K]^T + bias ) (+ resid)
// A, W, bias, resid, out32 are fp32; W is the MASTER weight (no bf16 shadow involved).
//
// Why it exists.  The cosine-similarity matrix has to match the fp32 reference within 1e-3 (north_star), and a
// pipeline whose GEMM operands are all bf16 lands at 0.7-1.7e-3 (embedding error ~6e-3 per tower; measured on the
// reference golden vectors and reproduced by rounding emulation in the CPU oracle).  Two places carry that error and
// both are tiny in FLOPs: the text tower (0.7 % of the step) and the CLS row of the video tower (one row per clip
// whose own rounding errors do not average out, while those of the ~1.5 k patch keys it attends do).  Both run their
// linear layers through this kernel - exact fp32 products, fp32 accumulate, bitwise an fmaf chain per output - beside
// the bf16 MFMA kernels that keep every large GEMM.  /root/reference/OATrans/model/oa_model.py:106-133 (compute_text,
// compute_video, the projections), video_transformer.py:46-50,102,133 for the CLS rows.
//
// Outputs: out32 (optional) and up to two bf16 copies for the bf16 backward path:
//   act 0: out32 = out16 = y            act 1 (GELU): out32 = out16 = gelu(y), out16b = gelu'(y)
//   act 2: ReLU applied to A on load (txt_proj = Sequential(ReLU, Linear), oa_model.py:68-70)
// The next K-step's global loads are in flight while the current one is multiplied.  157 TF/s is the chip's f32 matrix
// peak; this kernel carries <= ~100 GFLOP per step on side streams and is latency-, not throughput-critical.
#include "common.h"
#include <type_traits>

namespace oat {

struct LinArgs {
  const float* A; int lda;
  const float* W; int ldw;
  const float* bias;
  int M, N, K;
  float* out32; int ldo;
  bf16* out16; int ld16;
  bf16* out16b; int ld16b;
  const float* resid; int ldr;
  // PARTS launches (linear_x3_kernel only): the N output columns are `part_n`-wide slices of up to three weight / bias tensors of
  // their own (W, W2, W3: the q | k | v linears of an attention layer, one launch for one [M, 3 part_n] output); 0 = one tensor
  const float* W2; const float* W3; const float* bias2; const float* bias3; int part_n;
};

// The LDS-tiled kernel (many rows):
//   large (the text tower, M = B * L)              128 x 128 x 32 (x 16 when K % 32 != 0), waves 2 x 2, 4 x 4 MFMA tiles per
//       wave (8 LDS reads per 16 MFMAs); 64 x 64 tiles when that would give fewer than 128 workgroups
// KG = 2: the workgroup is TWO such wave quartets (512 threads) that split K between them - each with its own LDS tiles,
// the same barriers - and add their partial tiles through LDS at the end (quartet 1 hands over, quartet 0 finishes: a
// fixed order).  The 64 x 64 configuration runs ONE workgroup per CU (192 workgroups for the text tower's N = 768), i.e. one
// wave per SIMD with nothing to hide its global -> LDS -> MFMA round trips behind; two waves per SIMD do.
template <int ACT, int BM, int BN, int BK, int WM, int WN, int KG = 1>
__global__ __launch_bounds__(256 * KG) void linear_f32_kernel(LinArgs g) {
  constexpr int PITCH = BK + 1, FM = BM / WM / 16, FN = BN / WN / 16, KQ = BK / 4;
  constexpr int LA = BM * KQ / 256, LB = BN * KQ / 256;
  static_assert(WM * WN == 4 && LA >= 1 && LB >= 1, "256 threads per quartet, at least one float4 per thread and operand");
  __shared__ float sAB[KG][(BM + BN) * PITCH];
  __shared__ float sX[KG == 2 ? BM * BN : 1];               // quartet 1's partial tile
  const int kg = KG == 2 ? (int)(threadIdx.x >> 8) : 0;
  float* const sA = sAB[kg];
  float* const sB = sAB[kg] + BM * PITCH;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // K range of this quartet: whole BK steps, the first quartet takes the odd one
  const int nsteps = g.K / BK, my_steps = KG == 2 ? (kg == 0 ? (nsteps + 1) / 2 : nsteps / 2) : nsteps;
  const int kbeg = KG == 2 && kg == 1 ? ((nsteps + 1) / 2) * BK : 0, kend = kbeg + my_steps * BK;
  const int loop_steps = KG == 2 ? (nsteps + 1) / 2 : nsteps;   // both quartets run the same number of barrier pairs
  f32x4 ra[LA], rb[LB];
  auto load = [&](int k0) {
#pragma unroll
    for (int l = 0; l < LA; ++l) {
      const int idx = tid + l * 256, row = idx / KQ, kq = (idx % KQ) * 4;
      f32x4 v = m0 + row < g.M ? *reinterpret_cast<const f32x4*>(g.A + (size_t)(m0 + row) * g.lda + k0 + kq) : zero;
      if constexpr (ACT == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      ra[l] = v;
    }
#pragma unroll
    for (int l = 0; l < LB; ++l) {
      const int idx = tid + l * 256, row = idx / KQ, kq = (idx % KQ) * 4;
      rb[l] = n0 + row < g.N ? *reinterpret_cast<const f32x4*>(g.W + (size_t)(n0 + row) * g.ldw + k0 + kq) : zero;
    }
  };
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = zero;
  const int fr = lane & 15, fk = lane >> 4;
  if (my_steps > 0) load(kbeg);
  for (int st = 0; st < loop_steps; ++st) {
    const int k0 = kbeg + st * BK;
    const bool live = k0 < kend;                               // the second quartet may run one (empty) step more
#pragma unroll
    for (int l = 0; l < LA; ++l) {
      const int idx = tid + l * 256, row = idx / KQ, kq = (idx % KQ) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) sA[row * PITCH + kq + e] = ra[l][e];
    }
#pragma unroll
    for (int l = 0; l < LB; ++l) {
      const int idx = tid + l * 256, row = idx / KQ, kq = (idx % KQ) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) sB[row * PITCH + kq + e] = rb[l][e];
    }
    __syncthreads();
    if (k0 + BK < kend) load(k0 + BK);
    if (live) {
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      float fa[FM], fb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[i] = sA[(wm * FM * 16 + i * 16 + fr) * PITCH + ks * 4 + fk];
#pragma unroll
      for (int j = 0; j < FN; ++j) fb[j] = sB[(wn * FN * 16 + j * 16 + fr) * PITCH + ks * 4 + fk];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    }
    __syncthreads();
  }
  if constexpr (KG == 2) {
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          *reinterpret_cast<f32x4*>(sX + ((wave * FM + i) * FN + j) * 256 + lane * 4) = acc[i][j];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(sX + ((wave * FM + i) * FN + j) * 256 + lane * 4);
  }
  // D[row = 4 * (lane >> 4) + r][col = lane & 15]
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = n0 + wn * FN * 16 + j * 16 + fr;
      if (col >= g.N) continue;
      const float b = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * FM * 16 + i * 16 + fk * 4 + r;
        if (row >= g.M) continue;
        float y = acc[i][j][r] + b;
        float dg = 0.f;
        if constexpr (ACT == 1) {
          float gl;
          gelu_both(y, gl, dg);
          y = gl;
        }
        if (g.resid) y += g.resid[(size_t)row * g.ldr + col];
        if (g.out32) g.out32[(size_t)row * g.ldo + col] = y;
        if (g.out16) g.out16[(size_t)row * g.ld16 + col] = f2bf(y);
        if constexpr (ACT == 1) {
          if (g.out16b) g.out16b[(size_t)row * g.ld16b + col] = f2bf(dg);
        }
      }
    }
}

// Many rows (the text tower, M = B * L >= 1024): the same product on the bf16 matrix pipe at ~fp32 accuracy.
// Every fp32 operand element is split x = hi + lo, hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits together) and
//     A W^T ~= Ah Wh^T + Ah Wl^T + Al Wh^T        (fp32 accumulate; the dropped lo x lo term and the split error are 2^-16 relative
// per product, random in sign: measured 1e-5 of |y| max against an fp64 product, tests/test_kernels_gpu.py)
// - three v_mfma_f32_16x16x32_bf16 (16 cycles each, K = 32) where the exact path issues eight v_mfma_f32_16x16x4_f32 of 32
// cycles: 5x less matrix-pipe time.  Why it matters although the text tower is 0.7 % of the FLOPs: its linears run on a
// side stream BESIDE the video tower, and a 192-workgroup launch of the exact kernel holds 192 CUs for ~40-60 us during which
// the persistent GEMM workgroups of the main stream wait for their CU; measured by switching these launches off:
// 0.93 ms of a 46 ms step (frozen), 3.0 ms of 55 (global_local: two text passes).  128 x 128 tiles: 48 workgroups for
// N = 768, four waves of 64 x 64, K-step 32; global -> registers (next step in flight) -> split -> LDS -> ds_read_b128.
// KG = 2: two wave quartets split K (each with its own LDS planes; quartet 1 hands its partial tile over through the LDS at the
// end, a fixed order) and every quartet keeps the global loads of TWO K-steps in flight: the kernel is bound by the
// global -> register round trip of a K-step (~1.5 us beside the video tower's kernels), not by its 0.4 us of MFMAs.
template <int ACT, int KG>
__global__ __launch_bounds__(256 * KG) void linear_x3_kernel(LinArgs g) {
  constexpr int BM = 128, BN = 128, BK = 32, PITCH = BK * 2 + 16;        // bytes per LDS row: 64 + 16 (conflict-free 16-row reads)
  constexpr int PLANE = (BM + BN) * PITCH;                                 // one plane (hi or lo) of both operands
  __shared__ __attribute__((aligned(16))) char smem_all[KG * 2 * PLANE];
  const int kg = KG == 2 ? (int)(threadIdx.x >> 8) : 0;
  char* const sm = smem_all + kg * 2 * PLANE;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // PARTS: this workgroup's 128 columns lie in ONE part (part_n % 128 == 0): its weight rows and bias start at column nW0
  const int part = g.part_n > 0 ? n0 / g.part_n : 0, nW0 = part * g.part_n;        // workgroup-uniform
  const float* const Wp = part == 0 ? g.W : part == 1 ? g.W2 : g.W3;
  const float* const bp = part == 0 ? g.bias : part == 1 ? g.bias2 : g.bias3;
  const int nWend = g.part_n > 0 ? nW0 + g.part_n : g.N;
  const int nsteps = g.K / BK, my_steps = KG == 2 ? (kg == 0 ? (nsteps + 1) / 2 : nsteps / 2) : nsteps;
  const int kbeg = KG == 2 && kg == 1 ? ((nsteps + 1) / 2) * BK : 0;
  const int loop_steps = KG == 2 ? (nsteps + 1) / 2 : nsteps;            // both quartets run the same number of barrier pairs
  f32x4 ra[2][4], rb[2][4];
  auto load = [&](int st, auto B) {
    constexpr int bsel = decltype(B)::value;
    const bool live = st < my_steps;
    const int k0 = kbeg + st * BK;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int idx = tid + l * 256, row = idx >> 3, kq = (idx & 7) * 4;
      f32x4 v = (live && m0 + row < g.M) ? *reinterpret_cast<const f32x4*>(g.A + (size_t)(m0 + row) * g.lda + k0 + kq) : zero;
      if constexpr (ACT == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      ra[bsel][l] = v;
      rb[bsel][l] = (live && n0 + row < nWend) ? *reinterpret_cast<const f32x4*>(Wp + (size_t)(n0 - nW0 + row) * g.ldw + k0 + kq) : zero;
    }
  };
  auto split_store = [&](const f32x4 v, char* dst) {
    const bf16x4 hi = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    const bf16x4 lo = {f2bf(v[0] - bf2f(hi[0])), f2bf(v[1] - bf2f(hi[1])), f2bf(v[2] - bf2f(hi[2])), f2bf(v[3] - bf2f(hi[3]))};
    *reinterpret_cast<bf16x4*>(dst) = hi;
    *reinterpret_cast<bf16x4*>(dst + PLANE) = lo;
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = zero;
  const int fr = lane & 15, fk = lane >> 4;
  const char* const pa = sm + (wm * 64 + fr) * PITCH + fk * 16;
  const char* const pb = sm + (BM + wn * 64 + fr) * PITCH + fk * 16;
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  auto step = [&](int st, auto B) {
    constexpr int bsel = decltype(B)::value;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int idx = tid + l * 256, row = idx >> 3, kq = (idx & 7) * 4;
      split_store(ra[bsel][l], sm + row * PITCH + kq * 2);
      split_store(rb[bsel][l], sm + (BM + row) * PITCH + kq * 2);
    }
    __syncthreads();
    load(st + 2, B);                                                     // this register set is free again: two steps ahead
    if (st < my_steps) {
      bf16x8 ah[4], al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(pa + i * 16 * PITCH);
        al[i] = *reinterpret_cast<const bf16x8*>(pa + i * 16 * PITCH + PLANE);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(pb + j * 16 * PITCH);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(pb + j * 16 * PITCH + PLANE);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // small terms first, then the leading one.  The WEIGHT fragment is the first operand: the tile comes out transposed - lane
          // (fk, fr) holds columns 4 fk .. 4 fk + 3 of row fr - so that the epilogue moves 16 bytes (fp32) / 8 bytes (bf16) per lane
          // and row instead of one element (64 scalar stores per lane and output tensor before, 16 vector stores now)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[i], acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  };
  load(0, B0{});
  load(1, B1{});
  for (int st = 0; st < loop_steps; st += 2) {
    step(st, B0{});
    if (st + 1 < loop_steps) step(st + 1, B1{});
  }
  if constexpr (KG == 2) {                                                // quartet 1 hands over through the (now dead) operand planes
    float* const sx = reinterpret_cast<float*>(smem_all);
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(sx + ((wave * 4 + i) * 4 + j) * 256 + lane * 4) = acc[i][j];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(sx + ((wave * 4 + i) * 4 + j) * 256 + lane * 4);
  }
  // D^T: lane (fk, fr) holds output row m = .. + fr, columns n = .. + 4 fk + r.  A workgroup's tile is interior (whole vector accesses,
  // every pointer 16-byte aligned) unless it touches the ragged edge of M or N or a leading dimension is not a multiple of 4.
  // (views with an odd element offset - out32[:, 1:], a bias slice of a flat buffer - take the scalar path: the base pointers' low
  // address bits are part of the predicate, 16 bytes for the fp32 tensors, 8 for the bf16 ones)
  const bool vec = m0 + BM <= g.M && n0 + BN <= g.N && (g.ldo & 3) == 0 && (g.ld16 & 3) == 0 && (g.ld16b & 3) == 0 && (g.ldr & 3) == 0 &&
                   ((nW0 | n0) & 3) == 0 &&
                   (((uintptr_t)g.out32 | (uintptr_t)g.resid | (uintptr_t)bp) & 15) == 0 && (((uintptr_t)g.out16 | (uintptr_t)g.out16b) & 7) == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + wm * 64 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn * 64 + j * 16 + fk * 4;
      if (vec) {
        f32x4 y = acc[i][j];
        if (bp) y += *reinterpret_cast<const f32x4*>(bp + col - nW0);
        f32x4 dg = zero;
        if constexpr (ACT == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { float gl, d; gelu_both(y[r], gl, d); y[r] = gl; dg[r] = d; }
        }
        if (g.resid) y += *reinterpret_cast<const f32x4*>(g.resid + (size_t)row * g.ldr + col);
        if (g.out32) *reinterpret_cast<f32x4*>(g.out32 + (size_t)row * g.ldo + col) = y;
        if (g.out16) *reinterpret_cast<bf16x4*>(g.out16 + (size_t)row * g.ld16 + col) = bf16x4{f2bf(y[0]), f2bf(y[1]), f2bf(y[2]), f2bf(y[3])};
        if constexpr (ACT == 1) {
          if (g.out16b) *reinterpret_cast<bf16x4*>(g.out16b + (size_t)row * g.ld16b + col) = bf16x4{f2bf(dg[0]), f2bf(dg[1]), f2bf(dg[2]), f2bf(dg[3])};
        }
        continue;
      }
      if (row >= g.M) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = col + r;
        if (c >= g.N) continue;
        float y = acc[i][j][r] + (bp ? bp[c - nW0] : 0.f);
        float dg = 0.f;
        if constexpr (ACT == 1) {
          float gl;
          gelu_both(y, gl, dg);
          y = gl;
        }
        if (g.resid) y += g.resid[(size_t)row * g.ldr + c];
        if (g.out32) g.out32[(size_t)row * g.ldo + c] = y;
        if (g.out16) g.out16[(size_t)row * g.ld16 + c] = f2bf(y);
        if constexpr (ACT == 1) {
          if (g.out16b) g.out16b[(size_t)row * g.ld16b + c] = f2bf(dg);
        }
      }
    }
  }
}

// Few rows (M <= 64: the CLS lane, the projections): latency is everything, because these launches sit on a side stream
// beside CU-filling GEMMs and only run in the gaps.  One workgroup per 32 rows x 16 output columns; its four waves split
// K four ways (each wave streams its quarter of the two operand slabs straight from global memory into MFMA operands:
// a float4 per lane covers four consecutive k of its row - the k order inside a 16-chunk is permuted identically for
// both operands, which a dot product does not see) and their partial tiles are summed through LDS.  N / 16 workgroups,
// 12 dependent load batches for K = 3072 instead of 48 K-steps.
template <int ACT>
__global__ __launch_bounds__(256) void linear_f32_small_kernel(LinArgs g) {
  constexpr int CB = 6;                                   // 16-k chunks per load batch
  __shared__ f32x4 sred[3][2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 16;
  const int kq = g.K >> 2, kbase = wave * kq;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const bool ok0 = m0 + r < g.M, ok1 = m0 + 16 + r < g.M, okb = n0 + r < g.N;
  const float* a0 = g.A + (size_t)(ok0 ? m0 + r : 0) * g.lda + kbase + 4 * q;
  const float* a1 = g.A + (size_t)(ok1 ? m0 + 16 + r : 0) * g.lda + kbase + 4 * q;
  const float* bw = g.W + (size_t)(okb ? n0 + r : 0) * g.ldw + kbase + 4 * q;
  f32x4 acc[2] = {zero, zero};
  for (int c0 = 0; c0 < kq; c0 += CB * 16) {
    f32x4 va[2][CB], vb[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const int k = c0 + c * 16;
      const bool in = k < kq;
      va[0][c] = (in && ok0) ? *reinterpret_cast<const f32x4*>(a0 + k) : zero;
      va[1][c] = (in && ok1) ? *reinterpret_cast<const f32x4*>(a1 + k) : zero;
      vb[c] = (in && okb) ? *reinterpret_cast<const f32x4*>(bw + k) : zero;
      if constexpr (ACT == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { va[0][c][e] = fmaxf(va[0][c][e], 0.f); va[1][c][e] = fmaxf(va[1][c][e], 0.f); }
      }
    }
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[0][c][t], vb[c][t], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[1][c][t], vb[c][t], acc[1], 0, 0, 0);
      }
  }
  if (wave > 0) { sred[wave - 1][0][lane] = acc[0]; sred[wave - 1][1][lane] = acc[1]; }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const f32x4 v = acc[f] + sred[0][f][lane] + sred[1][f][lane] + sred[2][f][lane];     // fixed order: deterministic
    const int col = n0 + r;
    if (col >= g.N) continue;
    const float b = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = m0 + f * 16 + q * 4 + rr;
      if (row >= g.M) continue;
      float y = v[rr] + b;
      float dg = 0.f;
      if constexpr (ACT == 1) {
        float gl;
        gelu_both(y, gl, dg);
        y = gl;
      }
      if (g.resid) y += g.resid[(size_t)row * g.ldr + col];
      if (g.out32) g.out32[(size_t)row * g.ldo + col] = y;
      if (g.out16) g.out16[(size_t)row * g.ld16 + col] = f2bf(y);
      if constexpr (ACT == 1) {
        if (g.out16b) g.out16b[(size_t)row * g.ld16b + col] = f2bf(dg);
      }
    }
  }
}

// Fixed since round 6 (the three A/B environment knobs of this dispatch are gone): M > 64 runs on the split-bf16 kernel
// (linear_x3_kernel) unless the call asks for exact products; K-tiles of 32 (half the barrier pairs and global round trips of 16:
// text tower forward 2124 -> 1950 us alone, same accumulation order, bit-identical); too few 128 x 128 tiles -> 64 x 64 workgroups of
// two wave quartets that split K.
constexpr int g_lin_kg2 = 1, g_lin_x3 = 1, g_lin_bk = 32;
template <int ACT>
static void launch_linear(const LinArgs& g, bool exact, hipStream_t s) {
  const bool bk32 = g_lin_bk == 32 && g.K % 32 == 0;
  if (g.M <= 64 && g.K % 64 == 0) {
    OAT_LAUNCH(linear_f32_small_kernel<ACT>, dim3((g.N + 15) / 16, (g.M + 31) / 32), dim3(256), 0, s, g);
  } else if (g_lin_x3 && !exact && g.M > 64 && g.K % 32 == 0) {
    // K split between two wave quartets unless the launch already has plenty of workgroups or K is short
    if (g.K >= 256 && ((g.N + 127) / 128) * ((g.M + 127) / 128) <= 256)
      OAT_LAUNCH((linear_x3_kernel<ACT, 2>), dim3((g.N + 127) / 128, (g.M + 127) / 128), dim3(512), 0, s, g);
    else
      OAT_LAUNCH((linear_x3_kernel<ACT, 1>), dim3((g.N + 127) / 128, (g.M + 127) / 128), dim3(256), 0, s, g);
  } else if (((g.N + 127) / 128) * ((g.M + 127) / 128) >= 128) {
    if (bk32) OAT_LAUNCH((linear_f32_kernel<ACT, 128, 128, 32, 2, 2>), dim3((g.N + 127) / 128, (g.M + 127) / 128), dim3(256), 0, s, g);
    else OAT_LAUNCH((linear_f32_kernel<ACT, 128, 128, 16, 2, 2>), dim3((g.N + 127) / 128, (g.M + 127) / 128), dim3(256), 0, s, g);
  } else if (g_lin_kg2 && g.K >= 256) {   // too few 128 x 128 tiles to occupy the GPU (text tower, N = 768: 48): quarter tiles,
    // two wave quartets per workgroup splitting K (two waves per SIMD)
    if (bk32) OAT_LAUNCH((linear_f32_kernel<ACT, 64, 64, 32, 2, 2, 2>), dim3((g.N + 63) / 64, (g.M + 63) / 64), dim3(512), 0, s, g);
    else OAT_LAUNCH((linear_f32_kernel<ACT, 64, 64, 16, 2, 2, 2>), dim3((g.N + 63) / 64, (g.M + 63) / 64), dim3(512), 0, s, g);
  } else {
    OAT_LAUNCH((linear_f32_kernel<ACT, 64, 64, 16, 2, 2>), dim3((g.N + 63) / 64, (g.M + 63) / 64), dim3(256), 0, s, g);
  }
}

}  // namespace oat

extern "C" int oat_linear_f32(const float* A, int lda, const float* W, int ldw, const float* bias, int M, int N, int K,
                              float* out32, int ldo, void* out16, int ld16, void* out16b, int ld16b,
                              const float* resid, int ldr, int act, void* stream) {
  using namespace oat;
  if (M <= 0 || N <= 0 || K <= 0) { set_error("linear_f32: empty problem"); return -1; }
  if (K % 16 != 0 || lda % 4 != 0 || ldw % 4 != 0) { set_error("linear_f32: K % 16, lda % 4, ldw % 4 must be 0"); return -2; }
  if (!A || !W || (!out32 && !out16)) { set_error("linear_f32: null pointer"); return -4; }
  // act = activation (0 none, 1 GELU on the output, 2 ReLU on the input: OAT_LIN_GELU / OAT_LIN_RELU_IN of include/oatrans_hip.h) | OAT_LIN_EXACT (0x100): keep the exact-f32 MFMA at every M.  Without the flag
  // M > 64 rows run on the split-bf16 kernel (2^-16 relative per product: the text tower's forward); the video tower's fp32 CLS lane and
  // the projection heads - the rows the 1e-3 sim-matrix bound hangs on - pass the flag so that a batch of more than 64 clips keeps f32 products.
  const bool exact = (act & 0x100) != 0;
  act &= 0xff;
  if (act < 0 || act > 2) { set_error("linear_f32: unknown activation"); return -5; }
  LinArgs g{A, lda, W, ldw, bias, M, N, K, out32, ldo, (bf16*)out16, ld16, (bf16*)out16b, ld16b, resid, ldr, nullptr, nullptr, nullptr, nullptr, 0};
  hipStream_t s = (hipStream_t)stream;
  if (act == 1) launch_linear<1>(g, exact, s);
  else if (act == 2) launch_linear<2>(g, exact, s);
  else launch_linear<0>(g, exact, s);
  return check_launch("linear_f32");
}

// ---- backward of a FEW-row linear layer (the projection heads: txt_proj = ReLU -> Linear(768, 256), vid_proj = Linear(768, 256) on B rows,
// /root/reference/OATrans/model/oa_model.py:66-78): dx, dW and db of y = act(x) W^T + b in ONE launch of plain fp32 arithmetic.  The
// bf16 GEMM path took a zero-fill, two casts, a weight-gradient GEMM + reduce, a column sum + reduce, a data-gradient GEMM and a ReLU
// mask - eleven launches of ~5 us around 6 MFLOP, in the serial stretch between the towers' forward and backward.
// Workgroup = 16 columns of K: dy [M, N] (pitch N + 1), act(x)[:, 16] and W[:, 16] in LDS; thread n owns dW[n, 16], four threads own a row of dx.
namespace oat {
struct LinBwdArgs { const float* x; int ldx; const float* dy; int lddy; const float* W; int ldw; int M, N, K, relu_in;
                    float* dx; int lddx; float* dW; float* db; };
__global__ __launch_bounds__(256) void linear_small_bwd_kernel(LinBwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  const int P = g.N + 1;
  float* const dys = lsm;                       // [M][N + 1]
  float* const xs = dys + g.M * P;              // [M][16]   act(x)
  float* const ms = xs + g.M * 16;              // [M][16]   1 where the input passes the ReLU (all ones without it)
  float* const ws = ms + g.M * 16;              // [N][16]
  const int k0 = blockIdx.x * 16, tid = threadIdx.x;
  for (int i = tid; i < g.M * g.N; i += 256) { const int m = i / g.N, n = i - m * g.N; dys[m * P + n] = g.dy[(size_t)m * g.lddy + n]; }
  for (int i = tid; i < g.M * 16; i += 256) {
    const int m = i >> 4, kk = i & 15;
    const float v = g.x[(size_t)m * g.ldx + k0 + kk];
    const bool pass = !g.relu_in || v > 0.f;
    xs[i] = pass ? v : 0.f;
    ms[i] = pass ? 1.f : 0.f;
  }
  for (int i = tid; i < g.N * 16; i += 256) ws[i] = g.W[(size_t)(i >> 4) * g.ldw + k0 + (i & 15)];
  __syncthreads();
  // dW[n, k0 .. k0 + 16) = sum_m dy[m, n] act(x)[m, k]        (rows in order: deterministic)
  for (int n = tid; n < g.N; n += 256) {
    float a[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) a[kk] = 0.f;
    for (int m = 0; m < g.M; ++m) {
      const float d = dys[m * P + n];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) a[kk] = __builtin_fmaf(d, xs[m * 16 + kk], a[kk]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(g.dW + (size_t)n * g.K + k0 + q * 4) = f32x4{a[q * 4], a[q * 4 + 1], a[q * 4 + 2], a[q * 4 + 3]};
    if (blockIdx.x == 0 && g.db) {
      float sb = 0.f;
      for (int m = 0; m < g.M; ++m) sb += dys[m * P + n];
      g.db[n] = sb;
    }
  }
  // dx[m, k0 .. k0 + 16) = mask * sum_n dy[m, n] W[n, k]
  if (g.dx) {
    for (int i = tid; i < g.M * 4; i += 256) {
      const int m = i >> 2, kq = (i & 3) * 4;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      for (int n = 0; n < g.N; ++n) a += dys[m * P + n] * *reinterpret_cast<const f32x4*>(ws + n * 16 + kq);
      const f32x4 mk = *reinterpret_cast<const f32x4*>(ms + m * 16 + kq);
      *reinterpret_cast<f32x4*>(g.dx + (size_t)m * g.lddx + k0 + kq) = a * mk;
    }
  }
}
}  // namespace oat

// dx [M, K] (optional), dW [N, K] (dense), db [N] (optional) of y = act(x) W^T + b given dy [M, N]; relu_in: act = ReLU (txt_proj).
// M <= 64, K % 16 == 0, (M (N + 1) + 32 M + 16 N) floats of LDS <= 96 KB.  Results are written, not accumulated.
extern "C" int oat_linear_small_bwd(const float* x, int ldx, const float* dy, int lddy, const float* W, int ldw, int M, int N, int K,
                                    int relu_in, float* dx, int lddx, float* dW, float* db, void* stream) {
  using namespace oat;
  if (M <= 0 || M > 64 || N <= 0 || K <= 0 || K % 16 != 0) { set_error("linear_small_bwd: 1 <= M <= 64, K % 16 == 0"); return -1; }
  if (!x || !dy || !W || !dW) { set_error("linear_small_bwd: null pointer"); return -4; }
  const size_t lds = ((size_t)M * (N + 1) + 32 * (size_t)M + 16 * (size_t)N) * sizeof(float);
  if (lds > 96 * 1024) { set_error("linear_small_bwd: M x N too large for the LDS"); return -3; }
  if (lddx % 4 != 0 && dx) { set_error("linear_small_bwd: lddx % 4 must be 0"); return -2; }
  OAT_MAX_LDS(linear_small_bwd_kernel, 96 * 1024);
  LinBwdArgs g{x, ldx, dy, lddy, W, ldw, M, N, K, relu_in, dx, lddx, dW, db};
  OAT_LAUNCH(linear_small_bwd_kernel, dim3(K / 16), dim3(256), (unsigned)lds, (hipStream_t)stream, g);
  return check_launch("linear_small_bwd");
}

// out[M, 3 n] = A[M, K] . [Wq; Wk; Wv]^T + [bq | bk | bv]: the three n x K linear layers of an attention block (HF DistilBERT keeps q_lin /
// k_lin / v_lin as separate parameters, /root/reference/OATrans/model/oa_model.py:27 -> transformers 4.6 MultiHeadSelfAttention) in ONE launch
// of the split-bf16 kernel - same arithmetic per output element as three oat_linear_f32 calls, three times the workgroups per launch
// (the text tower's launches are latency-bound: 48 workgroups each).  n % 128 == 0, K % 32 == 0, M > 64.
extern "C" int oat_linear_f32_qkv(const float* A, int lda, const float* Wq, const float* Wk, const float* Wv, int ldw,
                                  const float* bq, const float* bk, const float* bv, int M, int n, int K,
                                  float* out32, int ldo, void* out16, int ld16, void* stream) {
  using namespace oat;
  if (M <= 64 || n <= 0 || K <= 0) { set_error("linear_f32_qkv: M > 64, n > 0, K > 0"); return -1; }
  if (n % 128 != 0 || K % 32 != 0 || lda % 4 != 0 || ldw % 4 != 0) { set_error("linear_f32_qkv: n % 128, K % 32, lda % 4, ldw % 4 must be 0"); return -2; }
  if (!A || !Wq || !Wk || !Wv || (!out32 && !out16)) { set_error("linear_f32_qkv: null pointer"); return -4; }
  if ((bq == nullptr) != (bk == nullptr) || (bq == nullptr) != (bv == nullptr)) { set_error("linear_f32_qkv: all three biases or none"); return -4; }
  LinArgs g{A, lda, Wq, ldw, bq, M, 3 * n, K, out32, ldo, (bf16*)out16, ld16, nullptr, 0, nullptr, 0, Wk, Wv, bk, bv, n};
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((3 * n) / 128, (M + 127) / 128);
  if (K >= 256 && grid.x * grid.y <= 256) OAT_LAUNCH((linear_x3_kernel<0, 2>), grid, dim3(512), 0, s, g);
  else OAT_LAUNCH((linear_x3_kernel<0, 1>), grid, dim3(256), 0, s, g);
  return check_launch("linear_f32_qkv");
}
