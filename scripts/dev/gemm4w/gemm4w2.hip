// Dev probe 2 (not part of the library): 4-wave NT bf16 GEMM (one wave per SIMD, 128-column x TM/2-row wave tiles, accumulators in
// AGPRs) whose bf16 epilogue is INTERLEAVED with the matrix work of the neighbouring K-tiles: row group i of a finished tile is
// converted and stored while the last K-tile's second half still multiplies the row groups behind it, the last row group while the
// NEXT tile's first half-K-tile multiplies (C = 0 form) - the 8-wave ping-pong kernel (csrc/gemm_nt_pp.hip) has no registers for
// that (2 waves x 256) and pays ~3.5 us per 256 x 256 tile, 20 % of a K = 768 launch.
// hipcc --offload-arch=gfx950 -O3 -I../../../oa-transformer_amd/csrc
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
using namespace oat;
namespace oat { void set_error(const char*) {} int check_launch(const char*) { return 0; } bool tape_recording() { return false; } void tape_push(std::function<void()>&&) {} }

constexpr int STAGE = 65536, BOFF = 32768, BIAS_OFF = 2 * STAGE, MAXN = 4096, LDS_BYTES = BIAS_OFF + MAXN * 4;

struct Args { const bf16* A; const bf16* B; bf16* C; const float* bias; int M, N, K, lda, ldb, ldc; };

// TM: tile rows (256 or 224); NI = row groups of 16 per wave row
template <int TM>
__global__ __launch_bounds__(256) void gemm4w2_kernel(Args g) {
  constexpr int WR = TM / 2, NI = WR / 16, NG = NI * 2;          // NG groups of 4 MFMAs per half K-tile (NI x 8 MFMAs)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = g.N >> 8, ntm = (g.M + TM - 1) / TM, ntiles = ntm * ntn;
  const int nk = g.K >> 6;
  const int grid = (int)gridDim.x;
  const int ntl = (ntiles - 1 - (int)blockIdx.x) / grid + 1;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  float* const sbias = reinterpret_cast<float*>(smem + BIAS_OFF);
  for (int i = tid; i < g.N; i += 256) sbias[i] = g.bias ? g.bias[i] : 0.f;
  struct Tile { int m0, n0; };
  auto tile_of = [&](int t) __attribute__((always_inline)) {       // XCD-contiguous, bijective (as gemm_nt_pp)
    const int w = (int)blockIdx.x + t * grid;
    const int q = ntiles >> 3, r = ntiles & 7, xcd = w & 7, idx = w >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = bid / ntn;
    return Tile{tm * TM, (bid - tm * ntn) << 8};
  };
  // ---- staging.  piece = 8 LDS rows x 128 B; wave w stages A pieces w*8..w*8+7 and B pieces w*8..w*8+7 of every K-tile.
  const int srow = lane >> 3;
  const uint32_t lda2 = (uint32_t)g.lda * 2, ldb2 = (uint32_t)g.ldb * 2;
  const uint32_t c16_0 = (uint32_t)(((lane & 7) ^ (srow >> 1)) << 4);
  // A: LDS row lr = p*8 + srow -> tile row (lr >> 7) * WR + min(lr & 127, WR - 1); pieces whose rows lie past WR - 1 re-read row WR - 1
  const uint32_t voffA = (uint32_t)srow * lda2 + c16_0, voffAc = c16_0;
  // B: LDS row r = p*8 + srow <- B row (r & ~63) | ((r & 15) << 2) | ((r >> 4) & 3)  (direct epilogue: lane owns 4 consecutive columns)
  const uint32_t voffB = (uint32_t)(srow * 4) * ldb2 + c16_0;
  const char* const A0 = reinterpret_cast<const char*>(g.A);
  const char* const B0 = reinterpret_cast<const char*>(g.B);
  // staging cursor (K-tile s + 2 of the stream)
  int ckt = 0, ctl = 0;
  const char *ca, *cb;
  uint32_t live = 1;
  { const Tile t = tile_of(0); ca = A0 + (size_t)t.m0 * lda2; cb = B0 + (size_t)t.n0 * ldb2; }
  auto advance = [&]() __attribute__((always_inline)) {
    ++ckt; ca += 128; cb += 128;
    if (ckt == nk) {
      const bool more = ctl + 1 < ntl;
      ctl += more ? 1 : 0;
      const Tile t = tile_of(ctl);
      ckt = more ? 0 : nk - 1;
      ca = more ? A0 + (size_t)t.m0 * lda2 : ca - 128;
      cb = more ? B0 + (size_t)t.n0 * ldb2 : cb - 128;
      live = more ? live : 0u;
    }
  };
  auto dma_piece = [&](int e, int buf) __attribute__((always_inline)) {     // e = 0..7: A piece, 8..15: B piece
    const int p = wave * 8 + (e & 7);
    if (e < 8) {
      const int within = (p & 15) * 8;                                      // first row of the piece inside its wave row (LDS rows)
      const bool past = TM != 256 && within >= WR;                          // wave-uniform: the whole piece lies past the wave row's last real row
      const size_t rowbase = (size_t)((p >> 4) * WR + (past ? WR - 1 : within)) * lda2;
      glds16_asm_lds(ca + rowbase, (past ? voffAc : voffA) ^ (uint32_t)((e & 1) << 6), lds0 + buf * STAGE + p * 1024);
    } else {
      const size_t rowbase = (size_t)((p >> 3) * 64 + (p & 1) * 32 + ((p >> 1) & 3)) * ldb2;
      glds16_asm_lds(cb + rowbase, voffB ^ (uint32_t)((e & 1) << 6), lds0 + buf * STAGE + BOFF + p * 1024);
    }
  };
  // ---- fragment addresses
  const int frow = lane & 15, fk = lane >> 4, sw = (frow >> 1) & 7;
  uint32_t pA[2], pB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ch = ((kk * 4 + fk) ^ sw) << 4;
    pA[kk] = lds0 + (wm * 128 + frow) * 128 + ch;
    pB[kk] = lds0 + BOFF + (wn * 128 + frow) * 128 + ch;
  }
  typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
  f32x4 acc[NI][8];
  bf16x8 a0[NI], b0[8], a1[NI], b1[8];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // ---- epilogue pieces.  Row group i of the wave tile: lane (fk, frow) owns rows 16 i + 4 fk + r and, per 64-column block, the 4
  // consecutive columns 4 frow + t (tile j = 4 blk + t): 8 bytes per lane, 128 B per 16 lanes.  8 stores per row group.
  int em0 = 0, en0 = 0;                 // the tile the pending accumulators belong to
  auto store_row = [&](int i) __attribute__((always_inline)) {
    char* const ob = reinterpret_cast<char*>(g.C) + ((size_t)(em0 + wm * WR + i * 16) * g.ldc + en0 + wn * 128) * 2;
    const uint32_t lo = (uint32_t)(fk * 4 * g.ldc + frow * 4) * 2;
    const bool interior = em0 + TM <= g.M;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + en0 + wn * 128 + blk * 64 + frow * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bf16x4 o = {f2bf(acc[i][4 * blk + 0][r] + bv[0]), f2bf(acc[i][4 * blk + 1][r] + bv[1]),
                          f2bf(acc[i][4 * blk + 2][r] + bv[2]), f2bf(acc[i][4 * blk + 3][r] + bv[3])};
        if (interior || em0 + wm * WR + i * 16 + fk * 4 + r < g.M)
          *reinterpret_cast<bf16x4*>(ob + (size_t)((uint32_t)r * (uint32_t)g.ldc * 2) + lo + blk * 128) = o;
      }
    }
  };

  // ---- prologue: K-tiles 0 and 1 of the stream
#pragma unroll
  for (int e = 0; e < 16; ++e) dma_piece(e, 0);
  advance();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) dma_piece(e, 1);
  advance();
#pragma unroll
  for (int i = 0; i < NI; ++i) a0[i] = *(lds_frag)(uintptr_t)(pA[0] + i * 2048);
#pragma unroll
  for (int j = 0; j < 8; ++j) b0[j] = *(lds_frag)(uintptr_t)(pB[0] + j * 2048);
  __builtin_amdgcn_sched_barrier(0);

  bool pending = false;
  int s = 0;
  for (int tl = 0; tl < ntl; ++tl) {
    const Tile tile = tile_of(tl);
    for (int kt = 0; kt < nk; ++kt, ++s) {
      const uint32_t cur = (s & 1) * STAGE, nxt = ((s + 1) & 1) * STAGE;
      const bool first = kt == 0, last = kt == nk - 1;
      // ---- phase 1: kk = 0 from (a0, b0); the kk = 1 fragments of this K-tile are read underneath
      if (first) {
        // C = 0 form; the last row group of the previous tile leaves first (its accumulators are overwritten last)
        if (pending) store_row(NI - 1);
#pragma unroll
        for (int gI = 0; gI < NG; ++gI) {
          if (gI < NI) a1[gI] = *(lds_frag)(uintptr_t)(pA[1] + cur + gI * 2048);
          else if (gI - NI < 8) b1[gI - NI] = *(lds_frag)(uintptr_t)(pB[1] + cur + (gI - NI) * 2048);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int t = gI * 4 + q, i = t >> 3, j = t & 7;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b0[j], zero, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (NG < NI + 8) {
#pragma unroll
          for (int x = NG; x < NI + 8; ++x) b1[x - NI] = *(lds_frag)(uintptr_t)(pB[1] + cur + (x - NI) * 2048);
        }
      } else {
#pragma unroll
        for (int gI = 0; gI < NG; ++gI) {
          if (gI < NI) a1[gI] = *(lds_frag)(uintptr_t)(pA[1] + cur + gI * 2048);
          else if (gI - NI < 8) b1[gI - NI] = *(lds_frag)(uintptr_t)(pB[1] + cur + (gI - NI) * 2048);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int t = gI * 4 + q, i = t >> 3, j = t & 7;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (NG < NI + 8) {
#pragma unroll
          for (int x = NG; x < NI + 8; ++x) b1[x - NI] = *(lds_frag)(uintptr_t)(pB[1] + cur + (x - NI) * 2048);
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // K-tile s + 1 landed (this wave's pieces); every read of `cur` done
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase 2: kk = 1 from (a1, b1); K-tile s + 2 is requested into `cur`, fragments (s + 1, kk = 0) are read
      if (last) { em0 = tile.m0; en0 = tile.n0; }
#pragma unroll
      for (int gI = 0; gI < NG; ++gI) {
        dma_piece(gI, s & 1);
        if (gI == 0) dma_piece(NG, s & 1);                    // 16 pieces over NG (14 or 16) groups
        if (gI == 1 && NG + 1 < 16) dma_piece(NG + 1, s & 1);
        if (gI < NI) a0[gI] = *(lds_frag)(uintptr_t)(pA[0] + nxt + gI * 2048);
        else if (gI - NI < 8) b0[gI - NI] = *(lds_frag)(uintptr_t)(pB[0] + nxt + (gI - NI) * 2048);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = gI * 4 + q, i = t >> 3, j = t & 7;
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
        }
        // the tile's last MFMAs: row group i is final after groups 2i, 2i + 1 - it leaves two groups later, under the next rows' MFMAs
        if (last && (gI & 1) && gI >= 3) store_row((gI - 3) >> 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (NG < NI + 8) {
#pragma unroll
        for (int x = NG; x < NI + 8; ++x) b0[x - NI] = *(lds_frag)(uintptr_t)(pB[0] + nxt + (x - NI) * 2048);
      }
      advance();
      if (last) {
        // rows (NG - 3) / 2 .. NI - 2 could not leave inside the loop; row NI - 1 waits for the next tile's first phase
#pragma unroll
        for (int i = ((NG - 1 - 3) >> 1) + 1; i < NI - 1; ++i) store_row(i);
        pending = true;
      }
    }
  }
  if (pending) store_row(NI - 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------- host
static unsigned short f2b(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float b2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void ref_kernel(const bf16* A, const bf16* B, const float* bias, float* C, int M, int N, int K) {
  const int n = blockIdx.x * 16 + threadIdx.x, m = blockIdx.y * 16 + threadIdx.y;
  if (m >= M || n >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += (float)A[(size_t)m * K + k] * (float)B[(size_t)n * K + k];
  C[(size_t)m * N + n] = s + bias[n];
}

template <int TM>
static void launch(const Args& g, int grid_slots) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w2_kernel<TM>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); attr = true; }
  const int nt = ((g.M + TM - 1) / TM) * (g.N / 256);
  hipLaunchKernelGGL(gemm4w2_kernel<TM>, dim3(nt < grid_slots ? nt : grid_slots), dim3(256), LDS_BYTES, 0, g);
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : 0;
  // ---- validation: several tiles per workgroup, ragged M, both tile heights
  for (int tmsel = 0; tmsel < 2 && !only; ++tmsel) {
    const int M = 1800, N = 768, K = 512, Mp = 2048;
    std::vector<unsigned short> hA((size_t)Mp * K), hB((size_t)N * K);
    std::vector<float> hbias(N);
    srand(3 + tmsel);
    for (auto& v : hA) v = f2b((rand() % 2001 - 1000) / 1000.f);
    for (auto& v : hB) v = f2b((rand() % 2001 - 1000) / 1000.f);
    for (auto& v : hbias) v = (rand() % 2001 - 1000) / 500.f;
    bf16 *dA, *dB, *dC; float *dbias, *dR;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)Mp * N * 2); hipMalloc(&dbias, N * 4); hipMalloc(&dR, (size_t)M * N * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice);
    hipMemset(dC, 0x7f, (size_t)Mp * N * 2);
    Args g{dA, dB, dC, dbias, M, N, K, K, K, N};
    if (tmsel == 0) launch<256>(g, 5); else launch<224>(g, 5);
    hipLaunchKernelGGL(ref_kernel, dim3(N / 16, (M + 15) / 16), dim3(16, 16), 0, 0, dA, dB, dbias, dR, M, N, K);
    hipDeviceSynchronize();
    std::vector<unsigned short> hC((size_t)Mp * N); std::vector<float> hR((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hR.data(), dR, hR.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0; size_t bad = 0, touched = 0;
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
      const double e = fabs(hR[(size_t)m * N + n] - b2f(hC[(size_t)m * N + n])) / fmax(1.0, fabs(hR[(size_t)m * N + n]));
      if (!(e < 1e-2)) ++bad;
      if (e == e) maxerr = fmax(maxerr, e);
    }
    for (int m = M; m < Mp; ++m) for (int n = 0; n < N; ++n) if (hC[(size_t)m * N + n] != 0x7f7f) ++touched;
    printf("validation TM %d: max rel err %.3e, %zu bad elements, %zu elements touched beyond M (%s)\n", tmsel ? 224 : 256, maxerr, bad, touched,
           bad == 0 && touched == 0 ? "ok" : "WRONG");
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dbias); hipFree(dR);
  }
  // ---- timing
  struct Shape { int M, N, K; } shapes[] = {{8192, 8192, 8192}, {50208, 768, 3072}, {50208, 768, 2304}, {50208, 2304, 768}, {50208, 768, 768}, {50208, 3072, 768}};
  for (auto sh : shapes) {
    const int Mp = (sh.M + 255) / 256 * 256 + 256;
    const int NS = 3;
    bf16 *dA[NS], *dB, *dC[NS]; float* dbias;
    std::vector<unsigned short> h((size_t)Mp * sh.K);
    for (auto& v : h) v = (unsigned short)(((rand() & 1) << 15) | ((120 + rand() % 8) << 7) | (rand() & 127));
    for (int s = 0; s < NS; ++s) { hipMalloc(&dA[s], (size_t)Mp * sh.K * 2); hipMemcpy(dA[s], h.data(), (size_t)Mp * sh.K * 2, hipMemcpyHostToDevice); hipMalloc(&dC[s], (size_t)Mp * sh.N * 2); }
    hipMalloc(&dB, (size_t)sh.N * sh.K * 2); hipMemcpy(dB, h.data(), (size_t)sh.N * sh.K * 2, hipMemcpyHostToDevice);
    hipMalloc(&dbias, sh.N * 4); hipMemset(dbias, 0, sh.N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int tmsel = 0; tmsel < 2; ++tmsel) {
      auto run = [&](int s) { Args g{dA[s], dB, dC[s], dbias, sh.M, sh.N, sh.K, sh.K, sh.K, sh.N}; if (tmsel == 0) launch<256>(g, 256); else launch<224>(g, 256); };
      for (int i = 0; i < 4; ++i) run(i % NS);
      hipDeviceSynchronize();
      const int reps = 12;
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); for (int i = 0; i < reps; ++i) run(i % NS); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms);
      }
      printf("M %6d N %5d K %5d TM %d: %8.1f us  %7.0f TFLOP/s\n", sh.M, sh.N, sh.K, tmsel ? 224 : 256, best / reps * 1e3, 2.0 * sh.M * sh.N * sh.K / (best / reps) / 1e9);
    }
    for (int s = 0; s < NS; ++s) { hipFree(dA[s]); hipFree(dC[s]); } hipFree(dB); hipFree(dbias);
  }
  return 0;
}
