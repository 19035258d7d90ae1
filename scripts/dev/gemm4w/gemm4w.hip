// Dev probe (not part of the library): a 4-wave "NT" bf16 GEMM, C[M,N] = A[M,K] B[N,K]^T, 256 x 256 x 64 tiles, ONE wave per SIMD
// with a 128 x 128 wave tile - 4 ds_read_b128 per 16 MFMAs instead of the 6 of the 8-wave ping-pong kernel (csrc/gemm_nt_pp.hip),
// which scripts/dev/mfma_power says is worth 1.55 -> 1.7 PF under the power cap.  Question: does a single instruction stream per SIMD
// (no partner wave to hide LDS / DMA latency behind) sustain it?   hipcc --offload-arch=gfx950 -O3 -I../../../oa-transformer_amd/csrc
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
using namespace oat;

namespace oat { void set_error(const char*) {} int check_launch(const char*) { return 0; } bool tape_recording() { return false; } void tape_push(std::function<void()>&&) {} }

constexpr int STAGE = 65536, BOFF = 32768;

struct Args { const bf16* A; const bf16* B; bf16* C; int M, N, K, lda, ldb, ldc; int noepi; };

template <int DUMMY>
__global__ __launch_bounds__(256) void gemm4w_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = g.N >> 8, ntm = (g.M + 255) >> 8, ntiles = ntm * ntn;
  const int nk = g.K >> 6;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // staging: wave w stages pieces w*8 .. w*8+7 of A and of B (piece = 8 rows x 128 B)
  const int srow = lane >> 3;
  const uint32_t lda2 = (uint32_t)g.lda * 2, ldb2 = (uint32_t)g.ldb * 2;
  const uint32_t c16_0 = (uint32_t)(((lane & 7) ^ (srow >> 1)) << 4);
  const uint32_t voffA = (uint32_t)(wave * 64 + srow) * lda2 + c16_0, voffB = (uint32_t)(wave * 64 + srow) * ldb2 + c16_0;
  // fragment addresses
  const int frow = lane & 15, fk = lane >> 4, sw = (frow >> 1) & 7;
  uint32_t pA[2], pB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ch = ((kk * 4 + fk) ^ sw) << 4;
    pA[kk] = lds0 + (wm * 128 + frow) * 128 + ch;
    pB[kk] = lds0 + BOFF + (wn * 128 + frow) * 128 + ch;
  }
  typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tm = tile / ntn, tn = tile - tm * ntn;
    const int m0 = tm << 8, n0 = tn << 8;
    const char* const Ab = reinterpret_cast<const char*>(g.A) + (size_t)(g.noepi == 2 ? (m0 & 0xfff) : m0) * lda2;   // noepi == 2: A rows from a 4096-row window (L2 / MALL resident)
    const char* const Bb = reinterpret_cast<const char*>(g.B) + (size_t)n0 * ldb2;
    // piece e of this wave: rows wave*64 + (e&7)*8 + srow.  Wave-uniform part of the address in SGPRs (base + K-tile + 8-row step),
    // per-lane part = (wave*64 + srow) * ld + swizzled chunk: two values per operand (even / odd piece)
    auto dma_piece = [&](int kt, int e, int buf) __attribute__((always_inline)) {     // e = 0..7: A piece, 8..15: B piece
      const int p = wave * 8 + (e & 7);
      if (e < 8) glds16_asm_lds(Ab + (size_t)kt * 128 + (size_t)((e & 7) * 8) * lda2, voffA ^ (uint32_t)((e & 1) << 6), lds0 + buf * STAGE + p * 1024);
      else glds16_asm_lds(Bb + (size_t)kt * 128 + (size_t)((e & 7) * 8) * ldb2, voffB ^ (uint32_t)((e & 1) << 6), lds0 + buf * STAGE + BOFF + p * 1024);
    };
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a0[8], b0[8], a1[8], b1[8];
    // prologue
#pragma unroll
    for (int e = 0; e < 16; ++e) dma_piece(0, e, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int e = 0; e < 16; ++e) dma_piece(min(1, nk - 1), e, 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) { a0[i] = *(lds_frag)(uintptr_t)(pA[0] + i * 2048); b0[i] = *(lds_frag)(uintptr_t)(pB[0] + i * 2048); }
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; ++kt) {
      const uint32_t cur = (kt & 1) * STAGE, nxt = ((kt + 1) & 1) * STAGE;
      // ---- phase 1: MFMAs of kk = 0, fragments of kk = 1 of the same K-tile fly underneath
#pragma unroll
      for (int gI = 0; gI < 16; ++gI) {
        if (gI < 8) a1[gI] = *(lds_frag)(uintptr_t)(pA[1] + cur + gI * 2048);
        else b1[gI - 8] = *(lds_frag)(uintptr_t)(pB[1] + cur + (gI - 8) * 2048);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = gI * 4 + q, i = t >> 3, j = t & 7;
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // K-tile kt + 1 landed (this wave's pieces); every read of buffer `cur` done
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase 2: MFMAs of kk = 1; K-tile kt + 2 is requested into buffer `cur`, fragments (kt + 1, kk = 0) are read
      // (past the end of the K range the requests degenerate to re-reads of the last K-tile into a buffer nobody reads any more and
      // the fragment reads fetch stale bytes nobody uses: no branches in the stream)
      const int kt2 = min(kt + 2, nk - 1);
#pragma unroll
      for (int gI = 0; gI < 16; ++gI) {
        dma_piece(kt2, gI, kt & 1);
        if (gI < 8) a0[gI] = *(lds_frag)(uintptr_t)(pA[0] + nxt + gI * 2048);
        else b0[gI - 8] = *(lds_frag)(uintptr_t)(pB[0] + nxt + (gI - 8) * 2048);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = gI * 4 + q, i = t >> 3, j = t & 7;
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // the next tile's prologue overwrites buffer 0
    if (g.noepi) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
      continue;
    }
    // plain epilogue (validation / first timing): D[row = 4 fk + r][col = frow] of every 16 x 16 block
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * 128 + i * 16 + fk * 4 + r, col = n0 + wn * 128 + j * 16 + frow;
          if (row < g.M) g.C[(size_t)row * g.ldc + col] = f2bf(acc[i][j][r]);
        }
  }
}

static unsigned short f2b(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float b2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  // ---- validation on a small ragged problem
  {
    const int M = 700, N = 512, K = 256, Mp = 768;
    std::vector<unsigned short> hA((size_t)Mp * K), hB((size_t)N * K), hC((size_t)Mp * N, 0);
    srand(3);
    for (auto& v : hA) v = f2b((rand() % 2001 - 1000) / 1000.f);
    for (auto& v : hB) v = f2b((rand() % 2001 - 1000) / 1000.f);
    bf16 *dA, *dB, *dC;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, hC.size() * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dC, 0, hC.size() * 2);
    Args g{dA, dB, dC, M, N, K, K, K, N, 0};
    hipLaunchKernelGGL(gemm4w_kernel<0>, dim3(4), dim3(256), 2 * STAGE, 0, g);
    hipDeviceSynchronize();
    hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int m = 0; m < M; m += 7) for (int n = 0; n < N; n += 5) {
      double s = 0; for (int k = 0; k < K; ++k) s += (double)b2f(hA[(size_t)m * K + k]) * b2f(hB[(size_t)n * K + k]);
      maxerr = fmax(maxerr, fabs(s - b2f(hC[(size_t)m * N + n])) / fmax(1.0, fabs(s)));
    }
    printf("validation: max rel err %.3e (%s)\n", maxerr, maxerr < 1e-2 ? "ok" : "WRONG");
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  // ---- timing
  struct Shape { int M, N, K; } shapes[] = {{8192, 8192, 8192}, {50208, 768, 3072}, {50208, 768, 2304}, {50208, 2304, 768}, {50208, 768, 768}};
  for (auto sh : shapes) {
    const int Mp = (sh.M + 255) / 256 * 256;
    const int NS = 3;
    bf16 *dA[NS], *dB, *dC[NS];
    std::vector<unsigned short> h((size_t)Mp * sh.K);
    for (auto& v : h) v = (unsigned short)(((rand() & 1) << 15) | ((120 + rand() % 8) << 7) | (rand() & 127));
    for (int s = 0; s < NS; ++s) { hipMalloc(&dA[s], (size_t)Mp * sh.K * 2); hipMemcpy(dA[s], h.data(), (size_t)Mp * sh.K * 2, hipMemcpyHostToDevice); hipMalloc(&dC[s], (size_t)Mp * sh.N * 2); }
    hipMalloc(&dB, (size_t)sh.N * sh.K * 2); hipMemcpy(dB, h.data(), (size_t)sh.N * sh.K * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int noepi = 0; noepi < 3; ++noepi) {
      auto run = [&](int s) { Args g{dA[s], dB, dC[s], sh.M, sh.N, sh.K, sh.K, sh.K, sh.N, noepi}; hipLaunchKernelGGL(gemm4w_kernel<0>, dim3(256), dim3(256), 2 * STAGE, 0, g); };
      for (int i = 0; i < 4; ++i) run(i % NS);
      hipDeviceSynchronize();
      const int reps = 12;
      hipEventRecord(e0); for (int i = 0; i < reps; ++i) run(i % NS); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("M %6d N %5d K %5d %s: %8.1f us  %7.0f TFLOP/s\n", sh.M, sh.N, sh.K, noepi == 2 ? "no epi, A hot " : noepi ? "no epilogue   " : "plain epilogue", ms / reps * 1e3,
             2.0 * sh.M * sh.N * sh.K / (ms / reps) / 1e9);
    }
    for (int s = 0; s < NS; ++s) { hipFree(dA[s]); hipFree(dC[s]); } hipFree(dB);
  }
  return 0;
}
