// Dev probe: what the matrix pipes sustain under the power cap on RANDOM bf16 operands, per MFMA shape, with no memory traffic
// at all (operands live in registers).  hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// LDSR > 0: per 16 MFMAs, LDSR ds_read_b128 of random data refresh the operand fragments (the ping-pong GEMM reads 6 per 16 MFMAs,
// a 128 x 128 wave tile would read 4): what the LDS -> register traffic costs under the power cap
template <int LDSR>
__global__ __launch_bounds__(512) void probe_lds(const bf16x8* src, float* sink, int iters) {
  __shared__ bf16x8 tile[64 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 64 * 64; i += 512) tile[i] = src[i];
  __syncthreads();
  bf16x8 f[8];
  for (int i = 0; i < 8; ++i) f[i] = tile[(wave * 8 + i) * 64 % 4096 + lane];
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  int cur = wave * 64;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < LDSR; ++r) { f[r] = tile[(cur + lane) & 4095]; cur += 64; }
#pragma unroll
    for (int i = 0; i < 16; ++i)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(f[i & 3]), "v"(f[4 + ((i >> 2) + i) % 4]));
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) sink[0] = s;
}

template <int SHAPE, int NFRAG>   // SHAPE 0: 16x16x32 (16 accumulators), 1: 32x32x16 (8 accumulators of 16 regs)
__global__ __launch_bounds__(512) void probe(const bf16x8* src, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[NFRAG], b[NFRAG];
  for (int i = 0; i < NFRAG; ++i) { a[i] = src[(blockIdx.x * 7 + i) % 64 * 64 + lane]; b[i] = src[(blockIdx.x * 3 + i + 11) % 64 * 64 + lane]; }
  if constexpr (SHAPE == 0) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i % NFRAG]), "v"(b[(i / 4 + i) % NFRAG]));
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
  } else {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i % NFRAG]), "v"(b[(i / 2 + i) % NFRAG]));
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.678f) sink[0] = s;
  }
}

int main(int argc, char** argv) {
  const int zero = argc > 1 ? atoi(argv[1]) : 0;
  std::vector<unsigned short> h(64 * 64 * 8);
  srand(1);
  for (auto& v : h) {
    // bf16 of a value in roughly N(0, 1): random sign, exponent 120..128, random 7-bit mantissa
    v = zero ? 0 : (unsigned short)(((rand() & 1) << 15) | ((120 + rand() % 8) << 7) | (rand() & 127));
  }
  bf16x8* src; float* sink;
  hipMalloc(&src, h.size() * 2); hipMalloc(&sink, 4);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * (argc > 2 ? atoi(argv[2]) : 1), iters = 20000;
  for (int rep = 0; rep < 3; ++rep)
    for (int shape = 0; shape < 2; ++shape) {
      auto run = [&]() {
        if (shape == 0) probe<0, 4><<<grid, 512>>>(src, sink, iters);
        else probe<1, 4><<<grid, 512>>>(src, sink, iters);
      };
      run(); hipDeviceSynchronize();
      hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)grid * 8 * iters * (shape == 0 ? 16 * 2.0 * 16 * 16 * 32 : 8 * 2.0 * 32 * 32 * 16);
      printf("%s operands, %s: %.2f ms, %.0f TFLOP/s\n", zero ? "zero" : "random", shape == 0 ? "16x16x32" : "32x32x16", ms, flop / ms / 1e9);
    }
  for (int rep = 0; rep < 2; ++rep)
    for (int l = 0; l < 4; ++l) {
      auto run = [&]() {
        if (l == 0) probe_lds<0><<<grid, 512>>>(src, sink, iters);
        else if (l == 1) probe_lds<4><<<grid, 512>>>(src, sink, iters);
        else if (l == 2) probe_lds<6><<<grid, 512>>>(src, sink, iters);
        else probe_lds<8><<<grid, 512>>>(src, sink, iters);
      };
      run(); hipDeviceSynchronize();
      hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)grid * 8 * iters * 16 * 2.0 * 16 * 16 * 32;
      printf("%s operands, 16x16x32 + %d ds_read_b128 per 16 MFMAs: %.2f ms, %.0f TFLOP/s\n", zero ? "zero" : "random", l == 0 ? 0 : l == 1 ? 4 : l == 2 ? 6 : 8, ms, flop / ms / 1e9);
    }
  return 0;
}
