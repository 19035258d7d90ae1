#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// A: [16][128] bytes row-major (fp8 e4m3), B: [16][128] (N x K), out C[16][16]
__global__ void k(const uint8_t* A, const uint8_t* B, float* C, int scale_a, int scale_b) {
  const int l = threadIdx.x;
  i32x8 a, b;
  const int r = l & 15, kb = (l >> 4) * 32;
  for (int i = 0; i < 8; ++i) {
    a[i] = *reinterpret_cast<const int*>(A + r * 128 + kb + 4 * i);
    b[i] = *reinterpret_cast<const int*>(B + r * 128 + kb + 4 * i);
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
  for (int i = 0; i < 4; ++i) C[((l >> 4) * 4 + i) * 16 + (l & 15)] = c[i];
}
__global__ void cvt(const float* x, uint8_t* y, int n) {
  int i = threadIdx.x + blockIdx.x * blockDim.x;
  if (i * 4 < n) {
    int packed = 0;
    packed = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i], x[4 * i + 1], packed, false);
    packed = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i + 2], x[4 * i + 3], packed, true);
    *reinterpret_cast<int*>(y + 4 * i) = packed;
  }
}
extern "C" void run(const void* A, const void* B, float* C, int sa, int sb) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const uint8_t*)A, (const uint8_t*)B, C, sa, sb); }
extern "C" void run_cvt(const float* x, void* y, int n) { hipLaunchKernelGGL(cvt, dim3((n / 4 + 63) / 64), dim3(64), 0, 0, x, (uint8_t*)y, n); }
