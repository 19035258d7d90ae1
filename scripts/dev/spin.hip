// Dev experiment: a CU-filling MFMA kernel with a chosen register footprint, to test whether an HBM-bound kernel of
// another stream can co-reside with it (see DESIGN.md "tried and measured").  Not part of the product library.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int VG>
__global__ __launch_bounds__(512) void spin_kernel(float* out, int iters) {
  extern __shared__ char smem[];
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f); b[e] = (__bf16)1.0f; }
  if (VG == 192) asm volatile("v_mov_b32 v191, 0" ::: "v191");
  if (VG == 232) asm volatile("v_mov_b32 v231, 0" ::: "v231");
  if (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  for (int it = 0; it < iters; ++it)
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (s == 123.456f) out[0] = s + smem[threadIdx.x];
}
extern "C" int spin_launch(int vg, int blocks, int lds, int iters, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (vg == 192) { hipFuncSetAttribute((const void*)&spin_kernel<192>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(spin_kernel<192>, dim3(blocks), dim3(512), lds, s, out, iters); }
  else if (vg == 232) { hipFuncSetAttribute((const void*)&spin_kernel<232>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(spin_kernel<232>, dim3(blocks), dim3(512), lds, s, out, iters); }
  else { hipFuncSetAttribute((const void*)&spin_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(spin_kernel<128>, dim3(blocks), dim3(512), lds, s, out, iters); }
  return (int)hipGetLastError();
}
