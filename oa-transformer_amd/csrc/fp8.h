// Shared pieces of the fp8 (OCP e4m3fn) path: saturating packed conversion and the per-site amax commit (see fp8.hip).
#pragma once
#include "common.h"

namespace oat {

constexpr float F8_MAX = 448.f;

OAT_DEV uint32_t pack_fp8x4(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(a, -F8_MAX), F8_MAX), fminf(fmaxf(b, -F8_MAX), F8_MAX), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(c, -F8_MAX), F8_MAX), fminf(fmaxf(d, -F8_MAX), F8_MAX), w, true);
  return (uint32_t)w;
}
// e5m2 ("bf8", largest finite 57344): the gradient format of the backward GEMMs
constexpr float BF8_MAX = 57344.f;
OAT_DEV uint32_t pack_bf8x4(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_bf8_f32(fminf(fmaxf(a, -BF8_MAX), BF8_MAX), fminf(fmaxf(b, -BF8_MAX), BF8_MAX), w, false);
  w = __builtin_amdgcn_cvt_pk_bf8_f32(fminf(fmaxf(c, -BF8_MAX), BF8_MAX), fminf(fmaxf(d, -BF8_MAX), BF8_MAX), w, true);
  return (uint32_t)w;
}
template <bool E5M2>
OAT_DEV uint32_t pack_f8x4(float a, float b, float c, float d) {
  if constexpr (E5M2) return pack_bf8x4(a, b, c, d);
  else return pack_fp8x4(a, b, c, d);
}
// One atomic per 256-thread block at most, and none when the block cannot raise the value: atomics on ONE address
// serialise in the L2 (16 k of them - one per wave of a 4096-block launch - cost ~130 us, seven times the kernel itself).
OAT_DEV void amax_commit(float m, float* amax) {
  __shared__ float part[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    const unsigned int bits = __float_as_uint(m);
    if (bits > __atomic_load_n(reinterpret_cast<unsigned int*>(amax), __ATOMIC_RELAXED))
      atomicMax(reinterpret_cast<unsigned int*>(amax), bits);
  }
}
// wave-level variant for kernels whose waves do not meet at a block barrier (GEMM epilogues)
OAT_DEV void amax_commit_wave(float m, float* amax) {
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) {
    const unsigned int bits = __float_as_uint(m);
    if (bits > __atomic_load_n(reinterpret_cast<unsigned int*>(amax), __ATOMIC_RELAXED))
      atomicMax(reinterpret_cast<unsigned int*>(amax), bits);
  }
}

}  // namespace oat
