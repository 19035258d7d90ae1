// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of liboatrans_hip.so.
// Wave = 64 lanes; MFMA 16x16x32 bf16 fragments:
//   A operand: lane l holds A[row = l & 15][k = (l >> 4) * 8 .. +8]   (8 contiguous bf16 = 16 B)
//   B operand: lane l holds B[k = (l >> 4) * 8 .. +8][col = l & 15]
//   C/D      : lane l holds D[row = (l >> 4) * 4 + r][col = l & 15], r = 0..3
#pragma once
#include <hip/hip_runtime.h>

// One target.  Several kernels hand data between workgroups with returnless atomics + `s_waitcnt vmcnt(0)` and no fence
// (rowops.hip: ln_fold_grads, attn_space.hip: the fused CLS-row finalize) and count LDS-DMA / store retirement with vmcnt:
// that is gfx950 behaviour (vmcnt covers stores and returnless atomics, atomics execute at the memory side) and would be
// wrong on a target with a separate store counter.  The device pass of any other architecture must not compile.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "liboatrans_hip is written for gfx950 (MI355X / CDNA4) only: build with --offload-arch=gfx950"
#endif
#include <stdint.h>
#include <functional>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define OAT_DEV __device__ __forceinline__

namespace oat {

void set_error(const char* msg);     // defined in capi.hip; thread-local message
int check_launch(const char* what);  // hipGetLastError -> error code

// ---- launch tape (tape.hip).  Every kernel launch of the library goes through oat::launch.  While a tape is being
// recorded on the calling thread (oat_tape_begin .. oat_tape_end) the launch is executed AND appended to the tape with
// its arguments; oat_tape_replay re-issues the recorded launches from C, with no Python, ctypes or argument marshalling
// in between (a training step is ~1100 launches; issued one by one from Python they cost 20-45 ms of host time).
bool tape_recording();
void tape_push(std::function<void()>&& op);
template <class K, class... A>
inline void launch(K kernel, dim3 grid, dim3 block, unsigned lds, hipStream_t s, A... args) {
  hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
  if (tape_recording()) tape_push([=] { hipLaunchKernelGGL(kernel, grid, block, lds, s, args...); });
}
#define OAT_LAUNCH(...) ::oat::launch(__VA_ARGS__)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: the "already set" memo of a launch site is
// keyed on the current device, so every GPU a process drives gets it (one process per GPU is the design; a test or a tool may not be).
inline bool first_use_on_device(bool (&seen)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  if (seen[dev]) return false;
  seen[dev] = true;
  return true;
}
#define OAT_MAX_LDS(kernel, bytes)                                                                                              \
  do {                                                                                                                          \
    static bool seen_[64] = {};                                                                                                 \
    if (::oat::first_use_on_device(seen_))                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));   \
  } while (0)

OAT_DEV float bf2f(bf16 v) { return static_cast<float>(v); }
OAT_DEV bf16 f2bf(float v) { return static_cast<bf16>(v); }

OAT_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
OAT_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below bf16 resolution).
OAT_DEV float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);      // v_rcp_f32 (1 ulp); __frcp_rn expands to the 12-instruction IEEE division
  float p = 1.061405429f;
  p = p * t - 1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t - 0.284496736f;
  p = p * t + 0.254829592f;
  const float y = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(y, x);
}
// exact-erf GELU (reference: nn.GELU default, video_transformer.py:37)
OAT_DEV float gelu_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118f)); }
OAT_DEV float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118f));
  return cdf + x * 0.3989422804f * __expf(-0.5f * x * x);
}
// gelu(x) and gelu'(x) from ONE erf / exp evaluation (exp(-x^2/2) is both the A-S tail and the normal density).
// Since round 6 only the fp32 lanes call this (linear_f32.hip: the CLS lane and the text tower, where the absolute 1.5e-7 matters and
// the rate does not); the bf16 GEMM epilogues use gelu_pair below.
// The GEMM epilogues that called this are VALU-bound (ISA count: 26 issue slots per element, 8 of them the two quarter-rate
// transcendentals), so the constants are folded: m = |x| sqrt(log2(e)/2) makes exp(-x^2/2) = exp2(-m^2) (v_exp_f32 IS
// exp2) and 0.3275911 |x|/sqrt(2) = C m; the A-S coefficients are halved so that 0.5 (1 - p t e) is one fma.
OAT_DEV void gelu_both(float x, float& gl, float& dg) {
  const float m = fabsf(x) * 0.8493218002880191f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(m, 0.2727374808792225f, 1.0f));   // v_rcp_f32 (1 ulp)
  float p = 0.5307027145f;
  p = __builtin_fmaf(p, t, -0.7265760135f);
  p = __builtin_fmaf(p, t, 0.7107068705f);
  p = __builtin_fmaf(p, t, -0.142248368f);
  p = __builtin_fmaf(p, t, 0.127414796f);
  const float e = __builtin_amdgcn_exp2f(-m * m);
  const float cdf = 0.5f + copysignf(__builtin_fmaf(-(p * t), e, 0.5f), x);
  gl = x * cdf;
  dg = __builtin_fmaf(x * 0.3989422804f, e, cdf);
}

// ---- GELU of the bf16 GEMM epilogues (round 6): gelu(x) and gelu'(x) - 1/2 for TWO elements on packed fp32 math.
// The fc1 epilogue is VALU-bound (the matrix pipe idles while a tile is finished: 12.5 of a K = 768 tile's 31 us), so what
// counts is issue slots per element.  gelu_both above costs 20.75 (ISA count: two quarter-rate transcendentals = 8, the
// A-S polynomial in t = 1 / (1 + p |x|), sign handling, 8-bit pack); this form costs 16:
//     a = min(|x|, 6)       e = exp2(-a^2 log2(e) / 2) = exp(-a^2 / 2)        q(a) = Phi(-a) exp(a^2 / 2) = erfcx(a / sqrt 2) / 2
//     Phi(-a) = e q(a)      gelu(x) = max(x, 0) - a e q(a)                    gelu'(x) - 1/2 = copysign((a / sqrt(2 pi) - q(a)) e + 1/2, x)
// ONE transcendental (v_exp_f32), no reciprocal: q is smooth and well conditioned on [0, 6] - a degree-8 minimax polynomial in a
// holds it to 3.0e-4 RELATIVE (Lawson iteration on 6000 Chebyshev nodes, scripts/dev/gelu_fit.py), so Phi(-a) keeps its relative
// accuracy down the tail where the A-S form (absolute error 1.5e-7) is 5e-3 off at x = -4 and 50 % at x = -5.  Everything but
// min / max / exp / copysign is v_pk_{fma,mul}_f32 (two elements per issue slot).  (a/sqrt(2 pi) - q) e + 1/2 >= 0 for every a
// (it is 0 at a = 0 and rises), so one copysign carries the derivative's odd part.  Beyond |x| = 6 the tail term is frozen at
// 6 Phi(-6) = 6e-9 (a is clamped: no overflow of the polynomial, never a NaN from a finite input).
// Against fp64 erf-GELU on 2e6 points of [-8, 8] (tests/test_kernels_gpu.py::test_gelu_pair_dense_sweep): the bf16 result is within
// 0.58 bf16 ulp or 1e-7 absolute (A-S form: 1.8 ulp), gelu' within 1.5e-4 absolute.
// nn.GELU default = exact erf (video_transformer.py:37,45-51).
OAT_DEV f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
OAT_DEV f32x2 pk_bc(float v) { return f32x2{v, v}; }
OAT_DEV void gelu_pair(const f32x2 x, f32x2& gl, f32x2& dgh) {
  // fminf / fmaxf canonicalise their operands first (one more v_max each).  a: v_med3_f32 does not, takes |x| as a source modifier and
  // -1.0 as an inline constant - median(|x|, -1, 6) = min(|x|, 6); max(x, 0) has no such form hipcc leaves alone (it folds
  // median(x, 0, inf) back into a canonicalising fmax), so that one instruction is spelled out
  const f32x2 a = {__builtin_amdgcn_fmed3f(__builtin_fabsf(x[0]), -1.0f, 6.0f), __builtin_amdgcn_fmed3f(__builtin_fabsf(x[1]), -1.0f, 6.0f)};
  f32x2 r;
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(r[0]) : "v"(x[0]));
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(r[1]) : "v"(x[1]));
  const f32x2 earg = (a * a) * -0.72134752f;
  const f32x2 e = {__builtin_amdgcn_exp2f(earg[0]), __builtin_amdgcn_exp2f(earg[1])};
  f32x2 q = pk_fma(pk_bc(2.672734809e-06f), a, pk_bc(-7.911286957e-05f));
  q = pk_fma(q, a, pk_bc(1.008693944e-03f));
  q = pk_fma(q, a, pk_bc(-7.307060994e-03f));
  q = pk_fma(q, a, pk_bc(3.360059857e-02f));
  q = pk_fma(q, a, pk_bc(-1.048302799e-01f));
  q = pk_fma(q, a, pk_bc(2.347224802e-01f));
  q = pk_fma(q, a, pk_bc(-3.954404891e-01f));
  q = pk_fma(q, a, pk_bc(4.998524785e-01f));
  gl = pk_fma(-a, e * q, r);
  const f32x2 z = pk_fma(pk_fma(a, pk_bc(0.3989422804f), -q), e, pk_bc(0.5f));
  dgh = f32x2{__builtin_copysignf(z[0], x[0]), __builtin_copysignf(z[1], x[1])};
}
// four elements: gelu and the full derivative
OAT_DEV void gelu_quad(const f32x4 v, f32x4& gl, f32x4& dg) {
  f32x2 g0, d0, g1, d1;
  gelu_pair(f32x2{v[0], v[1]}, g0, d0);
  gelu_pair(f32x2{v[2], v[3]}, g1, d1);
  d0 += 0.5f;
  d1 += 0.5f;
  gl = f32x4{g0[0], g0[1], g1[0], g1[1]};
  dg = f32x4{d0[0], d0[1], d1[0], d1[1]};
}

// 16-byte async global -> LDS copy: lane i's 16 B land at lds_base + 16 * i.
OAT_DEV void glds16(const void* gptr, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Same copy issued from inline asm: hipcc does not see it, so it neither counts it nor inserts its own
// "s_waitcnt vmcnt(0)" in front of later ds_reads (which drains a multi-stage ring every iteration).
// The CALLER owns the vmcnt accounting: counted s_waitcnt vmcnt(N) + s_barrier before the data is read.
OAT_DEV void glds16_asm(const void* gptr, void* lds_wave_base) {
  const uint32_t lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)lds_wave_base));
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(lds) : "memory");
}

// Same, with the address split into a wave-uniform 64-bit base (SGPR pair) and a per-lane 32-bit byte offset: half the
// address VGPRs of the flat form and no 64-bit VALU add per piece.
OAT_DEV void glds16_asm_so(const void* uniform_base, uint32_t lane_byte_off, void* lds_wave_base) {
  const uint32_t lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)lds_wave_base));
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_byte_off), "s"(uniform_base), "s"(lds) : "memory");
}

// Same, destination given as a wave-uniform 32-bit LDS byte address (no generic-pointer cast, no null check).
OAT_DEV void glds16_asm_lds(const void* uniform_base, uint32_t lane_byte_off, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_byte_off), "s"(uniform_base), "s"(lds_addr) : "memory");
}

// LDS transpose read: within each 16-lane group, lane s fetches 4 contiguous bf16 at its own
// address; output lane i, element j = fetched[lane 4*j + (i >> 2)][i & 3].
OAT_DEV s16x4 lds_tr16(const void* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_ptr);
}

}  // namespace oat
