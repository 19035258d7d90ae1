// Shared declarations of the "NT" GEMM family (gemm_nt.hip: lockstep 256x256 / 128x128 kernels and the dispatcher;
// gemm_nt_pp.hip: the two-wave-group "ping-pong" 256x256 kernel).
#pragma once
#include "common.h"

namespace oat {

enum GemmEpi : int {
  EPI_BF16 = 0,       // out(bf16) = acc (+bias)
  EPI_F32 = 1,        // out(f32)  = acc (+bias) (+resid[row % resid_mod])
  EPI_GELU_DUAL = 2,  // out(bf16) = h = acc + bias ; out2(bf16) = gelu(h)
  EPI_DGELU = 3,      // out(bf16) = acc * gelu'(aux[row, col])   (aux = saved pre-activation h)
  EPI_F32_BF16 = 4,   // EPI_F32 plus a bf16 copy in out2
  EPI_GELU_GRAD = 5,  // h = acc + bias (fp32) ; out(bf16) = gelu'(h) ; out2(bf16) = gelu(h)
  EPI_MUL_AUX = 6,    // out(bf16) = (acc + bias) * aux[row, col]      (aux = the gelu'(h) saved by EPI_GELU_GRAD)
};

struct GemmArgs {
  const bf16* A; const bf16* B;
  int M, N, K, lda, ldb;
  void* out; int ldc;
  void* out2; int ld2;
  const float* bias;
  const float* resid; int ldr; int resid_mod;
  const bf16* aux; int ldaux;
  int dbg;
  int row0;            // first row of this launch inside the caller's problem (tail launches; used by resid_mod)
  int* ctr;            // unused since round 6 (kept so that the aggregate initialisers of the callers stay as they are)
  int* ctr_reset;
  const float* dq_a;   // fp8 launches: device scalars, dequantisation scale of A and of B (value = quantised * dq)
  const float* dq_b;
  void* out8; int ld8;     // fp8 launches: optional copy for the next GEMM - e4m3 of out2 (EPI_GELU_GRAD) / e5m2 of out (EPI_MUL_AUX) ...
  const float* q_out;      // ... quantised with this device scalar,
  float* amax_out;         // ... its max |value| recorded here
  int h_u8;                // the GELU-derivative tensor (out of EPI_GELU_GRAD / aux of EPI_MUL_AUX) is 8-bit fixed point,
                           // one byte per element, ldc / ldaux in bytes (ping-pong kernel only; gemm_nt_pp.hip HU8_*)
  float* sk_ws; int* sk_ctr;   // unused since round 6 (split-K of the last round left the library)
  int band;                    // ping-pong kernel, PPF_BAND: column tiles per band group of the per-XCD tile walk (0 = row-major walk)
};

constexpr int BK = 64;

struct TnArgs {
  const bf16* P; const bf16* Q;
  int M, N1, N2, ldp, ldq;
  float* slabs;            // [splits][N1][N2]
  float* bias_slabs;       // [splits][N1] or nullptr
  int chunks_per_split;    // in units of TK rows
  int splits;
  int dbg;                 // unused since round 6 (the run-time ablation bits left the kernels)
};

// gemm_tn_pp.hip: ping-pong 256x256 weight-gradient kernel.  Writes the same fp32 slabs as gemm_tn_kernel (the caller
// runs tn_reduce afterwards); g.chunks_per_split is in units of 64 rows there.
int launch_tn_pp(const TnArgs& g, int flags, hipStream_t s);

// gemm_nt_pp.hip.  pp_supported: does the ping-pong kernel cover this launch (epilogue, shape)?
// Per-call tuning of oat_gemm_nt (decoded from its `tune` / `grid` arguments): nothing of it lives in the library between calls.
struct GemmTune {
  int variant = 0;   // 0 auto, 1 force 128x128 tiles, 2 force the lockstep 256x256 kernel, 4 force the ping-pong kernel where it applies
  int grid = 0;      // persistent workgroups: 0 = one per CU, 0xffff = one workgroup per tile, else the count
  int m224 = 1;      // 224-row tiles of the ping-pong kernel: 0 never, 1 where they save a round's worth (default), 2 always
  int band = -1;     // band-grouped tile walk of the ping-pong kernel (PPF_BAND): column tiles per group, 0 = off, -1 = auto
};
bool pp_supported(int epi, const GemmArgs& g);
int launch_pp(int epi, const GemmArgs& g, int grid_slots, const GemmTune& t, hipStream_t s);
// the same kernel on OCP fp8 (e4m3) operands with per-tensor scales (GemmArgs::dq_a / dq_b)
bool pp_f8_supported(int epi, const GemmArgs& g);
int launch_pp_f8(int epi, const GemmArgs& g, int grid_slots, hipStream_t s);

}  // namespace oat
