// bf16 MFMA GEMM, "NT" form:  C[M,N] = A[M,K] * B[N,K]^T  (+ fused epilogue)
//
// Replaces the ATen linear calls on the hot path of the reference:
//   qkv / proj   /root/reference/OATrans/model/video_transformer.py:102,133
//   fc1 / fc2    /root/reference/OATrans/model/video_transformer.py:46-50
//   vid_proj/txt_proj  /root/reference/OATrans/model/oa_model.py:68-74
// and their data-gradients in backward (B = W^T shadow).
//
// gfx950 design (v1, "2-phase"): 128x128x64 tile, 256 threads = 4 waves (2x2), each wave
// a 64x64 sub-tile = 4x4 MFMA 16x16x32 tiles (64 fp32 accumulators / lane).  Both operands
// are K-contiguous, so A and B fragments are one ds_read_b128 each.  Tiles are staged with
// 16-byte global_load_lds (no VGPR round trip) into a double-buffered 64 KB LDS image; the
// LDS image is lane-linear, so the bank-conflict swizzle (chunk ^= (row >> 1) & 7 inside a
// 128-byte row) is applied to the per-lane GLOBAL source address and again on the read.
// MFMA operands are swapped (acc = mfma(Bfrag, Afrag)) so each lane owns 4 CONSECUTIVE
// output columns of one row: bias/residual loads are float4 and stores are 8/16 bytes.
#include "common.h"

namespace oat {

enum GemmEpi : int {
  EPI_BF16 = 0,       // out(bf16) = acc (+bias)
  EPI_F32 = 1,        // out(f32)  = acc (+bias) (+resid[row % resid_mod])
  EPI_GELU_DUAL = 2,  // out(bf16) = h = acc + bias ; out2(bf16) = gelu(h)
  EPI_DGELU = 3,      // out(bf16) = acc * gelu'(aux[row, col])   (aux = saved pre-activation h)
  EPI_F32_BF16 = 4,   // EPI_F32 plus a bf16 copy in out2
};

struct GemmArgs {
  const bf16* A; const bf16* B;
  int M, N, K, lda, ldb;
  void* out; int ldc;
  void* out2; int ld2;
  const float* bias;
  const float* resid; int ldr; int resid_mod;
  const bf16* aux; int ldaux;
};

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;   // 32 KB

// physical byte offset of logical 16B-chunk `lc` (0..7) of row r in a [rows][64 bf16] tile
OAT_DEV int swz(int r, int lc) { return r * 128 + ((lc ^ ((r >> 1) & 7)) << 4); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: consecutive ids on one XCD share the A row-panel (M-major walk of N).
  const int ntn = (g.N + BN - 1) / BN;
  const int ntm = (g.M + BM - 1) / BM;
  int bid = blockIdx.x;
  {
    const int nwg = ntm * ntn;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective remap
  }
  const int tm = bid / ntn, tn = bid % ntn;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging addresses: wave w stages rows [w*32, w*32+32) of A and of B, 4 glds each
  const int srow = lane >> 3;                   // row within an 8-row slab
  const bf16* a_src[4];
  const bf16* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 32 + i * 8 + srow;
    const int lc = (lane & 7) ^ ((r >> 1) & 7);            // inverse swizzle on the source side
    const int ra = min(m0 + r, g.M - 1);                   // clamp: ragged M/N read a valid row
    const int rb = min(n0 + r, g.N - 1);
    a_src[i] = g.A + (size_t)ra * g.lda + lc * 8;
    b_src[i] = g.B + (size_t)rb * g.ldb + lc * 8;
  }
  auto stage = [&](int buf, int k0) {
    char* base = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(a_src[i] + k0, base + (wave * 32 + i * 8) * 128);
      glds16(b_src[i] + k0, base + BM * 128 + (wave * 32 + i * 8) * 128);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  stage(0, 0);
  const int frow = lane & 15, fk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
    const char* sa = smem + (kt & 1) * STAGE_BYTES;
    const char* sb = sa + BM * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra = wm * 64 + i * 16 + frow;
        af[i] = *reinterpret_cast<const bf16x8*>(sa + swz(ra, kk * 4 + fk));
        const int rb = wn * 64 + i * 16 + frow;
        bfr[i] = *reinterpret_cast<const bf16x8*>(sb + swz(rb, kk * 4 + fk));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane owns C[row = .. + (lane & 15)][col = .. + (lane >> 4) * 4 + 0..3]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + wm * 64 + i * 16 + frow;
    if (row >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn * 64 + j * 16 + fk * 4;
      if (col >= g.N) continue;            // N is a multiple of 4 (checked on the host)
      f32x4 v = acc[i][j];
      if (g.bias) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(g.bias + col);
        v += b;
      }
      if constexpr (EPI == EPI_BF16) {
        bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *reinterpret_cast<bf16x4*>((bf16*)g.out + (size_t)row * g.ldc + col) = o;
      } else if constexpr (EPI == EPI_F32 || EPI == EPI_F32_BF16) {
        if (g.resid) {
          const int rr = g.resid_mod > 0 ? row % g.resid_mod : row;
          v += *reinterpret_cast<const f32x4*>(g.resid + (size_t)rr * g.ldr + col);
        }
        *reinterpret_cast<f32x4*>((float*)g.out + (size_t)row * g.ldc + col) = v;
        if constexpr (EPI == EPI_F32_BF16) {
          bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
          *reinterpret_cast<bf16x4*>((bf16*)g.out2 + (size_t)row * g.ld2 + col) = o;
        }
      } else if constexpr (EPI == EPI_GELU_DUAL) {
        bf16x4 h = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        // GELU is evaluated on the bf16-rounded pre-activation so that backward (which only
        // has the saved bf16 h) differentiates exactly the function that forward applied.
        bf16x4 a = {f2bf(gelu_f(bf2f(h[0]))), f2bf(gelu_f(bf2f(h[1]))),
                    f2bf(gelu_f(bf2f(h[2]))), f2bf(gelu_f(bf2f(h[3])))};
        *reinterpret_cast<bf16x4*>((bf16*)g.out + (size_t)row * g.ldc + col) = h;
        *reinterpret_cast<bf16x4*>((bf16*)g.out2 + (size_t)row * g.ld2 + col) = a;
      } else if constexpr (EPI == EPI_DGELU) {
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(g.aux + (size_t)row * g.ldaux + col);
        bf16x4 o = {f2bf(v[0] * dgelu_f(bf2f(h[0]))), f2bf(v[1] * dgelu_f(bf2f(h[1]))),
                    f2bf(v[2] * dgelu_f(bf2f(h[2]))), f2bf(v[3] * dgelu_f(bf2f(h[3])))};
        *reinterpret_cast<bf16x4*>((bf16*)g.out + (size_t)row * g.ldc + col) = o;
      }
    }
  }
}

template <int EPI>
static int launch(const GemmArgs& g, hipStream_t s) {
  const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<EPI>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3(ntm * ntn), dim3(256), 2 * STAGE_BYTES, s, g);
  return check_launch("gemm_nt");
}

}  // namespace oat

extern "C" int oat_gemm_nt(const void* A, const void* B, int M, int N, int K, int lda, int ldb,
                           int epi, void* out, int ldc, void* out2, int ld2,
                           const float* bias, const float* resid, int ldr, int resid_mod,
                           const void* aux, int ldaux, void* stream) {
  using namespace oat;
  if (M <= 0 || N <= 0 || K <= 0) { set_error("gemm_nt: empty problem"); return -1; }
  if (K % BK != 0) { set_error("gemm_nt: K must be a multiple of 64"); return -2; }
  if (N % 4 != 0 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 4 != 0) {
    set_error("gemm_nt: N%4, lda%8, ldb%8, ldc%4 alignment required"); return -3;
  }
  if (!A || !B || !out) { set_error("gemm_nt: null pointer"); return -4; }
  GemmArgs g{(const bf16*)A, (const bf16*)B, M, N, K, lda, ldb, out, ldc, out2, ld2,
             bias, resid, ldr, resid_mod, (const bf16*)aux, ldaux};
  hipStream_t s = (hipStream_t)stream;
  switch (epi) {
    case EPI_BF16: return launch<EPI_BF16>(g, s);
    case EPI_F32: return launch<EPI_F32>(g, s);
    case EPI_GELU_DUAL:
      if (!out2) { set_error("gemm_nt: EPI_GELU_DUAL needs out2"); return -4; }
      return launch<EPI_GELU_DUAL>(g, s);
    case EPI_DGELU:
      if (!aux) { set_error("gemm_nt: EPI_DGELU needs aux"); return -4; }
      return launch<EPI_DGELU>(g, s);
    case EPI_F32_BF16:
      if (!out2) { set_error("gemm_nt: EPI_F32_BF16 needs out2"); return -4; }
      return launch<EPI_F32_BF16>(g, s);
    default: set_error("gemm_nt: unknown epilogue"); return -5;
  }
}
