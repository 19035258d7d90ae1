/* liboatrans_hip.so - C ABI of the MI355X (gfx950) kernels behind the OA-Transformer training
 * hot path.  The reference (FingerRec/OA-Transformer) is 100 % Python on ATen and has NO
 * native/FFI interface; each entry point below therefore cites the
 * reference *Python* lines whose ATen work it replaces.  A maintainer of the reference binds
 * these with ctypes - there are no torch types in any signature.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; oat_last_error() gives the message
 *     (thread-local).  Nothing is allocated, freed or synchronised inside the library: all
 *     buffers, including workspaces, are caller-owned device memory; every launch is ordered
 *     on the `stream` argument (a hipStream_t passed as void*).
 *   - no hidden state: what a launch does is fixed by its arguments (launch policy included: the `tune` / `grid` arguments of
 *     oat_gemm_nt / oat_gemm_tn).  No environment variable is read, no setter exists (oat_abi_version() >= 2).  The one thing the
 *     library keeps between calls is what the caller asks it to keep: launch tapes (oat_tape_begin .. oat_tape_end record the
 *     launches of the calling thread, oat_tape_replay re-issues them, oat_tape_free drops them).
 *   - bf16 tensors are passed as `void*` (raw uint16 storage), fp32 as `float*`.
 *   - token-row layout used engine-wide: patch (b,f,n) -> row (b*T+f)*N+n ; CLS(b) -> row
 *     B*T*N+b  (M = B*T*N + B rows).  ld* arguments are row pitches in ELEMENTS.
 *   - head_dim is fixed at 64 (ViT-B/16 and DistilBERT-base).
 */
#ifndef OATRANS_HIP_H
#define OATRANS_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* oat_last_error(void);
int oat_abi_version(void);
/* ---- GEMM -------------------------------------------------------------------------------
 * C[M,N] = A[M,K] * B[N,K]^T (+epilogue), bf16 in, fp32 accumulate.  K % 64 == 0.
 * Replaces nn.Linear forward / dgrad: video_transformer.py:102 (qkv), :133 (proj), :46-50
 * (fc1/GELU/fc2), oa_model.py:68-74 (txt_proj/vid_proj), and Conv2d patch-embed :69-75.
 * epi: 0 out(bf16)=acc+bias | 1 out(f32)=acc+bias+resid[row % resid_mod] |
 *      2 out(bf16)=h=acc+bias, out2(bf16)=gelu(h) | 3 out(bf16)=acc*gelu'(aux) |
 *      4 like 1 plus bf16 copy in out2 |
 *      5 h=acc+bias (fp32): out(bf16)=gelu'(h), out2(bf16)=gelu(h) | 6 out(bf16)=(acc+bias)*aux[row,col]
 *      (5 + 6 are the MLP pair the engine uses: the derivative is evaluated once, in forward, next to the
 *      activation it shares its erf / exp with; backward only multiplies).
 *      epi | 0x100 (with 5 / 6, shapes the ping-pong kernel covers: N % 256 == 0, N <= 4096, K / 64 even, M >= 256):
 *      the derivative tensor - `out` of 5, `aux` of 6 - is 8-bit fixed point, q = round((g' + 0.135) * 255 / 1.27),
 *      ONE byte per element with ldc / ldaux in bytes: half the traffic at bf16's own absolute error (0.0025).
 *      bias/resid/out2/aux may be NULL where unused.
 * Per-call launch policy (round 6: the library keeps NO tuning state between calls - the oat_gemm_set_* entry points of
 * rounds 1-5 are gone):
 *   tune  0 = the shipped choice.  bits 0-7: kernel - 0 auto, 1 force 128x128 tiles (4 waves), 2 force the lockstep 256x256
 *         kernel, 4 force the ping-pong 256x256 kernel where it applies (tests: both 256x256 kernels are bit-identical);
 *         bits 16-17: 224-row tiles of the ping-pong kernel - 0 auto (where rounds x tile rows is smaller than with 256-row
 *         tiles, e.g. M = 50208, N = 768: 3 rounds of 224 rows instead of 3 of 256), 1 never, 2 always; bits 18-25:
 *         band-grouped tile walk of the ping-pong kernel - 0 auto (the fc1 forward launch only, where it measures faster),
 *         1 off (row-major walk), n + 1 = groups of n column tiles.  Every choice gives bit-identical results.
 *   grid  workgroups of the persistent 256x256 launch: 0 = one per CU, 0xffff = one workgroup per tile (what a multi-rank
 *         job uses in backward, so that a persistent GEMM never waits for a CU RCCL holds), else the count. */
int oat_gemm_nt(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi,
                void* out, int ldc, void* out2, int ld2, const float* bias, const float* resid,
                int ldr, int resid_mod, const void* aux, int ldaux, int tune, int grid, void* stream);

/* out[N1,N2] (fp32, (+)=) sum_m P[m,N1]^T Q[m,N2]  - weight gradients of every nn.Linear.
 * Rows past M - 1 are never read (round 5: the ragged last chunk re-reads row M - 1; earlier builds required rows
 * [M, round_up(M,64)) to be readable, which a row slice ending at its CLS rows is not).
 * workspace: fp32 split-M slabs; oat_gemm_tn_workspace_bytes(M, .., tune) is exact for that launch, M <= 0 the worst case.
 * tune (per call, 0 = the shipped choice): tile shape - 0 auto, 1 force 128x128, 2 force the lockstep 256x256 kernel,
 * 4 force the ping-pong 256x256 kernel (N1, N2 multiples of 256). */
size_t oat_gemm_tn_workspace_bytes(int M, int N1, int N2, int tune);
int oat_gemm_tn(const void* P, const void* Q, int M, int N1, int N2, int ldp, int ldq, float* out,
                float* bias_out /* [N1] column sums of P, or NULL */, int accumulate, void* workspace,
                size_t workspace_bytes, int tune, void* stream);
/* ---- grouped weight gradients (csrc/gemm_tn_sk.hip): out_p (+)= P_p^T Q_p, bias_p (+)= colsum(P_p) for a LIST of
 * problems in ONE persistent launch + one fix-up launch.  Replaces the per-layer `dW = dY^T X` of autograd for the six
 * nn.Linear of a SpaceTimeBlock (/root/reference/OATrans/model/video_transformer.py:46-50,102,133) and the 36 of a
 * DistilBERT pass (HF DistilBertModel, call site model/oa_model.py:113).
 * Tables (same layout on host and device, little endian):
 *   OatTnProblem (64 B): const void* P, Q; float* out; float* bias_out (NULL: none); int M, N1, N2, ldp, ldq, accumulate, 0, 0
 *                        (N1, N2 multiples of 256; rows [M, round_up(M, 64)) of P and Q readable)
 *   OatTnSeg (32 B):     int prob, c1, c2, t2, kt0, n, slot (-1: whole tile, direct store), last
 *   OatTnFix (32 B):     int prob, c1, c2, t2, slot0, nslots, 0, 0
 * oat_tn_group_plan (host only, no GPU) -> segments, seg_off[blocks + 1], fix records; counts = {segments, fix records,
 * slabs, blocks}.  splits == 0: the sequence of K-tile pairs of all output tiles (tile-major) is cut into `grid` contiguous
 * shares (blocks = grid; for many small tiles).  splits >= 1: every tile is split that many ways over M (same M in every
 * problem), one segment per workgroup, blocks = tiles x splits in split-major, XCD-contiguous order (a row range of the
 * operands is fetched once per XCD); splits == 1 writes the gradients directly.  The plan depends on the shapes only.  The
 * caller copies the tables to device memory and provides oat_tn_group_slab_bytes(counts[2]) bytes; launch with grid = blocks. */
int oat_tn_group_plan(const void* problems, int n, int grid, int splits, void* segs_out, int seg_cap, int* seg_off,
                      void* fix_out, int fix_cap, int* counts);
size_t oat_tn_group_slab_bytes(int nslots);
int oat_tn_group_run(const void* d_problems, const void* d_segs, const void* d_seg_off, int grid,
                     const void* d_fixes, int nfix, void* d_slabs, void* stream);

/* ---- LayerNorm (video_transformer.py:164,167,174,346; DistilBERT LayerNorms) ------------- */
int oat_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, void* y_bf16,
                      int ldy, float* y_f32, int ldy32, float* mean, float* rstd, int M, int D,
                      float eps, void* stream);
/* s = x + add16 (bf16 branch output); sum32 = s (new fp32 residual stream, may alias x); y = LN(s) */
int oat_add_layernorm_fwd(const float* x, int ldx, const void* add16, int ldadd, float* sum32, int ldsum,
                          const float* gamma, const float* beta, void* y_bf16, int ldy, float* y_f32, int ldy32,
                          float* mean, float* rstd, int M, int D, float eps, void* stream);
/* the same with TWO bf16 addends: sum32 = x + add16 + add16b (out = x + space + mlp of a SpaceTimeBlock, video_transformer.py:170,175,
 * formed by the next block's first LayerNorm so that y = x + space is never stored) */
int oat_add2_layernorm_fwd(const float* x, int ldx, const void* add16, int ldadd, const void* add16b, int ldaddb,
                           float* sum32, int ldsum, const float* gamma, const float* beta, void* y, int ldy,
                           float* y32, int ldy32, float* mean, float* rstd, int M, int D, float eps, void* stream);
/* the same with an fp32 addend: s = x + add32 (the precise CLS lane of the video tower) */
int oat_add32_layernorm_fwd(const float* x, int ldx, const float* add32, int ldadd, float* sum32, int ldsum,
                            const float* gamma, const float* beta, void* y, int ldy, float* y32, int ldy32,
                            float* mean, float* rstd, int M, int D, float eps, void* stream);
int oat_ln_bwd_blocks(int M);   /* partial workspace = blocks * 2 * D floats */
int oat_layernorm_bwd(const void* dy, int dy_is_bf16, int lddy, const float* x, int ldx,
                      const float* mean, const float* rstd, const float* gamma, const float* dres,
                      int lddres, float* dx, int lddx, void* dx_bf16, int lddx16, int dx16_excl_res,
                      float* dgamma, float* dbeta, int accumulate, float* part, int M, int D, void* stream);
/* dx = LNbwd(dy) + dres ; dx_bf16 = bf16(dx) or, with dx16_excl_res, bf16(LNbwd(dy)) only. */

/* ---- LayerNorm on a bf16 residual stream (round 4) -------------------------------------------------------------------
 * The three residual adds of a SpaceTimeBlock (video_transformer.py:166,170,175) and the LayerNorms that follow them
 * (:164,167,174,346) with the token stream x STORED as bf16: s = x + add_a + add_b in fp32, sum16 = bf16(s) (the new stream),
 * y / y32 = LN(s) from the unrounded sum.  x is bf16, or fp32 when x_is_f32 (block 0 reads the patch embedding's fp32 output);
 * every addend / output is optional (NULL).  The cosine-similarity matrix is taken from the fp32 CLS lane and does not move;
 * parameter gradients differ from the fp32 oracle's by 2.0e-2 instead of 1.8e-2 relative L2 (scripts/dev/rounding_study3.py). */
int oat_layernorm_fwd_r16(const void* x, int x_is_f32, int ldx, const void* add_a_bf16, int ldadd_a, const void* add_b_bf16,
                          int ldadd_b, void* sum16, int ldsum, const float* gamma, const float* beta, void* y_bf16,
                          int ldy, float* y_f32, int ldy32, float* mean, float* rstd, int M, int D, float eps,
                          void* stream);
/* The same with an OCP e4m3 copy of y for an fp8 forward GEMM (config 5: fp8 forward linears ON the bf16 stream):
 * y8 = e4m3(y * *qscale) with the site's delayed scale, max |y| -> *amax for the next step's scale (cf. oat_layernorm_fwd_f8,
 * the fp32-stream form).  y (bf16) is still written: backward and the weight gradient read it. */
int oat_layernorm_fwd_r16_f8(const void* x, int x_is_f32, int ldx, const void* add_a_bf16, int ldadd_a, const void* add_b_bf16,
                             int ldadd_b, void* sum16, int ldsum, const float* gamma, const float* beta, void* y_bf16,
                             int ldy, void* y8_e4m3, int ld8, const float* qscale, float* amax, float* mean, float* rstd,
                             int M, int D, float eps, void* stream);
/* oat_layernorm_bwd with the forward input x as bf16 and a bf16 residual-gradient addend: dx (fp32 | NULL) and dx16 =
 * LNbwd(dy) + dres16; dres16 may be dx16 itself (in place).  The final norm and the region tap of the bf16 stream. */
int oat_layernorm_bwd_r16(const void* dy, int dy_is_bf16, int lddy, const void* x_bf16, int ldx, const float* mean,
                          const float* rstd, const float* gamma, const void* dres16, int lddres16, float* dx, int lddx,
                          void* dx16, int lddx16, float* dgamma, float* dbeta, int accumulate, float* part, int M, int D,
                          void* stream);

/* ---- reductions (bias / positional-table gradients) ------------------------------------- */
int oat_colsum_rows(int M);     /* partial workspace = rows * N floats */
int oat_colsum(const void* A, int is_bf16, int lda, int M, int N, float* out, int accumulate,
               float* part, void* stream);
int oat_periodic_rowsum(const float* in, int ld, int R, int P, int D, float* out, int accumulate, void* stream);
int oat_grouped_rowsum(const float* in, int ld, int G, int R, int D, float* out, int accumulate, void* stream);

/* ---- patch embedding inputs (video_transformer.py:71-76, 313-324) ----------------------- */
int oat_im2col(const void* video, int is_bf16, void* A_bf16, int BT, int C, int R, int ps, int lda, void* stream);
int oat_pos_table(const float* pos, const float* temporal, const float* cls_token, float* table,
                  float* cls0, int T, int N, int D, void* stream);
int oat_broadcast_rows(const float* src, float* dst, int ld, int R, int D, void* stream);
int oat_cast_bf16(const float* src, void* dst_bf16, void* dstT_bf16, int R, int C, void* stream);
/* All weight shadows of a module in one launch.  desc: device array of n_matrices records of NINE int64
 * {src f32*, dst bf16* | 0, dstT bf16* | 0, R, C, ldd, ldT, first_tile, colscale f32* | 0}; colscale[c] multiplies column c
 * before the cast (the LayerNorm scale folded into the following linear layer, W' = W diag(gamma)); matrix m owns the T x T tiles (T = oat_cast_bf16_tile())
 * [first_tile[m], first_tile[m+1]) of the grid, total_tiles blocks in all (the per-weight `.to(bfloat16)` /
 * `.t().contiguous()` / `torch.cat` ATen work a mixed-precision port of video_transformer.py:102,133,46-50 performs). */
int oat_cast_bf16_tile(void);   /* edge of the square tiles oat_cast_bf16_multi counts in (first_tile, total_tiles) */
int oat_cast_bf16_multi(const void* desc, int n_matrices, int total_tiles, const int* tile_matrix /* device int32[total_tiles]: matrix of each tile, or NULL (binary search per block) */,
                        void* stream);

/* ---- folded LayerNorm (norm1 / norm2 / norm3 of a SpaceTimeBlock, video_transformer.py:161-176) -------------------------
 * z = W (gamma * xhat + beta) + b  is executed as  z = W' xhat + b'  with W' = W diag(gamma) (colscale of
 * oat_cast_bf16_multi) and b' = b + W beta (oat_fold_bias_multi): oat_layernorm_fwd with gamma = beta = NULL then writes the
 * plain normalised row, the data-gradient GEMM of W'^T delivers d(xhat), and oat_layernorm_bwd_xhat needs the saved bf16 xhat
 * and rstd only (539 instead of 616 MB per call at M = 50208, no (dgamma, dbeta) partial sums).  The weight-gradient GEMM
 * on xhat gives dW' and db'; oat_ln_fold_grads turns them into dW = dW' diag(gamma) + db' beta^T, dgamma[k] =
 * sum_n W[n,k] dW'[n,k], dbeta[k] = sum_n W[n,k] db'[n].  Same function, same gradients as autograd of nn.LayerNorm + nn.Linear.
 *   oat_fold_bias_multi: desc = device array of {W, beta, b | 0, out, N, K, first_row, 0} (8 x 8 B); 4 rows per block;
 *                        block_desc[i] = descriptor of block i; first_row = 4 x the descriptor's first block
 *   oat_ln_fold_grads:   desc = device array of {dWp, dbp, W, gamma, beta, dW, db, dgamma, dbeta, N, K, first_block,
 *                        accumulate} (13 x 8 B); a layer owns oat_ln_fold_blocks(K) consecutive blocks; dW may alias dWp
 *                        (in place), db may alias dbp; work = total_blocks * 128 floats + total_blocks ints, the ints ZERO
 *                        before the first launch (ticket counters; every launch leaves them zero); deterministic */
/* add_a / add_b (bf16 | NULL): further addends of dx / dx16; dxp16 (bf16 | NULL): the plain result before any addend - so that
 * the fp32 residual-gradient stream is read by norm2's and read + written by norm3's backward only (norm1's touches none) */
int oat_layernorm_bwd_xhat(const void* dxh_bf16, int lddxh, const void* xhat_bf16, int ldxh, const float* rstd,
                           const float* dres, int lddres, float* dx, int lddx, void* dx16, int lddx16,
                           int dx16_excl_res, const void* add_a, int ldadd_a, const void* add_b, int ldadd_b,
                           void* dxp16, int lddxp, int M, int D, void* stream);
int oat_fold_bias_multi(const void* desc, const int* block_desc, int total_blocks, void* stream);
int oat_ln_fold_blocks(int K);
int oat_ln_fold_grads(const void* desc, int n_desc, int total_blocks, void* work, void* stream);

/* ---- divided space-time attention (video_transformer.py:99-135, :28-32) ------------------
 * qkv: bf16 [M, 3*D] (q | k | v, heads contiguous); out: bf16 [M, D]; lse: fp32 [M, H].
 * The CLS-query kernel must run before either backward (it writes lse/out of the CLS rows). */
int oat_attn_space_fwd(const void* qkv, int ldqkv, void* out, int ldo, float* lse, int B, int T, int N,
                       int H, int D, float scale, void* stream);
int oat_attn_time_fwd(const void* qkv, int ldqkv, void* out, int ldo, float* lse, int B, int T, int N,
                      int H, int D, float scale, void* stream);
int oat_attn_cls_fwd(const void* qkv, int ldqkv, void* out, int ldo, float* lse, int B, int T, int N,
                     int H, int D, float scale, void* stream);
/* CLS query attention with a second, PRECISE query: q32 fp32 [B, D] attends the same bf16 keys / values, context written
 * in fp32 to o32 [B, D].  out / lse of the CLS row (the bf16 path backward uses) as oat_attn_cls_fwd. */
int oat_attn_cls_fwd_dual(const void* qkv, int ldqkv, void* out, int ldo, float* lse, const float* q32, int ldq32,
                          float* o32, int ldo32, int B, int T, int N, int H, int D, float scale, void* stream);
/* cls_side: fp32 [B,H,3,64], zero on entry; finish with oat_attn_cls_finalize, which writes the CLS row of dqkv and
 * leaves cls_side zero again for the next backward launch. */
int oat_attn_space_bwd(const void* qkv, int ldqkv, const void* out, int ldo, const float* lse,
                       const void* dout, int lddo, void* dqkv, int lddqkv, float* cls_side, int B, int T,
                       int N, int H, int D, float scale, void* stream);
int oat_attn_time_bwd(const void* qkv, int ldqkv, const void* out, int ldo, const float* lse,
                      const void* dout, int lddo, void* dqkv, int lddqkv, float* cls_side, int B, int T,
                      int N, int H, int D, float scale, void* stream);
int oat_attn_cls_finalize(float* cls_side, void* dqkv, int lddqkv, int B, int T, int N, int H, int D,
                          void* stream);
/* The two steps above in ONE launch (replaces the same ATen work, video_transformer.py:99-135 backward): `done` = int [B,H]
 * tickets, zero on entry and on exit; the last workgroup that feeds cls_side[b][h] swaps the sums out (atomic exchange: cls_side
 * is left zero), writes the CLS row of dqkv and resets its ticket.  Backward: T <= 16 (the MFMA kernel on 16-row mini problems). */
int oat_attn_space_bwd_fin(const void* qkv, int ldqkv, const void* out, int ldo, const float* lse,
                           const void* dout, int lddo, void* dqkv, int lddqkv, float* cls_side, int* done, int B, int T,
                           int N, int H, int D, float scale, void* stream);
/* Space attention of TWO clips of one geometry (N, H, D, leading dimensions) in one launch each way: the object frame and the
 * video clip the OA models send through the same encoder (oa_model_global_local.py:170, oa_model_region_mem.py:120; the
 * reference runs them as one 2B-clip batch).  A one-frame clip alone is B x H problems - a third of the GPU.  n_clips = 1 or 2;
 * backward = oat_attn_space_bwd_fin per clip (every clip brings its cls_side and ticket buffers). */
typedef struct OatAttnClip {
  const void* qkv; void* out; float* lse;          /* rows of this clip (patches, then its B CLS rows) */
  const void* dout; void* dqkv; float* cls_side; int* done;   /* backward only (NULL in forward) */
  int B, T;
} OatAttnClip;
int oat_attn_space_fwd_clips(const OatAttnClip* clips, int n_clips, int ldqkv, int ldo, int N, int H, int D, float scale,
                             void* stream);
/* cls_query_only != 0 (round 6; the pruned top block of the engine, which consumes the encoder's CLS row alone:
 * video_transformer.py:349-351 -> oa_model.py:129-133): the caller guarantees that dO of every PATCH query is zero and its lse
 * is +inf (3.4e38), so only the CLS query carries a gradient; frames of 97..447 patches then run an instance that computes the
 * one query tile and query pair that hold it and writes zeros for the other dQ rows - bit-identical to the full launch, which
 * adds the same exact zeros (smaller frames run the full kernel either way). */
int oat_attn_space_bwd_clips(const OatAttnClip* clips, int n_clips, int ldqkv, int ldo, int lddo, int lddqkv, int N, int H,
                             int D, float scale, int cls_query_only, void* stream);
/* TIME attention backward + CLS-row finalize of TWO clips of different frame counts (powers of two <= 16) in one launch: the one-frame
 * object clip of the OA models (oa_model_global_local.py:170) beside the T-frame video clip; = oat_attn_time_bwd_fin per clip. */
int oat_attn_time_bwd_clips(const OatAttnClip* clips, int n_clips, int ldqkv, int ldo, int lddo, int lddqkv, int N, int H,
                            int D, float scale, void* stream);
int oat_attn_time_bwd_fin(const void* qkv, int ldqkv, const void* out, int ldo, const float* lse,
                          const void* dout, int lddo, void* dqkv, int lddqkv, float* cls_side, int* done, int B, int T,
                          int N, int H, int D, float scale, void* stream);

/* ---- fp32 linear layer for small row counts (exact-f32 MFMA 16x16x4; fp32 master weights, no bf16 shadow) ---------
 * out = act(in(A)[M,K] * W[N,K]^T + bias) (+ resid).  Carries the two places where bf16 operand rounding would break
 * the 1e-3 sim-matrix bound at negligible FLOPs: the text tower and the CLS row of the video tower (oa_model.py:106-133,
 * video_transformer.py:46-50,102,133 for those rows).  act 0: out32 = out16 = y; 1 (GELU): out32 = out16 = gelu(y),
 * out16b = gelu'(y); 2: ReLU on A while loading (txt_proj, oa_model.py:68-70).  out32 / out16 / out16b / bias / resid may
 * be NULL where unused (at least one of out32, out16).  K % 16 == 0, lda % 4 == 0, ldw % 4 == 0.
 * M <= 64 (CLS lane, projections): exact fp32 products.  M > 64 with K % 32 == 0 (the text tower): every fp32 operand element
 * is split into two bf16 (hi + lo, 16 mantissa bits) and the product runs as three bf16 MFMA passes with fp32 accumulation -
 * 2^-16 relative per product, 1e-5 of |y| max against fp64 (tests/test_kernels_gpu.py), 5x less matrix-pipe time; the launches
 * sit beside the video tower on a side stream and their duration is what they cost it.  act | OAT_LIN_EXACT: exact fp32 at every M.
 * act | 0x100 (OAT_LIN_EXACT): exact fp32 products at EVERY M - what the CLS lane and the projection heads pass, so that a plan of
 * more than 64 clips (batch 64, or two clips of more than 32 samples) keeps the lane's precision. */
#define OAT_LIN_NONE 0
#define OAT_LIN_GELU 1      /* GELU (exact erf) on the output */
#define OAT_LIN_RELU_IN 2   /* ReLU on A while loading */
#define OAT_LIN_EXACT 0x100 /* or-ed in: exact fp32 products at every M */
int oat_linear_f32(const float* A, int lda, const float* W, int ldw, const float* bias, int M, int N, int K,
                   float* out32, int ldo, void* out16, int ld16, void* out16b, int ld16b, const float* resid, int ldr,
                   int act, void* stream);
/* out[M, 3n] = A[M, K] . [Wq; Wk; Wv]^T + [bq | bk | bv]: the q_lin / k_lin / v_lin of a DistilBERT attention layer (three separate
 * nn.Linear parameters in HF 4.6 MultiHeadSelfAttention, reached from oa_model.py:27,113) as ONE launch of the split-bf16 kernel -
 * per output element the arithmetic of three oat_linear_f32 calls (bit-identical), three times the workgroups per launch.
 * n % 128 == 0, K % 32 == 0, M > 64, lda % 4 == 0, ldw % 4 == 0; biases: all three or none; out32 / out16: at least one. */
int oat_linear_f32_qkv(const float* A, int lda, const float* Wq, const float* Wk, const float* Wv, int ldw,
                       const float* bq, const float* bk, const float* bv, int M, int n, int K,
                       float* out32, int ldo, void* out16, int ld16, void* stream);

/* Backward of a FEW-row linear layer y = act(x) W^T + b (the projection heads txt_proj = ReLU -> Linear, vid_proj = Linear on B rows,
 * oa_model.py:66-78 and their autograd backward): dx [M, K] (NULL: not wanted), dW [N, K] dense, db [N] (NULL: no bias) from dy [M, N] in one
 * launch of fp32 arithmetic (deterministic row order).  relu_in: act = ReLU.  1 <= M <= 64, K % 16 == 0.  Written, not accumulated. */
int oat_linear_small_bwd(const float* x, int ldx, const float* dy, int lddy, const float* W, int ldw, int M, int N, int K,
                         int relu_in, float* dx, int lddx, float* dW, float* db, void* stream);

/* ---- text encoder (HF DistilBertModel, called at oa_model.py:113; third-party algorithm) ---------
 * ids / mask are int64.  Attention: qkv bf16 [B*L, 3*D]; masked keys are skipped. */
int oat_embed_fwd(const void* ids, const float* word, const float* pos, float* out, int ld, int M, int L,
                  int D, void* stream);
int oat_embed_bwd(const void* ids, const float* g, int ld, float* dword_zeroed, int M, int D, void* stream);
int oat_attn_text_fwd(const void* qkv, int ldqkv, const void* mask, void* out, int ldo, float* lse, int B,
                      int L, int H, int D, float scale, void* stream);
/* The same, plus the precise forward value: identical masked attention on the fp32 q|k|v in qkv32 (whose bf16 roundings
 * are `qkv`), written to out32.  out / lse remain the bf16-path results that oat_attn_text_bwd recomputes from. */
/* drop_p > 0 (training mode): dropout on the attention probabilities, mask element ((b*H + h)*L + i)*L + j of site
 * `drop_site` under the device rng state `rng` (see oat_rng_tick); the backward must be given the same three values. */
int oat_attn_text_fwd_dual(const void* qkv, int ldqkv, const float* qkv32, int ldqkv32, const void* mask, void* out,
                           int ldo, float* out32, int ldo32, float* lse, int B, int L, int H, int D, float scale,
                           float drop_p, const void* rng, unsigned drop_site, void* stream);
int oat_attn_text_bwd(const void* qkv, int ldqkv, const void* mask, const void* out, int ldo,
                      const float* lse, float* delta_scratch, const void* dout, int lddo, void* dqkv,
                      int lddqkv, int B, int L, int H, int D, float scale, float drop_p, const void* rng,
                      unsigned drop_site, void* stream);
/* ---- input pipeline, device side: decoded frames -> clip tensor.  Replaces, per batch and in one launch,
 * frames.float() / 255 (base/base_dataset.py:519,533,544) and the tensor transforms of data_loader/transforms.py:4-31 /
 * base_dataset_global_local.py:251-257: crop box + bilinear resize (= torchvision Resize on tensors =
 * F.interpolate(bilinear, align_corners=False)) + horizontal flip + Normalize.  frames: uint8 [F, H, W, 3] or float
 * [F, 3, H, W]; crop = {x0, y0, w, h} in source pixels or NULL; out [F, 3, OH, OW] bf16 / fp32; mean / std: 3 floats
 * (host pointers) or NULL. */
int oat_frames_resize(const void* frames, int in_u8_hwc, int F, int H, int W, const float* crop_xywh_or_null, int flip,
                      void* out, int out_bf16, int OH, int OW, float scale, const float* mean3, const float* std3, void* stream);

/* ---- launch tape: record the launches of a schedule once, replay them from C (csrc/tape.hip).  Stands where the reference
 * relies on PyTorch's eager dispatcher for every op of the step (model/video_transformer.py:303-351 forward, autograd for
 * backward).  Between oat_tape_begin and oat_tape_end (same thread) every oat_* launch is executed AND recorded with its
 * arguments; pointers and scalars are baked in, so a tape is valid while the buffers it touches stay where they are. */
int oat_tape_begin(void);
void oat_tape_abort(void);
void oat_tape_pause(int on);                  /* launches issued while paused run but are not recorded */
int oat_tape_mark(void);                      /* close a segment; returns its index */
int oat_tape_end(void);                       /* returns the tape id >= 0 */
int oat_tape_segments(int id);
int oat_tape_ops(int id);
int oat_tape_replay(int id, int seg_lo, int seg_hi);   /* segments [seg_lo, seg_hi); seg_hi < 0: to the end */
int oat_tape_free(int id);
/* stream / memory operations of a schedule, recordable like launches */
int oat_stream_edge(void* from_stream, void* to_stream);      /* `to` waits for what is on `from` now */
int oat_memset_async(void* dst, int byte_value, size_t bytes, void* stream);
int oat_copy_async(void* dst, const void* src, size_t bytes, void* stream);

/* ---- OCP fp8 (e4m3fn) forward GEMMs, per-tensor scaled (BASELINE.json config 5; stands where the bf16 oat_gemm_nt
 * serves the nn.Linear forwards of video_transformer.py:46-50,102,133).  A quantisation site owns three device floats:
 * amax (running max |x| of this step), qscale (q = sat(x * qscale)), dq = 1 / qscale. */
int oat_fp8_quant(const void* x, int is_bf16, int ldx, void* out8, int ld8, int M, int K, const float* qscale,
                  float* amax_or_null, void* stream);       /* quantise with *qscale (e4m3), record amax of x */
int oat_fp8_amax(const void* x, int is_bf16, int ldx, int M, int K, float* amax, void* stream);
int oat_fp8_chunk_elems(void);
/* many contiguous bf16 matrices in one launch; desc rows int64 {src, dst, n, site, first_block}, owner: block -> row */
int oat_fp8_multi(const void* desc, const int* owner, int total_blocks, const float* qscale, float* amax, int quant,
                  void* stream);
int oat_fp8_update_scales(float* amax, float* qscale, float* dq, int n_sites, float margin, void* stream);
/* C = dq_a dq_b (A8 . B8^T) + bias ; epi 0 (bf16 out) or 5 (out = gelu'(h), out2 = gelu(h)); K % 256 == 0,
 * N % 256 == 0, N <= 4096, M >= 256; lda / ldb in elements (bytes) */
int oat_gemm_nt_f8(const void* A8, const void* B8, int M, int N, int K, int lda, int ldb, int epi, void* out, int ldc,
                   void* out2, int ld2, const float* bias, const float* dq_a, const float* dq_b,
                   void* out8_or_null, int ld8, const float* q_out, float* amax_out,   /* epi 5: e4m3 copy of out2 for the next GEMM */
                   void* stream);
/* LayerNorm (optionally of x + add16, sum32 = the sum) writing y as bf16 AND as e4m3 (y8 = sat(y * *qscale)), amax of y
 * recorded: the producer-side quantisation of the fp8 GEMM operand */
int oat_layernorm_fwd_f8(const float* x, int ldx, const void* add16_or_null, int ldadd, float* sum32, int ldsum,
                         const float* gamma, const float* beta, void* y, int ldy, void* y8, int ld8, const float* qscale,
                         float* amax, float* mean, float* rstd, int M, int D, float eps, void* stream);
/* ---- dropout of the text tower's training mode (HF DistilBERT nn.Dropout sites: embeddings, attention probabilities,
 * ffn output; reference call site oa_model.py:56,113-121).  Counter-based Philox4x32-10 masks: element idx of `site`
 * draws word idx%4 of philox(counter = (idx/4 lo, idx/4 hi, site, offset), key = seed); rng = device uint64[2] =
 * {seed, offset}.  Nothing is stored: forward and backward regenerate the mask. */
int oat_rng_tick(void* rng, void* stream);                       /* offset += 1 (once per forward call) */
/* out = x * mask (+ resid), as fp32 (out32) and / or bf16 (out16); in place allowed */
int oat_dropout(const float* x, int ldx, const float* resid, int ldr, float* out32, int ldo, void* out16, int ld16, int M,
                int D, float p, const void* rng, unsigned site, void* stream);
int oat_dropout_mask(float* out, long long n, float p, const void* rng, unsigned site, void* stream);   /* multipliers of [0, n) */
int oat_philox4x32_10(const void* in_6words_each, void* out_4words_each, int n, void* stream);         /* known-answer access */
/* txt_proj's ReLU (oa_model.py:68) */
int oat_relu_bf16(const float* x, int ldx, void* y_bf16, int ldy, int M, int D, void* stream);
int oat_relu_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int M, int D, void* stream);

/* ---- loss + optimiser ---------------------------------------------------------------------------
 * sim_matrix (oa_model.py:192-200) + NormSoftmaxLoss (loss.py:13-25), forward and backward in one
 * call over the all-gathered embeddings; gradients only for local rows [r0, r0+nloc)
 * (AllGather_multi.backward, trainer_dist.py:41-45). */
size_t oat_sim_workspace_floats(int n, int m, int d);
int oat_sim_matrix_fwd(const float* t, const float* v, int n, int m, int d, float eps, float* sim, float* ws,
                       void* stream);
int oat_sim_matrix_bwd(const float* G, const float* ws, int n, int m, int d, float* dt, int t0, int tl,
                       float* dv, int v0, int vl, void* stream);
int oat_norm_softmax_loss(const float* sim, int n, float temperature, float* loss, float* G, float* ws_2n,
                          void* stream);
size_t oat_infonce_workspace_floats(int n, int d);
int oat_infonce(const float* t, const float* v, int n, int d, float temperature, float eps, float* loss,
                float* sim_out, float* dt, float* dv, int r0, int nloc, float* ws, void* stream);
/* transformers.AdamW (hf_style=1, train_dist_multi.py:66) or torch.optim.AdamW (0) over a flat range */
int oat_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
              float eps, float weight_decay, int step, int hf_style, float gscale, void* stream);
/* The same update with its step-dependent scalars on the device, so that a CAPTURED launch (hipGraph replay of a whole
 * training step) stays valid: oat_adam_tick does *step += 1 and writes coef = {*lr, 1 - beta1^step, 1 - beta2^step};
 * oat_adamw_dev reads lr and the bias corrections from coef.  step / lr / coef are caller-owned device memory. */
int oat_adam_tick(int* step, const float* lr, float beta1, float beta2, float* coef, void* stream);
int oat_adamw_dev(float* p, const float* g, float* m, float* v, size_t n, const float* coef, float beta1, float beta2,
                  float eps, float weight_decay, int hf_style, float gscale, void* stream);

/* ---- object-aware extras (mask-pool / region-sim einsums, BCE, pooled tails) -------------------------
 * oa_model_global_local.py:178,200 ; oa_model_region_mem.py:117,147-151 ; trainer_region_mem.py:97,166.
 * C[b,i,j] (+)= act(sum_k A[b,i,k] Bm[b,k,j]) with arbitrary element strides (fp32). */
int oat_bmm_strided(const float* A, const float* Bm, float* C, int nb, int I, int J, int K, long long sAb,
                    long long sAi, long long sAk, long long sBb, long long sBk, long long sBj, long long sCb,
                    long long sCi, long long sCj, int sigmoid, int accumulate, void* stream);
int oat_sigmoid_bwd(const float* s, const float* ds, float* dz, size_t n, void* stream);
int oat_bce_sum(const float* p, const float* y, size_t n, float* loss, float* partial256, void* stream);
int oat_bce_bwd(const float* p, const float* y, const float* g, float* dp, size_t n, void* stream);
int oat_grouped_broadcast(const float* src, int lds, float* dst, int ldd, int G, int R, int D, float scale,
                          int accumulate, void* stream);
int oat_axpby(const float* a, const float* b, float* out, size_t n, float alpha, float beta, void* stream);
/* tag-token masks built by a Python B x O loop at oa_model_global_local.py:183-196 (ends / n_txt int64) */
int oat_tag_masks(const void* ends, const void* ntxt, float* out, int B, int O, int L, void* stream);
/* bbox -> patch-grid masks built by numpy loops in the reference's datasets (base_dataset_global_local.py:348-356:
 * one mask per box, box_class = sel_class = NULL, O == NB; base_dataset_region_mem.py:233-247: mask o = union of the
 * boxes whose class equals sel_class[b,o]).  bbox fp32 [B,NB,ldb>=4] = x0,y0,x1,y1 in [0,1]; out fp32 [B,O,P*P]. */
int oat_patch_masks(const float* bbox, int ldb, const int* box_class, const int* sel_class, float* out, int B, int NB,
                    int O, int P, void* stream);

#ifdef __cplusplus
}
#endif
#endif
