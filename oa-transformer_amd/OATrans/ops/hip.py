"""ctypes binding of liboatrans_hip.so (include/oatrans_hip.h).

The library is the product path: if it is missing or a symbol is absent this
module raises - there is no CPU / eager fallback (a silent fallback would void
every parity claim).  ``import torch`` must happen before the library is loaded
so that it binds to the HIP runtime torch already mapped (same soname).
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(os.path.dirname(_HERE))            # oa-transformer_amd/
LIB_PATH = os.environ.get("OAT_LIB") or os.path.join(_PKG_ROOT, "liboatrans_hip.so")     # OAT_LIB: dev A/B of two builds
HEADER_PATH = os.path.join(os.path.dirname(_PKG_ROOT), "include", "oatrans_hip.h")

_lib = None


class OatError(RuntimeError):
    pass


def declared_symbols():
    """Every function name include/oatrans_hip.h declares (used by the CPU-side ABI test)."""
    with open(HEADER_PATH) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(oat_[a-z0-9_]+)\s*\(", text)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OatError(f"{LIB_PATH} not built - run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no fallback path")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.oat_last_error.restype = ctypes.c_char_p
        _lib.oat_gemm_tn_workspace_bytes.restype = ctypes.c_size_t
        _lib.oat_tn_group_slab_bytes.restype = ctypes.c_size_t
        _lib.oat_infonce_workspace_floats.restype = ctypes.c_size_t
        _lib.oat_sim_workspace_floats.restype = ctypes.c_size_t
        for name in declared_symbols():
            if not hasattr(_lib, name):
                raise OatError(f"liboatrans_hip.so lacks symbol {name}")
    return _lib


def _ptr(t):
    if t is None:
        return ctypes.c_void_p(0)
    if isinstance(t, torch.Tensor):
        return ctypes.c_void_p(t.data_ptr())
    raise TypeError(type(t))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(rc, what):
    if rc != 0:
        raise OatError(f"{what} failed ({rc}): {lib().oat_last_error().decode()}")


def _f(x):
    return ctypes.c_float(float(x))


EPI_BF16, EPI_F32, EPI_GELU_DUAL, EPI_DGELU, EPI_F32_BF16, EPI_GELU_GRAD, EPI_MUL_AUX = 0, 1, 2, 3, 4, 5, 6
EPI_U8 = 0x100      # or-ed into EPI_GELU_GRAD / EPI_MUL_AUX: the saved GELU derivative is 8-bit fixed point (1 byte per element)


# Launch policy of oat_gemm_nt / oat_gemm_tn is PER CALL (the `tune` / `grid` arguments; the library keeps no tuning state):
GEMM_AUTO, GEMM_128, GEMM_LOCKSTEP, GEMM_PINGPONG = 0, 1, 2, 4        # tune bits 0-7: kernel choice
GRID_PER_TILE = 0xffff                                                # grid: one workgroup per tile (0 = one per CU, else the count)


def gemm_tune(kernel=GEMM_AUTO, m224=None, band=None):
    """`tune` word of gemm_nt.  m224: None auto / 0 never / 2 always 224-row tiles of the ping-pong kernel; band: None auto /
    0 off / n column tiles per group of its band-grouped tile walk.  Every choice is bit-identical (tests/test_kernels_gpu.py)."""
    t = int(kernel)
    if m224 is not None:
        t |= {0: 1, 1: 0, 2: 2}[int(m224)] << 16
    if band is not None:
        t |= (0 if int(band) < 0 else int(band) + 1) << 18
    return t


def hu8_unblock(d8, N):
    """Row-major [rows, N] copy of the BLOCKED 8-bit GELU-derivative tensor that EPI_GELU_GRAD | EPI_U8 writes and
    EPI_MUL_AUX | EPI_U8 reads ([row / 16][col / 64][lane = 16 fk + frow][r][j] -> row 16 G + 4 fk + r, column 64 C + 4 frow + j;
    csrc/gemm_nt_pp.hip).  Tests only: the tensor is private to that pair of launches."""
    rows = d8.numel() // N // 16 * 16
    t = d8.reshape(-1)[:rows * N].view(rows // 16, N // 64, 4, 16, 4, 4)          # [G, C, fk, frow, r, j]
    return t.permute(0, 2, 4, 1, 3, 5).reshape(rows, N)


def gemm_nt(A, B, M, N, K, epi, out, out2=None, bias=None, resid=None, resid_mod=0, aux=None,
            lda=None, ldb=None, ldc=None, ld2=None, ldr=None, ldaux=None, tune=0, grid=0):
    """out[M,N] = A[M,K] @ B[N,K]^T (+epilogue).  Tensors may hold more rows than M.  tune / grid: per-call launch policy
    (gemm_tune, GRID_PER_TILE; 0 = the shipped choice)."""
    rc = lib().oat_gemm_nt(_ptr(A), _ptr(B), M, N, K, lda or A.stride(0), ldb or B.stride(0), epi,
                           _ptr(out), ldc or out.stride(0), _ptr(out2),
                           (ld2 or (out2.stride(0) if out2 is not None else 0)), _ptr(bias), _ptr(resid),
                           (ldr or (resid.stride(0) if resid is not None else 0)), resid_mod, _ptr(aux),
                           (ldaux or (aux.stride(0) if aux is not None else 0)), int(tune), int(grid), _stream())
    _check(rc, "oat_gemm_nt")


_tn_ws = {}


def _stream_key(device):
    """Workspaces are per (device, stream): kernels of different HIP streams may run concurrently."""
    return (str(device), torch.cuda.current_stream().cuda_stream)


def tn_workspace(device, M, N1, N2, tune=0):
    need = lib().oat_gemm_tn_workspace_bytes(M, N1, N2, int(tune))
    key = _stream_key(device)
    ws = _tn_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        if ws is not None:
            _retired.append(ws)          # recorded launch tapes may still point at it: outgrown workspaces are never freed
        ws = torch.empty(need // 4, dtype=torch.float32, device=device)
        _tn_ws[key] = ws
    return ws


def gemm_tn(P, Q, M, N1, N2, out, accumulate=False, ldp=None, ldq=None, bias_out=None, ws=None, tune=0):
    """out[N1,N2] (+)= P[:M,:N1]^T @ Q[:M,:N2]  (weight gradient); bias_out[N1] (+)= colsum(P).
    `ws`: caller-owned fp32 slab workspace (a launch stream needs its own; default = a per-device one
    for the current stream).  tune: tile choice of this call (GEMM_AUTO / GEMM_128 / GEMM_LOCKSTEP / GEMM_PINGPONG)."""
    if ws is None:
        ws = tn_workspace(out.device, M, N1, N2, tune)
    rc = lib().oat_gemm_tn(_ptr(P), _ptr(Q), M, N1, N2, ldp or P.stride(0), ldq or Q.stride(0), _ptr(out),
                           _ptr(bias_out), int(accumulate), _ptr(ws), ctypes.c_size_t(ws.numel() * 4), int(tune), _stream())
    _check(rc, "oat_gemm_tn")


def layernorm_fwd(x, gamma, beta, M, D, eps, y=None, y32=None, mean=None, rstd=None):
    rc = lib().oat_layernorm_fwd(_ptr(x), x.stride(0), _ptr(gamma), _ptr(beta), _ptr(y),
                                 y.stride(0) if y is not None else 0, _ptr(y32),
                                 y32.stride(0) if y32 is not None else 0, _ptr(mean), _ptr(rstd), M, D, _f(eps),
                                 _stream())
    _check(rc, "oat_layernorm_fwd")


def add_layernorm_fwd(x, add16, sum32, gamma, beta, M, D, eps, y=None, y32=None, mean=None, rstd=None):
    """sum32 = x + add16 ; y = LN(sum32)  (fused residual add + LayerNorm)."""
    rc = lib().oat_add_layernorm_fwd(_ptr(x), x.stride(0), _ptr(add16), add16.stride(0), _ptr(sum32),
                                     sum32.stride(0) if sum32 is not None else 0, _ptr(gamma), _ptr(beta), _ptr(y),
                                     y.stride(0) if y is not None else 0, _ptr(y32),
                                     y32.stride(0) if y32 is not None else 0, _ptr(mean), _ptr(rstd), M, D, _f(eps),
                                     _stream())
    _check(rc, "oat_add_layernorm_fwd")


def add2_layernorm_fwd(x, add16, add16b, sum32, gamma, beta, M, D, eps, y=None, y32=None, mean=None, rstd=None):
    """sum32 = x + add16 + add16b ; y = LN(sum32)  (two bf16 branch outputs added in one pass)."""
    s0 = lambda t: t.stride(0) if t is not None else 0
    rc = lib().oat_add2_layernorm_fwd(_ptr(x), x.stride(0), _ptr(add16), add16.stride(0), _ptr(add16b), add16b.stride(0),
                                      _ptr(sum32), s0(sum32), _ptr(gamma), _ptr(beta), _ptr(y), s0(y), _ptr(y32), s0(y32),
                                      _ptr(mean), _ptr(rstd), M, D, _f(eps), _stream())
    _check(rc, "oat_add2_layernorm_fwd")


def add32_layernorm_fwd(x, add32, sum32, gamma, beta, M, D, eps, y=None, y32=None, mean=None, rstd=None):
    """sum32 = x + add32 (fp32) ; y / y32 = LN(sum32)."""
    rc = lib().oat_add32_layernorm_fwd(_ptr(x), x.stride(0), _ptr(add32), add32.stride(0), _ptr(sum32),
                                       sum32.stride(0) if sum32 is not None else 0, _ptr(gamma), _ptr(beta), _ptr(y),
                                       y.stride(0) if y is not None else 0, _ptr(y32),
                                       y32.stride(0) if y32 is not None else 0, _ptr(mean), _ptr(rstd), M, D, _f(eps),
                                       _stream())
    _check(rc, "oat_add32_layernorm_fwd")


def layernorm_fwd_r16(x, M, D, eps, add_a=None, add_b=None, sum16=None, gamma=None, beta=None, y=None, y32=None, mean=None,
                      rstd=None, y8=None, qscale=None, amax=None):
    """LayerNorm on the bf16 residual stream: s = x + add_a + add_b ; sum16 = bf16(s) ; y / y32 = LN(s).  x bf16 or fp32.
    y8 (with qscale, amax): additionally the e4m3 copy of y for an fp8 GEMM (oat_layernorm_fwd_r16_f8)."""
    s0 = lambda t: t.stride(0) if t is not None else 0
    if y8 is not None:
        if y32 is not None or y is None:
            raise OatError("layernorm_fwd_r16: the fp8 form writes y (bf16) and y8, not y32")
        _check(lib().oat_layernorm_fwd_r16_f8(_ptr(x), int(x.dtype == torch.float32), x.stride(0), _ptr(add_a), s0(add_a), _ptr(add_b),
                                              s0(add_b), _ptr(sum16), s0(sum16), _ptr(gamma), _ptr(beta), _ptr(y), s0(y), _ptr(y8),
                                              y8.stride(0), _ptr(qscale), _ptr(amax), _ptr(mean), _ptr(rstd), M, D, _f(eps), _stream()),
               "oat_layernorm_fwd_r16_f8")
        return
    _check(lib().oat_layernorm_fwd_r16(_ptr(x), int(x.dtype == torch.float32), x.stride(0), _ptr(add_a), s0(add_a), _ptr(add_b),
                                       s0(add_b), _ptr(sum16), s0(sum16), _ptr(gamma), _ptr(beta), _ptr(y), s0(y), _ptr(y32),
                                       s0(y32), _ptr(mean), _ptr(rstd), M, D, _f(eps), _stream()), "oat_layernorm_fwd_r16")


_part_ws = {}
_retired = []


def _partials(device, n):
    key = _stream_key(device)
    ws = _part_ws.get(key)
    if ws is None or ws.numel() < n:
        if ws is not None:
            _retired.append(ws)
        ws = torch.empty(n, dtype=torch.float32, device=device)
        _part_ws[key] = ws
    return ws


def layernorm_bwd(dy, x, mean, rstd, gamma, M, D, dx=None, dx16=None, dres=None, dgamma=None, dbeta=None,
                  accumulate=False, dx16_excl_res=False):
    part = None
    if dgamma is not None or dbeta is not None:
        part = _partials(x.device, lib().oat_ln_bwd_blocks(M) * 2 * D)
    rc = lib().oat_layernorm_bwd(_ptr(dy), int(dy.dtype == torch.bfloat16), dy.stride(0), _ptr(x), x.stride(0),
                                 _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dres),
                                 dres.stride(0) if dres is not None else 0, _ptr(dx),
                                 dx.stride(0) if dx is not None else 0, _ptr(dx16),
                                 dx16.stride(0) if dx16 is not None else 0, int(dx16_excl_res), _ptr(dgamma),
                                 _ptr(dbeta),
                                 int(accumulate), _ptr(part), M, D, _stream())
    _check(rc, "oat_layernorm_bwd")


def layernorm_bwd_r16(dy, x16, mean, rstd, gamma, M, D, dx=None, dx16=None, dres16=None, dgamma=None, dbeta=None,
                      accumulate=False):
    """LayerNorm backward with the forward input as bf16 and a bf16 residual-gradient addend (may be dx16: in place)."""
    part = None
    if dgamma is not None or dbeta is not None:
        part = _partials(x16.device, lib().oat_ln_bwd_blocks(M) * 2 * D)
    s0 = lambda t: t.stride(0) if t is not None else 0
    _check(lib().oat_layernorm_bwd_r16(_ptr(dy), int(dy.dtype == torch.bfloat16), dy.stride(0), _ptr(x16), x16.stride(0),
                                       _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dres16), s0(dres16), _ptr(dx), s0(dx),
                                       _ptr(dx16), s0(dx16), _ptr(dgamma), _ptr(dbeta), int(accumulate), _ptr(part), M, D,
                                       _stream()), "oat_layernorm_bwd_r16")


def layernorm_bwd_xhat(dxh, xhat, rstd, M, D, dx=None, dx16=None, dres=None, dx16_excl_res=False, add_a=None, add_b=None,
                       dxp16=None):
    """LayerNorm backward, folded form: dxh / xhat bf16 (gradient w.r.t. the normalised row, the saved normalised row).
    dx (fp32) / dx16 (bf16) = result + dres + add_a + add_b (dx16 without them when dx16_excl_res); dxp16 = plain result."""
    s0 = lambda t: t.stride(0) if t is not None else 0
    _check(lib().oat_layernorm_bwd_xhat(_ptr(dxh), dxh.stride(0), _ptr(xhat), xhat.stride(0), _ptr(rstd), _ptr(dres), s0(dres),
                                        _ptr(dx), s0(dx), _ptr(dx16), s0(dx16), int(dx16_excl_res), _ptr(add_a), s0(add_a),
                                        _ptr(add_b), s0(add_b), _ptr(dxp16), s0(dxp16), M, D, _stream()),
           "oat_layernorm_bwd_xhat")


class FoldBiasTable:
    """b' = b + W beta for every folded linear layer of a module in one launch.  entries: (W fp32 [N, K], beta fp32 [K],
    b fp32 [N] | None, out fp32 [N])."""

    def __init__(self, entries):
        rows, owner, blocks = [], [], 0
        for W, beta, b, out in entries:
            N, K = W.shape
            rows.append([W.data_ptr(), beta.data_ptr(), b.data_ptr() if b is not None else 0, out.data_ptr(), N, K, blocks * 4, 0])
            nb = (N + 3) // 4
            owner.append(torch.full((nb,), len(rows) - 1, dtype=torch.int32))
            blocks += nb
        dev = entries[0][0].device
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.owner = torch.cat(owner).to(dev)
        self.blocks = blocks
        self.keep = entries

    def run(self):
        _check(lib().oat_fold_bias_multi(_ptr(self.table), _ptr(self.owner), self.blocks, _stream()), "oat_fold_bias_multi")


class FoldGradTable:
    """dW' / db' -> dW, db, dgamma, dbeta of folded LayerNorm + linear pairs (oat_ln_fold_grads).  entries: (dWp, dbp, W,
    gamma, beta, dW, db, dgamma, dbeta, accumulate) - fp32 tensors; dW may be dWp and db may be dbp (in place)."""

    def __init__(self, entries):
        rows, blocks = [], 0
        for dWp, dbp, W, gamma, beta, dW, db, dgamma, dbeta, acc in entries:
            N, K = W.shape
            rows.append([dWp.data_ptr(), dbp.data_ptr(), W.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dW.data_ptr(),
                         db.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), N, K, blocks, 1 if acc else 0])
            if K % 4:
                raise OatError("FoldGradTable: K must be a multiple of 4")
            blocks += lib().oat_ln_fold_blocks(K)
        dev = entries[0][0].device
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.n, self.blocks = len(rows), blocks
        self.key = tuple(tuple(r) for r in rows)
        self.keep = entries
        # partial (dgamma, dbeta) sums per block, then the ticket counters (zero now; every launch leaves them zero)
        self.work = torch.zeros(blocks * 128 + blocks, dtype=torch.float32, device=dev)

    def run(self):
        _check(lib().oat_ln_fold_grads(_ptr(self.table), self.n, self.blocks, _ptr(self.work), _stream()), "oat_ln_fold_grads")


def colsum(A, M, N, out, accumulate=False):
    part = _partials(A.device, lib().oat_colsum_rows(M) * N)
    rc = lib().oat_colsum(_ptr(A), int(A.dtype == torch.bfloat16), A.stride(0), M, N, _ptr(out), int(accumulate),
                          _ptr(part), _stream())
    _check(rc, "oat_colsum")


def periodic_rowsum(x, R, P, D, out, accumulate=False):
    _check(lib().oat_periodic_rowsum(_ptr(x), x.stride(0), R, P, D, _ptr(out), int(accumulate), _stream()),
           "oat_periodic_rowsum")


def grouped_rowsum(x, G, R, D, out, accumulate=False, ld=None):
    _check(lib().oat_grouped_rowsum(_ptr(x), ld or x.stride(0), G, R, D, _ptr(out), int(accumulate), _stream()),
           "oat_grouped_rowsum")


def im2col(video, A, BT, C, R, ps):
    _check(lib().oat_im2col(_ptr(video), int(video.dtype == torch.bfloat16), _ptr(A), BT, C, R, ps, A.stride(0),
                            _stream()), "oat_im2col")


def pos_table(pos, temporal, cls_token, table, cls0, T, N, D):
    _check(lib().oat_pos_table(_ptr(pos), _ptr(temporal), _ptr(cls_token), _ptr(table), _ptr(cls0), T, N, D,
                               _stream()), "oat_pos_table")


def broadcast_rows(src, dst, R, D):
    _check(lib().oat_broadcast_rows(_ptr(src), _ptr(dst), dst.stride(0), R, D, _stream()), "oat_broadcast_rows")


def cast_bf16(src, dst=None, dstT=None):
    R, C = src.shape
    _check(lib().oat_cast_bf16(_ptr(src), _ptr(dst), _ptr(dstT), R, C, _stream()), "oat_cast_bf16")


class CastTable:
    """Device descriptor table for oat_cast_bf16_multi, built once per set of (master, shadow) buffers.
    entries: (src fp32 [R, C] contiguous, dst bf16 | None, dstT bf16 | None, ldd, ldT[, colscale fp32 [C] | None]) - dst /
    dstT may be row or column slices of larger shadows (ldd / ldT = their leading dimensions)."""

    def __init__(self, entries):
        rows, tiles, T, owner = [], 0, lib().oat_cast_bf16_tile(), []
        for ent in entries:
            src, dst, dstT, ldd, ldT = ent[:5]
            colscale = ent[5] if len(ent) > 5 else None         # fp32 [C]: column scale applied before the cast (folded LayerNorm)
            R, C = src.shape
            rows.append([src.data_ptr(), dst.data_ptr() if dst is not None else 0,
                         dstT.data_ptr() if dstT is not None else 0, R, C, ldd, ldT, tiles,
                         colscale.data_ptr() if colscale is not None else 0])
            nt = ((R + T - 1) // T) * ((C + T - 1) // T)
            owner.append(torch.full((nt,), len(rows) - 1, dtype=torch.int32))
            tiles += nt
        self.owner = torch.cat(owner).to(entries[0][0].device)     # tile -> matrix (spares the kernel a binary search)
        self.n, self.tiles = len(rows), tiles
        self.key = tuple(r[0] for r in rows)
        self.table = torch.tensor(rows, dtype=torch.int64).to(entries[0][0].device)
        self.keep = entries                      # the table holds raw pointers: keep the tensors alive

    def run(self):
        _check(lib().oat_cast_bf16_multi(_ptr(self.table), self.n, self.tiles, _ptr(self.owner), _stream()), "oat_cast_bf16_multi")


class TnGroup:
    """Several weight gradients  out_p[N1,N2] (+)= P_p[:M]^T @ Q_p[:M],  bias_p[N1] (+)= colsum(P_p)  in ONE persistent
    launch + one fix-up launch (csrc/gemm_tn_sk.hip: the K-tile pairs of all output tiles of all problems form one
    sequence, cut into `grid` equal shares; only the tiles a share boundary cuts leave fp32 partial tiles).
    problems: (P bf16 [>=M, >=N1], Q bf16 [>=M, >=N2], M, N1, N2, out fp32 [N1, N2] contiguous, bias_out fp32 [N1] | None,
    accumulate).  N1, N2 multiples of 256.  The tables hold raw pointers: every tensor is plan-owned and static; `key`
    identifies the set (rebuild when a buffer moved).  The plan itself is computed by the library on the host
    (oat_tn_group_plan, no GPU needed)."""

    REC = 8       # int32 words per segment / fix record

    @staticmethod
    def plan(problems_meta, grid, splits=0):
        """Host-side decomposition: problems_meta = [(M, N1, N2)] -> (segs int32 [nseg, 8], seg_off int32 [blocks + 1],
        fixes int32 [nfix, 8], nslots).  splits = 0: the unit sequence cut into `grid` shares; splits >= 1: every tile split
        that many ways over M, split-major and XCD-contiguous (oat_tn_group_plan).  Segment record: prob, c1, c2, t2, kt0,
        n, slot, last; fix record: prob, c1, c2, t2, slot0, nslots, 0, 0.  CPU-testable."""
        import numpy as np
        n = len(problems_meta)
        tab = np.zeros((n, 8), dtype=np.int64)
        for i, (M, N1, N2) in enumerate(problems_meta):
            tab[i, 4] = (M & 0xffffffff) | (N1 << 32)
            tab[i, 5] = (N2 & 0xffffffff) | (8 << 32)        # ldp (unused by the planner beyond its % 8 check)
            tab[i, 6] = 8                                      # ldq | accumulate << 32
        tiles = sum((N1 // 256) * (N2 // 256) for _, N1, N2 in problems_meta)
        cap = max(tiles + 2 * grid + 8, tiles * max(splits, 1) + 8)
        segs = np.zeros((cap, TnGroup.REC), dtype=np.int32)
        seg_off = np.zeros(cap + 1, dtype=np.int32)
        fcap = max(grid, tiles) + 8
        fixes = np.zeros((fcap, TnGroup.REC), dtype=np.int32)
        counts = np.zeros(4, dtype=np.int32)
        rc = lib().oat_tn_group_plan(tab.ctypes.data_as(ctypes.c_void_p), n, grid, int(splits), segs.ctypes.data_as(ctypes.c_void_p), cap,
                                     seg_off.ctypes.data_as(ctypes.c_void_p), fixes.ctypes.data_as(ctypes.c_void_p), fcap,
                                     counts.ctypes.data_as(ctypes.c_void_p))
        _check(rc, "oat_tn_group_plan")
        return segs[:counts[0]].copy(), seg_off[:counts[3] + 1].copy(), fixes[:counts[1]].copy(), int(counts[2])

    @staticmethod
    def auto_splits(problems_meta, grid):
        """Uniform splits where the problems are big and share one M: the largest split count whose tiles x splits
        workgroups fit one round of the CUs; 0 (stream mode) for many small tiles."""
        tiles = sum((N1 // 256) * (N2 // 256) for _, N1, N2 in problems_meta)
        same_m = len({M for M, _, _ in problems_meta}) == 1
        if not same_m or tiles > grid:
            return 0
        return max(1, grid // tiles)

    @staticmethod
    def plan_layers(layers, grid):
        """Several uniform-split groups STACKED in one launch: `layers` = [[(M, N1, N2), ...], ...]; every layer is planned
        on its own (largest split count that fits `grid` workgroups) and workgroup b walks its segment of layer 0, then of
        layer 1, ...  ViT-B: layer 0 = {fc2, fc1, qkv, qkv} (126 tiles x 2 splits = 252 workgroups), layer 1 = {proj, proj}
        (18 tiles x 14 splits = 252): one GEMM launch + one fix-up per block instead of two each, every workgroup equally
        loaded.  Problem indices, slab slots and fix records of later layers are offset behind the earlier ones'.
        -> (segs, seg_off, fixes, nslots)"""
        import numpy as np
        parts, pbase, sbase = [], 0, 0
        for meta in layers:
            segs, off, fixes, nslots = TnGroup.plan(meta, grid, TnGroup.auto_splits(meta, grid) or 1)
            segs, fixes = segs.copy(), fixes.copy()
            segs[:, 0] += pbase
            segs[:, 6] = np.where(segs[:, 6] >= 0, segs[:, 6] + sbase, -1)
            if len(fixes):
                fixes[:, 0] += pbase
                fixes[:, 4] += sbase
            parts.append((segs, off, fixes))
            pbase += len(meta)
            sbase += nslots
        blocks = max(len(off) - 1 for _, off, _ in parts)
        out, seg_off = [], [0]
        for b in range(blocks):
            for segs, off, _ in parts:
                if b < len(off) - 1:
                    out.append(segs[off[b]:off[b + 1]])
            seg_off.append(sum(len(x) for x in out))
        fixes = [f for _, _, f in parts if len(f)]
        return (np.concatenate(out), np.asarray(seg_off, dtype=np.int32),
                np.concatenate(fixes) if fixes else np.zeros((0, TnGroup.REC), dtype=np.int32), sbase)

    def __init__(self, problems, grid=None, splits=None, slabs=None, layers=None):
        """slabs: optional caller-owned fp32 workspace shared by several groups that never run concurrently (grown by the
        caller; must hold oat_tn_group_slab_bytes(self.nslots) bytes - see `slab_floats`).
        layers: optional partition of `problems` (list of index lists, together covering range(len(problems)) in order):
        uniform-split groups stacked in one launch (plan_layers)."""
        import numpy as np
        dev = problems[0][0].device
        if grid is None:
            grid = torch.cuda.get_device_properties(dev).multi_processor_count
        rows, meta, keep = [], [], []
        for P, Q, M, N1, N2, out, bias_out, acc in problems:
            if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != N1 * N2:
                raise OatError("TnGroup: out must be a contiguous fp32 [N1, N2] tensor")
            if P.dtype != torch.bfloat16 or Q.dtype != torch.bfloat16 or P.stride(1) != 1 or Q.stride(1) != 1:
                raise OatError("TnGroup: P and Q must be bf16 with unit column stride")
            # the grouped kernel fetches whole 64-row K-tiles (the rows of a ragged last tile past M - 1 are zeroed in LDS after they
            # land): the operands must OWN those rows.  Engine buffers are [Mp, .] with Mp = M rounded up to 256 and are read from row 0;
            # a row slice that ends before round_up(M, 64) would be read past its end (oat_gemm_tn clamps instead: use it for such slices)
            need = (M + 63) // 64 * 64
            if P.shape[0] < need or Q.shape[0] < need:
                raise OatError(f"TnGroup: P / Q must hold round_up(M, 64) = {need} readable rows (got {P.shape[0]}, {Q.shape[0]})")
            rows.append([P.data_ptr(), Q.data_ptr(), out.data_ptr(), bias_out.data_ptr() if bias_out is not None else 0,
                         (M & 0xffffffff) | (N1 << 32), (N2 & 0xffffffff) | (P.stride(0) << 32),
                         (Q.stride(0) & 0xffffffff) | ((1 if acc else 0) << 32), 0])
            meta.append((M, N1, N2))
            keep.append((P, Q, out, bias_out))
        if layers is not None:
            if [k for ks in layers for k in ks] != list(range(len(meta))):
                raise OatError("TnGroup: layers must partition the problems in order")
            segs, seg_off, fixes, nslots = self.plan_layers([[meta[k] for k in ks] for ks in layers], grid)
            splits = -1
        else:
            if splits is None:
                splits = self.auto_splits(meta, grid)
            segs, seg_off, fixes, nslots = self.plan(meta, grid, splits)
        self.splits = splits
        self.grid, self.nfix, self.nslots, self.n = len(seg_off) - 1, len(fixes), nslots, len(rows)
        self.key = tuple(tuple(r[:4]) + (r[6] >> 32,) for r in rows)
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.segs = torch.from_numpy(segs).to(dev)
        self.seg_off = torch.from_numpy(seg_off).to(dev)
        self.fixes = torch.from_numpy(fixes if len(fixes) else np.zeros((1, self.REC), dtype=np.int32)).to(dev)
        self.slab_floats = lib().oat_tn_group_slab_bytes(nslots) // 4
        if slabs is not None and slabs.numel() < self.slab_floats:
            raise OatError("TnGroup: the shared slab workspace is too small")
        self.slabs = slabs if slabs is not None else torch.empty(self.slab_floats, dtype=torch.float32, device=dev)
        self.keep = keep

    def run(self):
        _check(lib().oat_tn_group_run(_ptr(self.table), _ptr(self.segs), _ptr(self.seg_off), self.grid, _ptr(self.fixes), self.nfix,
                                      _ptr(self.slabs), _stream()), "oat_tn_group_run")


def _attn_fwd(fn, name, qkv, out, lse, B, T, N, H, D, scale):
    _check(fn(_ptr(qkv), qkv.stride(0), _ptr(out), out.stride(0), _ptr(lse), B, T, N, H, D, _f(scale), _stream()),
           name)


def attn_space_fwd(qkv, out, lse, B, T, N, H, D, scale):
    _attn_fwd(lib().oat_attn_space_fwd, "oat_attn_space_fwd", qkv, out, lse, B, T, N, H, D, scale)


def attn_time_fwd(qkv, out, lse, B, T, N, H, D, scale):
    _attn_fwd(lib().oat_attn_time_fwd, "oat_attn_time_fwd", qkv, out, lse, B, T, N, H, D, scale)


def attn_cls_fwd(qkv, out, lse, B, T, N, H, D, scale):
    _attn_fwd(lib().oat_attn_cls_fwd, "oat_attn_cls_fwd", qkv, out, lse, B, T, N, H, D, scale)


def attn_cls_fwd_dual(qkv, out, lse, q32, o32, B, T, N, H, D, scale):
    _check(lib().oat_attn_cls_fwd_dual(_ptr(qkv), qkv.stride(0), _ptr(out), out.stride(0), _ptr(lse), _ptr(q32),
                                       q32.stride(0), _ptr(o32), o32.stride(0), B, T, N, H, D, _f(scale), _stream()),
           "oat_attn_cls_fwd_dual")


def _attn_bwd(fn, name, qkv, out, lse, dout, dqkv, cls_side, B, T, N, H, D, scale):
    _check(fn(_ptr(qkv), qkv.stride(0), _ptr(out), out.stride(0), _ptr(lse), _ptr(dout), dout.stride(0),
              _ptr(dqkv), dqkv.stride(0), _ptr(cls_side), B, T, N, H, D, _f(scale), _stream()), name)


def attn_space_bwd(qkv, out, lse, dout, dqkv, cls_side, B, T, N, H, D, scale):
    _attn_bwd(lib().oat_attn_space_bwd, "oat_attn_space_bwd", qkv, out, lse, dout, dqkv, cls_side, B, T, N, H, D,
              scale)


def attn_time_bwd(qkv, out, lse, dout, dqkv, cls_side, B, T, N, H, D, scale):
    _attn_bwd(lib().oat_attn_time_bwd, "oat_attn_time_bwd", qkv, out, lse, dout, dqkv, cls_side, B, T, N, H, D,
              scale)


def _attn_bwd_fin(fn, name, qkv, out, lse, dout, dqkv, cls_side, done, B, T, N, H, D, scale):
    _check(fn(_ptr(qkv), qkv.stride(0), _ptr(out), out.stride(0), _ptr(lse), _ptr(dout), dout.stride(0),
              _ptr(dqkv), dqkv.stride(0), _ptr(cls_side), _ptr(done), B, T, N, H, D, _f(scale), _stream()), name)


def attn_space_bwd_fin(qkv, out, lse, dout, dqkv, cls_side, done, B, T, N, H, D, scale):
    """attn_space_bwd + attn_cls_finalize in one launch; done: int32 [B, H] tickets, zero on entry and on exit"""
    _attn_bwd_fin(lib().oat_attn_space_bwd_fin, "oat_attn_space_bwd_fin", qkv, out, lse, dout, dqkv, cls_side, done, B, T,
                  N, H, D, scale)


def attn_time_bwd_fin(qkv, out, lse, dout, dqkv, cls_side, done, B, T, N, H, D, scale):
    _attn_bwd_fin(lib().oat_attn_time_bwd_fin, "oat_attn_time_bwd_fin", qkv, out, lse, dout, dqkv, cls_side, done, B, T,
                  N, H, D, scale)


class _AttnClip(ctypes.Structure):
    _fields_ = [("qkv", ctypes.c_void_p), ("out", ctypes.c_void_p), ("lse", ctypes.c_void_p), ("dout", ctypes.c_void_p),
                ("dqkv", ctypes.c_void_p), ("cls_side", ctypes.c_void_p), ("done", ctypes.c_void_p), ("B", ctypes.c_int),
                ("T", ctypes.c_int)]


def _clip_array(clips):
    arr = (_AttnClip * len(clips))()
    for i, c in enumerate(clips):
        for k in ("qkv", "out", "lse", "dout", "dqkv", "cls_side", "done"):
            t = c.get(k)
            setattr(arr[i], k, t.data_ptr() if t is not None else None)
        arr[i].B, arr[i].T = c["B"], c["T"]
    return arr


def attn_space_fwd_clips(clips, N, H, D, scale):
    """clips: list of 1-2 dicts {qkv, out, lse, B, T} (row views of one geometry) -> one launch"""
    q, o = clips[0]["qkv"], clips[0]["out"]
    _check(lib().oat_attn_space_fwd_clips(_clip_array(clips), len(clips), q.stride(0), o.stride(0), N, H, D, _f(scale), _stream()),
           "oat_attn_space_fwd_clips")


def attn_space_bwd_clips(clips, N, H, D, scale, cls_query_only=False):
    """clips: list of 1-2 dicts {qkv, out, lse, dout, dqkv, cls_side, done, B, T}; backward with the fused CLS-row finalize.
    cls_query_only: the caller guarantees dO = 0 and lse = 3.4e38 on every patch query (the pruned top block): the launch skips the
    exact zeros those queries contribute (bit-identical gradients)"""
    c = clips[0]
    _check(lib().oat_attn_space_bwd_clips(_clip_array(clips), len(clips), c["qkv"].stride(0), c["out"].stride(0), c["dout"].stride(0),
                                          c["dqkv"].stride(0), N, H, D, _f(scale), int(bool(cls_query_only)), _stream()), "oat_attn_space_bwd_clips")


def attn_time_bwd_clips(clips, N, H, D, scale):
    """TIME attention backward + fused CLS-row finalize of two clips (frame counts: powers of two <= 16) in one launch"""
    c = clips[0]
    _check(lib().oat_attn_time_bwd_clips(_clip_array(clips), len(clips), c["qkv"].stride(0), c["out"].stride(0), c["dout"].stride(0),
                                         c["dqkv"].stride(0), N, H, D, _f(scale), _stream()), "oat_attn_time_bwd_clips")


def attn_cls_finalize(cls_side, dqkv, B, T, N, H, D):
    _check(lib().oat_attn_cls_finalize(_ptr(cls_side), _ptr(dqkv), dqkv.stride(0), B, T, N, H, D, _stream()),
           "oat_attn_cls_finalize")


# ----------------------------------------------------------------------------- text encoder / loss / optimiser
def embed_fwd(ids, word, pos, out, M, L, D):
    _check(lib().oat_embed_fwd(_ptr(ids), _ptr(word), _ptr(pos), _ptr(out), out.stride(0), M, L, D, _stream()),
           "oat_embed_fwd")


def embed_bwd(ids, g, dword, M, D):
    _check(lib().oat_embed_bwd(_ptr(ids), _ptr(g), g.stride(0), _ptr(dword), M, D, _stream()), "oat_embed_bwd")


def attn_text_fwd(qkv, mask, out, lse, B, L, H, D, scale):
    _check(lib().oat_attn_text_fwd(_ptr(qkv), qkv.stride(0), _ptr(mask), _ptr(out), out.stride(0), _ptr(lse), B, L,
                                   H, D, _f(scale), _stream()), "oat_attn_text_fwd")


def attn_text_fwd_dual(qkv, qkv32, mask, out, out32, lse, B, L, H, D, scale, drop_p=0.0, rng=None, site=0):
    _check(lib().oat_attn_text_fwd_dual(_ptr(qkv), qkv.stride(0), _ptr(qkv32), qkv32.stride(0), _ptr(mask), _ptr(out),
                                        out.stride(0), _ptr(out32), out32.stride(0), _ptr(lse), B, L, H, D, _f(scale),
                                        _f(drop_p), _ptr(rng), ctypes.c_uint(site), _stream()), "oat_attn_text_fwd_dual")


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def frames_resize(frames, out_hw, crop=None, flip=False, mean=IMAGENET_MEAN, std=IMAGENET_STD, out=None, dtype=torch.bfloat16):
    """Decoded frames -> normalised clip: uint8 [F, H, W, 3] (or float [F, 3, H, W]) -> [F, 3, OH, OW].  crop = (x0, y0, w, h)
    in source pixels.  One launch for all frames (see oat_frames_resize)."""
    u8 = frames.dtype == torch.uint8
    frames = frames.contiguous()
    if u8:
        F, H, W, _ = frames.shape
    else:
        frames = frames.float()
        F, _, H, W = frames.shape
    OH, OW = out_hw
    if out is None:
        out = torch.empty(F, 3, OH, OW, dtype=dtype, device=frames.device)
    arr3 = ctypes.c_float * 3
    arr4 = ctypes.c_float * 4
    _check(lib().oat_frames_resize(_ptr(frames), int(u8), F, H, W, arr4(*[float(v) for v in crop]) if crop is not None else None,
                                   int(flip), _ptr(out), int(out.dtype == torch.bfloat16), OH, OW, _f(1.0 / 255.0 if u8 else 1.0),
                                   arr3(*mean) if mean is not None else None, arr3(*std) if std is not None else None, _stream()),
           "oat_frames_resize")
    return out


# ---- launch tape ----------------------------------------------------------------------------------------------------
def tape_begin():
    _check(lib().oat_tape_begin(), "oat_tape_begin")


def tape_abort():
    lib().oat_tape_abort()


def tape_mark():
    seg = lib().oat_tape_mark()
    if seg < 0:
        _check(seg, "oat_tape_mark")
    return seg


def tape_end():
    tid = lib().oat_tape_end()
    if tid < 0:
        _check(tid, "oat_tape_end")
    return tid


def tape_replay(tid, seg_lo=0, seg_hi=-1):
    _check(lib().oat_tape_replay(tid, seg_lo, seg_hi), "oat_tape_replay")


def tape_free(tid):
    lib().oat_tape_free(tid)


_side_streams = {}


def side_stream(prefix, device=None):
    """The stream of a side tower (`prefix` = "text": the text tower; "lane": the CLS lane + CLS-query attention).  (CU-masked side
    streams and stream priorities were measured in round 4 - 58-63 ms per step on 16 / 32 CUs, priorities equal - and are gone.)"""
    # ONE stream per (role, device) for the whole process: HIP multiplexes streams onto a handful of hardware queues
    # (GPU_MAX_HW_QUEUES, 4 by default) and two streams that share a queue run in enqueue order - a second model built in the
    # same process (bench.py's `other_configs`, validation models) used to get fresh streams that landed on the queue of its
    # own main stream: global_local measured 575 pairs/s inside bench.py against 603 alone
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    key = (prefix, dev)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=dev)
    return _side_streams[key]


def stream_edge(src_stream, dst_stream):
    """Everything enqueued on dst_stream from now on waits for what is on src_stream now (torch streams)."""
    _check(lib().oat_stream_edge(ctypes.c_void_p(src_stream.cuda_stream), ctypes.c_void_p(dst_stream.cuda_stream)), "oat_stream_edge")


def zero_(t):
    """t.zero_() as a recordable launch (t contiguous)."""
    assert t.is_contiguous()
    _check(lib().oat_memset_async(_ptr(t), 0, ctypes.c_size_t(t.numel() * t.element_size()), _stream()), "oat_memset_async")


def fill_bytes_(t, byte):
    """every BYTE of t (contiguous) set to `byte`, as a recordable launch (0x7f in an fp32 tensor: 3.39e38 per element)"""
    assert t.is_contiguous() and 0 <= byte <= 255
    _check(lib().oat_memset_async(_ptr(t), int(byte), ctypes.c_size_t(t.numel() * t.element_size()), _stream()), "oat_memset_async")


def copy_(dst, src):
    """dst.copy_(src) for contiguous tensors of one dtype and size, as a recordable launch."""
    assert dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.numel() == src.numel()
    _check(lib().oat_copy_async(_ptr(dst), _ptr(src), ctypes.c_size_t(dst.numel() * dst.element_size()), _stream()), "oat_copy_async")


# ---- fp8 (OCP e4m3fn) forward GEMMs ------------------------------------------------------------------------------
def fp8_quant(x, out8, M, K, qscale, amax=None):
    _check(lib().oat_fp8_quant(_ptr(x), int(x.dtype == torch.bfloat16), x.stride(0), _ptr(out8), out8.stride(0), M, K,
                               _ptr(qscale), _ptr(amax), _stream()), "oat_fp8_quant")


def fp8_amax(x, M, K, amax):
    _check(lib().oat_fp8_amax(_ptr(x), int(x.dtype == torch.bfloat16), x.stride(0), M, K, _ptr(amax), _stream()), "oat_fp8_amax")


def fp8_update_scales(amax, qscale, dq, n, margin=1.0):
    _check(lib().oat_fp8_update_scales(_ptr(amax), _ptr(qscale), _ptr(dq), n, _f(margin), _stream()), "oat_fp8_update_scales")


class Fp8Table:
    """Descriptor table for oat_fp8_multi: (src bf16 contiguous, dst uint8 same numel, site) per matrix."""

    def __init__(self, entries):
        chunk = lib().oat_fp8_chunk_elems()
        rows, owner, blocks = [], [], 0
        for src, dst, site in entries:
            n = src.numel()
            assert n % 8 == 0 and src.is_contiguous() and dst.is_contiguous() and dst.numel() == n
            rows.append([src.data_ptr(), dst.data_ptr(), n, site, blocks])
            nb = (n + chunk - 1) // chunk
            owner.append(torch.full((nb,), len(rows) - 1, dtype=torch.int32))
            blocks += nb
        dev = entries[0][0].device
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.owner = torch.cat(owner).to(dev)
        self.blocks, self.keep = blocks, entries

    def run(self, qscale, amax, quant):
        _check(lib().oat_fp8_multi(_ptr(self.table), _ptr(self.owner), self.blocks, _ptr(qscale), _ptr(amax), int(quant),
                                   _stream()), "oat_fp8_multi")


def gemm_nt_f8(A8, B8, M, N, K, epi, out, dq_a, dq_b, out2=None, bias=None, out8=None, q_out=None, amax_out=None):
    s0 = lambda t: t.stride(0) if t is not None else 0
    _check(lib().oat_gemm_nt_f8(_ptr(A8), _ptr(B8), M, N, K, A8.stride(0), B8.stride(0), int(epi), _ptr(out), out.stride(0),
                                _ptr(out2), s0(out2), _ptr(bias), _ptr(dq_a), _ptr(dq_b),
                                _ptr(out8), s0(out8), _ptr(q_out), _ptr(amax_out), _stream()), "oat_gemm_nt_f8")


def layernorm_fwd_f8(x, gamma, beta, M, D, eps, y, y8, qscale, amax, mean, rstd, add16=None, sum32=None):
    """y = LN(x [+ add16]) as bf16 and as e4m3 (y8); sum32 = x + add16 when add16 is given."""
    s0 = lambda t: t.stride(0) if t is not None else 0
    _check(lib().oat_layernorm_fwd_f8(_ptr(x), x.stride(0), _ptr(add16), s0(add16), _ptr(sum32), s0(sum32), _ptr(gamma),
                                      _ptr(beta), _ptr(y), y.stride(0), _ptr(y8), y8.stride(0), _ptr(qscale), _ptr(amax),
                                      _ptr(mean), _ptr(rstd), M, D, _f(eps), _stream()), "oat_layernorm_fwd_f8")


def new_rng_state(seed, device):
    """Device-resident dropout state {seed, offset} (int64[2]); oat_rng_tick advances the offset."""
    return torch.tensor([int(seed) & 0x7fffffffffffffff, 0], dtype=torch.int64, device=device)


def rng_tick(state):
    _check(lib().oat_rng_tick(_ptr(state), _stream()), "oat_rng_tick")


def dropout(x, M, D, p, rng, site, resid=None, out32=None, out16=None):
    s0 = lambda t: t.stride(0) if t is not None else 0
    _check(lib().oat_dropout(_ptr(x), x.stride(0), _ptr(resid), s0(resid), _ptr(out32), s0(out32), _ptr(out16), s0(out16),
                             M, D, _f(p), _ptr(rng), ctypes.c_uint(site), _stream()), "oat_dropout")


def dropout_mask(n, p, rng, site):
    out = torch.empty(n, dtype=torch.float32, device=rng.device)
    _check(lib().oat_dropout_mask(_ptr(out), ctypes.c_longlong(n), _f(p), _ptr(rng), ctypes.c_uint(site), _stream()),
           "oat_dropout_mask")
    return out


def philox4x32_10(words):
    """words: int64 [n, 6] (4 counter words, 2 key words, each < 2^32) -> int64 [n, 4]."""
    inp = words.to(torch.int64).contiguous()
    n = inp.shape[0]
    i32 = (inp & 0xffffffff).to(torch.int64)
    i32 = torch.where(i32 >= 2 ** 31, i32 - 2 ** 32, i32).to(torch.int32).contiguous()
    out = torch.empty(n, 4, dtype=torch.int32, device=inp.device)
    _check(lib().oat_philox4x32_10(_ptr(i32), _ptr(out), n, _stream()), "oat_philox4x32_10")
    return out.to(torch.int64) & 0xffffffff


LIN_NONE, LIN_GELU, LIN_RELU_IN = 0, 1, 2
LIN_EXACT = 0x100       # or-ed into `act`: exact-f32 MFMA products at every M (default: M > 64 rows run on the split-bf16 kernel, 2^-16 relative)


def linear_f32(A, W, M, N, K, bias=None, out32=None, out16=None, out16b=None, resid=None, act=LIN_NONE, lda=None, ldw=None):
    """out = act(in(A)[M,K] @ W[N,K]^T + bias) (+ resid) on fp32 operands.  M <= 64: exact-f32 MFMA.  M > 64: the split-bf16 three-pass
    kernel (hi + lo, 2^-16 relative per product: the text tower's forward) unless `act` carries LIN_EXACT (the CLS lane, the projection heads)."""
    s0 = lambda t: t.stride(0) if t is not None else 0
    _check(lib().oat_linear_f32(_ptr(A), lda or A.stride(0), _ptr(W), ldw or W.stride(0), _ptr(bias), M, N, K, _ptr(out32),
                                s0(out32), _ptr(out16), s0(out16), _ptr(out16b), s0(out16b), _ptr(resid), s0(resid),
                                int(act), _stream()), "oat_linear_f32")


def linear_small_bwd(x, dy, W, M, N, K, relu_in=False, want_dx=True, want_db=True):
    """(dx, dW, db) of y = act(x) @ W^T + b for M <= 64 rows in ONE launch (fp32, deterministic); None where not wanted."""
    dx = torch.empty(M, K, dtype=torch.float32, device=x.device) if want_dx else None
    dW = torch.empty(N, K, dtype=torch.float32, device=x.device)
    db = torch.empty(N, dtype=torch.float32, device=x.device) if want_db else None
    _check(lib().oat_linear_small_bwd(_ptr(x), x.stride(0), _ptr(dy), dy.stride(0), _ptr(W), W.stride(0), M, N, K, int(relu_in),
                                      _ptr(dx), dx.stride(0) if dx is not None else 0, _ptr(dW), _ptr(db), _stream()), "oat_linear_small_bwd")
    return dx, dW, db


def linear_f32_qkv(A, Wq, Wk, Wv, M, n, K, bq=None, bk=None, bv=None, out32=None, out16=None):
    """[q | k | v] = A[M,K] @ [Wq; Wk; Wv]^T + [bq | bk | bv] in one launch (three separate fp32 [n, K] weights, one [M, 3n] output);
    bit-identical to three linear_f32 calls on the column slices.  n % 128 == 0, K % 32 == 0, M > 64."""
    s0 = lambda t: t.stride(0) if t is not None else 0
    assert Wq.stride(0) == Wk.stride(0) == Wv.stride(0)
    _check(lib().oat_linear_f32_qkv(_ptr(A), A.stride(0), _ptr(Wq), _ptr(Wk), _ptr(Wv), Wq.stride(0), _ptr(bq), _ptr(bk), _ptr(bv),
                                    M, n, K, _ptr(out32), s0(out32), _ptr(out16), s0(out16), _stream()), "oat_linear_f32_qkv")


def attn_text_bwd(qkv, mask, out, lse, delta, dout, dqkv, B, L, H, D, scale, drop_p=0.0, rng=None, site=0):
    _check(lib().oat_attn_text_bwd(_ptr(qkv), qkv.stride(0), _ptr(mask), _ptr(out), out.stride(0), _ptr(lse),
                                   _ptr(delta), _ptr(dout), dout.stride(0), _ptr(dqkv), dqkv.stride(0), B, L, H, D,
                                   _f(scale), _f(drop_p), _ptr(rng), ctypes.c_uint(site), _stream()), "oat_attn_text_bwd")


def relu_bf16(x, y, M, D):
    _check(lib().oat_relu_bf16(_ptr(x), x.stride(0), _ptr(y), y.stride(0), M, D, _stream()), "oat_relu_bf16")


def relu_bwd(x, dy, dx, M, D):
    _check(lib().oat_relu_bwd(_ptr(x), x.stride(0), _ptr(dy), dy.stride(0), _ptr(dx), dx.stride(0), M, D, _stream()),
           "oat_relu_bwd")


_nce_ws = {}


def infonce(t, v, temperature=0.05, eps=1e-8, r0=0, nloc=None, want_sim=False, want_grads=True):
    """Fused sim_matrix + NormSoftmaxLoss fwd/bwd.  Returns (loss[1], sim|None, dt|None, dv|None)."""
    n, d = t.shape
    nloc = n if nloc is None else nloc
    need = lib().oat_infonce_workspace_floats(n, d)
    key = (_stream_key(t.device), need)
    ws = _nce_ws.get(key)
    if ws is None:
        ws = torch.empty(need, dtype=torch.float32, device=t.device)
        _nce_ws[key] = ws
    loss = torch.empty(1, dtype=torch.float32, device=t.device)
    sim = torch.empty(n, n, dtype=torch.float32, device=t.device) if want_sim else None
    dt = torch.empty(nloc, d, dtype=torch.float32, device=t.device) if want_grads else None
    dv = torch.empty(nloc, d, dtype=torch.float32, device=t.device) if want_grads else None
    _check(lib().oat_infonce(_ptr(t), _ptr(v), n, d, _f(temperature), _f(eps), _ptr(loss), _ptr(sim), _ptr(dt),
                             _ptr(dv), r0, nloc, _ptr(ws), _stream()), "oat_infonce")
    return loss, sim, dt, dv


def adamw(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, hf_style=True, gscale=1.0):
    _check(lib().oat_adamw(_ptr(p), _ptr(g), _ptr(m), _ptr(v), ctypes.c_size_t(p.numel()), _f(lr), _f(beta1),
                           _f(beta2), _f(eps), _f(weight_decay), int(step), int(hf_style), _f(gscale), _stream()),
           "oat_adamw")


def adam_tick(step, lr, beta1, beta2, coef):
    _check(lib().oat_adam_tick(_ptr(step), _ptr(lr), _f(beta1), _f(beta2), _ptr(coef), _stream()), "oat_adam_tick")


def adamw_dev(p, g, m, v, coef, beta1, beta2, eps, weight_decay, hf_style=True, gscale=1.0):
    _check(lib().oat_adamw_dev(_ptr(p), _ptr(g), _ptr(m), _ptr(v), ctypes.c_size_t(p.numel()), _ptr(coef), _f(beta1),
                               _f(beta2), _f(eps), _f(weight_decay), int(hf_style), _f(gscale), _stream()),
           "oat_adamw_dev")


def sim_matrix_fwd(t, v, eps=1e-8):
    n, d = t.shape
    m = v.shape[0]
    ws = torch.empty(lib().oat_sim_workspace_floats(n, m, d), dtype=torch.float32, device=t.device)
    sim = torch.empty(n, m, dtype=torch.float32, device=t.device)
    _check(lib().oat_sim_matrix_fwd(_ptr(t), _ptr(v), n, m, d, _f(eps), _ptr(sim), _ptr(ws), _stream()),
           "oat_sim_matrix_fwd")
    return sim, ws


def sim_matrix_bwd(G, ws, n, m, d):
    dt = torch.empty(n, d, dtype=torch.float32, device=G.device)
    dv = torch.empty(m, d, dtype=torch.float32, device=G.device)
    _check(lib().oat_sim_matrix_bwd(_ptr(G), _ptr(ws), n, m, d, _ptr(dt), 0, n, _ptr(dv), 0, m, _stream()),
           "oat_sim_matrix_bwd")
    return dt, dv


def norm_softmax_loss(sim, temperature, want_grad=True):
    n = sim.shape[0]
    loss = torch.empty(1, dtype=torch.float32, device=sim.device)
    G = torch.empty_like(sim) if want_grad else None
    ws = torch.empty(2 * n, dtype=torch.float32, device=sim.device)
    _check(lib().oat_norm_softmax_loss(_ptr(sim), n, _f(temperature), _ptr(loss), _ptr(G), _ptr(ws), _stream()),
           "oat_norm_softmax_loss")
    return loss, G


# ----------------------------------------------------------------------------- object-aware extras
def bmm_strided(A, Bm, C, nb, I, J, K, sA, sB, sC, sigmoid=False, accumulate=False):
    """C[b,i,j] (+)= act(sum_k A[b,i,k] * Bm[b,k,j]); sA=(b,i,k) sB=(b,k,j) sC=(b,i,j) element strides."""
    ll = ctypes.c_longlong
    _check(lib().oat_bmm_strided(_ptr(A), _ptr(Bm), _ptr(C), nb, I, J, K, ll(sA[0]), ll(sA[1]), ll(sA[2]), ll(sB[0]),
                                 ll(sB[1]), ll(sB[2]), ll(sC[0]), ll(sC[1]), ll(sC[2]), int(sigmoid), int(accumulate),
                                 _stream()), "oat_bmm_strided")


def sigmoid_bwd(s, ds, dz):
    _check(lib().oat_sigmoid_bwd(_ptr(s), _ptr(ds), _ptr(dz), ctypes.c_size_t(s.numel()), _stream()), "oat_sigmoid_bwd")


def bce_sum(p, y):
    loss = torch.empty(1, dtype=torch.float32, device=p.device)
    part = torch.empty(256, dtype=torch.float32, device=p.device)
    _check(lib().oat_bce_sum(_ptr(p), _ptr(y), ctypes.c_size_t(p.numel()), _ptr(loss), _ptr(part), _stream()),
           "oat_bce_sum")
    return loss


def bce_bwd(p, y, g, dp):
    _check(lib().oat_bce_bwd(_ptr(p), _ptr(y), _ptr(g), _ptr(dp), ctypes.c_size_t(p.numel()), _stream()), "oat_bce_bwd")


def grouped_broadcast(src, dst, G, R, D, scale=1.0, accumulate=False):
    _check(lib().oat_grouped_broadcast(_ptr(src), src.stride(0), _ptr(dst), dst.stride(0), G, R, D, _f(scale),
                                       int(accumulate), _stream()), "oat_grouped_broadcast")


def axpby(a, b, out, alpha, beta=0.0):
    _check(lib().oat_axpby(_ptr(a), _ptr(b), _ptr(out), ctypes.c_size_t(out.numel()), _f(alpha), _f(beta), _stream()),
           "oat_axpby")


def tag_masks(ends, ntxt, L):
    B, O = ends.shape
    out = torch.empty(B, O, L, dtype=torch.float32, device=ends.device)
    _check(lib().oat_tag_masks(_ptr(ends.contiguous()), _ptr(ntxt.contiguous()), _ptr(out), B, O, L, _stream()),
           "oat_tag_masks")
    return out


def patch_masks(bbox, P=14, box_class=None, sel_class=None):
    """bbox fp32 [B, NB, >=4] -> fp32 [B, O, P*P] patch-grid masks (see oat_patch_masks)."""
    bbox = bbox.float().contiguous()
    B, NB, ldb = bbox.shape
    if box_class is not None:
        box_class, sel_class = box_class.to(torch.int32).contiguous(), sel_class.to(torch.int32).contiguous()
        O = sel_class.shape[1]
    else:
        O = NB
    out = torch.empty(B, O, P * P, dtype=torch.float32, device=bbox.device)
    _check(lib().oat_patch_masks(_ptr(bbox), ldb, _ptr(box_class), _ptr(sel_class), _ptr(out), B, NB, O, P, _stream()),
           "oat_patch_masks")
    return out
