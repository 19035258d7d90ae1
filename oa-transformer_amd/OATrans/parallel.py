"""Data parallelism for the hot path: one process per GPU, RCCL (torch.distributed 'nccl' on ROCm)
over xGMI.  Two exchange steps exist:

  * embeddings all-gather with autograd (global-batch negatives):
      /root/reference/OATrans/trainer/trainer_dist.py:29-45 (AllGather_multi)
    forward = all_gather in rank order; backward = the LOCAL slice of the incoming gradient, no
    reduction (every rank computes the same global loss; the parameter-gradient mean then yields
    (1/W) dL_global/dtheta).  video and text are packed into ONE collective per step.
  * gradient mean all-reduce (the reference's DistributedDataParallel wrap, base_trainer.py:17-23).
    The HIP engines write gradients into flat per-module buffers, so the all-reduce runs over a few
    large contiguous ranges (big messages suit the per-link-bound xGMI ring), started while backward is
    still running (GradSync.on_ready) so that the exchange hides behind the remaining backward kernels.
"""
import torch
import torch.distributed as dist
from torch import nn


def bwd_nt_grid_default(device=None):
    """Workgroup count of the data-gradient GEMMs while collectives may be in flight (see GradSync.__init__)."""
    import os
    v = os.environ.get("OAT_BWD_NT_GRID", "0xffff")
    if v == "auto":
        if device is None or torch.device(device).type != "cuda":
            device = torch.cuda.current_device()          # (older torch versions reject None here)
        cus = torch.cuda.get_device_properties(device).multi_processor_count      # 256 on an MI355X in SPX mode, less in CPX / DPX
        return max(1, cus - int(os.environ.get("OAT_RCCL_CU_RESERVE", "16")))
    return int(v, 0)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


class AllGather_multi(torch.autograd.Function):
    """Same call signature as the reference: AllGather_multi.apply(tensor, n_gpu, args)."""

    @staticmethod
    def forward(ctx, tensor, n_gpu, args):
        ctx.rank, ctx.batch_size = args.rank, tensor.shape[0]
        if args.world_size == 1:
            return tensor.clone()
        out = torch.empty((args.world_size * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype,
                          device=tensor.device)
        dist.all_gather_into_tensor(out, tensor.contiguous())
        return out

    @staticmethod
    def backward(ctx, grad_output):
        b, r = ctx.batch_size, ctx.rank
        return grad_output[b * r: b * (r + 1)], None, None


def allgather_pair(a, b, args):
    """One collective for both embedding sets: [B, da + db] -> split after the gather.  A single rank has nothing to gather
    (the reference's AllGather_multi is then the identity, trainer_dist.py:29-45): no pack / copy / split launches."""
    if args.world_size == 1:
        return a, b
    packed = AllGather_multi.apply(torch.cat([a, b], dim=1), args.world_size, args)
    return packed[:, :a.shape[1]], packed[:, a.shape[1]:]


def allgather_packed(tensors, args):
    """ONE collective for any number of per-sample tensors [B, ...]: flatten each to [B, -1], concatenate,
    gather with the slice-only backward, split and restore shapes (the reference issues one NCCL call per
    tensor: 6 in trainer_global_local.py:171-182, 4 in trainer_region_mem.py:152-155)."""
    if args.world_size == 1:      # nothing to gather (cf. allgather_pair): no pack / clone / split launches, no slice backward
        return [t if t.dtype == torch.float32 else t.float() for t in tensors]
    flat = [t.reshape(t.shape[0], -1).float() for t in tensors]
    widths = [f.shape[1] for f in flat]
    packed = AllGather_multi.apply(torch.cat(flat, dim=1), args.world_size, args)
    outs, off = [], 0
    for t, w in zip(tensors, widths):
        outs.append(packed[:, off:off + w].reshape((packed.shape[0],) + tuple(t.shape[1:])))
        off += w
    return outs


class GradSync:
    """Mean all-reduce of parameter gradients over flat buffers.

    overlap=True: the engine modules announce gradient ranges as soon as the kernels writing them are
    enqueued (`grad_ready_hook`, one ViT block = 28 MB at a time, the whole text tower at once); each range
    is all-reduced asynchronously on RCCL's stream while the rest of backward runs.  xGMI is point-to-point,
    so at 2 GPUs one link carries the whole 600 MB exchange: hidden behind backward it costs nothing, after
    backward it would cost ~10 ms.  `all_reduce()` (after backward) reduces whatever was not announced,
    waits for the asynchronous work and applies the 1/W mean.  Every rank issues the same collectives in the
    same order (the launch schedule is deterministic).

    grad_dtype=torch.bfloat16 (or OAT_GRAD_DTYPE=bf16): the exchange carries bf16 - half the bytes on the per-link-bound
    xGMI ring.  Each range is cast into a staging buffer, summed by RCCL in bf16 and written back to the fp32 gradient.
    Error: one rounding at the cast (2^-9 relative per element) plus one per ring addition, so |error| <=
    (W + 1) * 2^-9 * max|partial sum| per element; the fp32 default is exact up to summation order (what DDP gives)."""

    def __init__(self, model, bucket_mb=256, overlap=True, force=False, grad_dtype=None, bucket_elems=None):
        import os
        self.model = model
        self.force = force           # issue the collectives even in a 1-rank group (exercises the RCCL path on one GPU)
        self.bucket = int(bucket_elems) if bucket_elems else int(bucket_mb * (1 << 20) // 4)
        if grad_dtype is None:
            grad_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[os.environ.get("OAT_GRAD_DTYPE", "fp32")]
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("grad_dtype: torch.float32 or torch.bfloat16")
        self.grad_dtype = grad_dtype
        self.started_last_step = 0
        self.bwd_nt_grid = 0         # what the engines' backward GEMMs were told (0 = untouched)
        self._pending = []           # async work handles of this step
        self._covered = []           # [byte_lo, byte_hi) address ranges already handed to RCCL this step
        W, _ = world()
        if (W > 1 or force) and torch.cuda.is_available():
            # RCCL's kernels hold CUs for the length of a collective.  A persistent GEMM launch (one workgroup per
            # CU walking its tiles) that finds some CUs taken runs its remaining workgroups AFTER the others - up to
            # twice the time.  The collectives of a step all start inside backward and are waited for before the
            # optimiser step, so only BACKWARD makes room (VideoEngine.bwd_nt_grid); forward keeps one workgroup per CU.
            # Default at W > 1: one workgroup per tile (0xffff), which adapts to whatever CUs RCCL leaves free; it costs
            # +0.55 ms per step on one MI355X with no collective in flight.  Persistent grids that
            # leave a FIXED reserve to RCCL are opt-in until a multi-GPU run has shown how many CUs RCCL holds:
            # OAT_BWD_NT_GRID=auto = the device's CU count minus OAT_RCCL_CU_RESERVE (default 16; the launcher clamps a
            # grid to the tile count), or an explicit number of workgroups.
            self.bwd_nt_grid = bwd_nt_grid_default(next(model.parameters()).device)
            engines = [m._engine for m in model.modules() if hasattr(getattr(m, "_engine", None), "bwd_nt_grid")]
            for eng in engines:
                eng.bwd_nt_grid = self.bwd_nt_grid
            # A model without an engine that exposes bwd_nt_grid gets no backward grid: the library keeps no process-wide GEMM
            # grid any more (round 6: the grid is a per-call argument of oat_gemm_nt), so there is nothing left to fall back to.
            # Known cost of the default, to revisit once a multi-GPU run has shown how many CUs RCCL holds: +0.55 ms per step
            # against OAT_BWD_NT_GRID=auto.
        if overlap and (W > 1 or force):     # a single rank leaves the announcements to the eager optimiser (optim.AdamW.attach)
            for m in model.modules():
                if hasattr(m, "flat_grad") and hasattr(m, "_engine_params"):
                    m.grad_ready_hook = self.on_ready

    def on_ready(self, module, lo, hi):
        """flat_grad()[lo:hi] of `module` is final once the work enqueued so far on the CURRENT stream is done."""
        W, _ = world()
        if (W == 1 and not self.force) or hi <= lo:
            return
        flat = module.flat_grad()[lo:hi]
        self._pending.append(self._start(flat))
        self._covered.append((flat.data_ptr(), flat.data_ptr() + 4 * flat.numel()))

    def _start(self, flat):
        """Asynchronous sum of one range; returns (work, staging buffer or None, range)."""
        if self.grad_dtype == torch.float32:
            return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), None, flat
        stage = flat.to(self.grad_dtype)
        return dist.all_reduce(stage, op=dist.ReduceOp.SUM, async_op=True), stage, flat

    def ranges(self):
        """Maximal contiguous gradient ranges (engine modules expose one each; loose params singly)."""
        out, cur = [], None
        for p in self.model.parameters():
            if p.grad is None:
                continue
            ptr, n = p.grad.data_ptr(), p.grad.numel()
            stor = p.grad.untyped_storage().data_ptr()
            if cur is not None and ptr == cur[2] and stor == cur[3]:
                cur[1] += n
                cur[2] += 4 * n
            else:
                cur = [p.grad, n, ptr + 4 * n, stor]
                out.append(cur)
        return [torch.as_strided(g, (n,), (1,)) for g, n, _, _ in out]

    def _uncovered(self, flat):
        """Pieces of `flat` outside every range announced through on_ready this step."""
        a, b = flat.data_ptr(), flat.data_ptr() + 4 * flat.numel()
        pieces, pos = [], a
        for lo, hi in sorted(c for c in self._covered if c[1] > a and c[0] < b):
            if lo > pos:
                pieces.append((pos, lo))
            pos = max(pos, hi)
        if pos < b:
            pieces.append((pos, b))
        return [flat[(lo - a) // 4:(hi - a) // 4] for lo, hi in pieces]

    def all_reduce(self, average=True):
        W, _ = world()
        if W == 1 and not self.force:
            return
        flats = self.ranges()
        rest = [piece for f in flats for piece in self._uncovered(f)]
        self._reduce(rest)               # the few loose ranges: on the current stream (RCCL runs them on its own)
        for work, stage, flat in self._pending:
            work.wait()                      # device-side: the current stream waits for RCCL's stream
            if stage is not None:
                flat.copy_(stage)
        self.started_last_step = len(self._pending)      # asynchronous collectives started from inside backward this step
        self._pending, self._covered = [], []
        if average and W > 1:
            for f in flats:
                f.div_(W)

    def _reduce(self, flats):
        for f in flats:
            for s in range(0, f.numel(), self.bucket):
                piece = f[s:s + self.bucket]
                if self.grad_dtype == torch.float32:
                    dist.all_reduce(piece, op=dist.ReduceOp.SUM)
                else:
                    stage = piece.to(self.grad_dtype)
                    dist.all_reduce(stage, op=dist.ReduceOp.SUM)
                    piece.copy_(stage)


class HipDataParallel(nn.Module):
    """Stands where the reference puts DistributedDataParallel (base_trainer.py:20-23): exposes
    `.module`, forwards calls, and owns the gradient synchronisation (torch DDP cannot be used: the
    engines write .grad in place, so its autograd hooks never fire)."""

    def __init__(self, module):
        super().__init__()
        self.module = module
        self.sync = GradSync(module)
        self.broadcast_parameters()

    def broadcast_parameters(self, src=0):
        """Every rank starts from rank `src`'s parameters and buffers, as DistributedDataParallel does at construction
        (base_trainer.py:20-23 of the reference): randomly initialised projections / temporal embeddings - or whole
        towers when no pretrained files exist - would otherwise differ per rank and gradient averaging would never
        bring the replicas together.  Also called after a resume.  Flat engine buffers go out as one message each."""
        W, _ = world()
        if W == 1:
            return
        done = set()
        for m in self.module.modules():
            flat = getattr(m, "_flat_param", None)
            if flat is not None:
                dist.broadcast(flat, src)
                done.update(id(p) for _, p in m._engine_params())
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            if id(t) not in done:
                dist.broadcast(t.data, src)
        try:
            from .engine.module import bump_weights_epoch
        except ImportError:
            from engine.module import bump_weights_epoch
        bump_weights_epoch()             # parameters changed through .data: the engines re-cast their bf16 shadows

    def forward(self, *a, **kw):
        return self.module(*a, **kw)

    def backward(self, loss):
        """loss.backward() with the 1/W of the gradient mean applied to the loss instead of to the reduced gradients:
        the all-reduce SUM then IS the mean (bit-identical for W a power of two) and the extra read-modify-write pass
        over all 180.9 M gradients after the collective (DDP's convention, base_trainer.py:20-23) disappears."""
        W, _ = world()
        self._prescaled = W > 1
        (loss * (1.0 / W) if W > 1 else loss).backward()

    def sync_gradients(self):
        self.sync.all_reduce(average=not getattr(self, "_prescaled", False))
        self._prescaled = False
