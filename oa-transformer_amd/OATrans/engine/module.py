"""Shared plumbing for nn.Modules whose forward/backward are HIP launch schedules: parameters stay
ordinary fp32 nn.Parameters (reference state_dict names), gradients live in ONE persistent flat fp32
buffer per module whose slices are the Parameters' .grad (kernels write them in place; the flat
buffer is also what the gradient all-reduce and the fused AdamW operate on)."""
import torch
from torch import nn


_WEIGHTS_EPOCH = [0]


def bump_weights_epoch():
    """Called by optimisers that update parameters through raw pointers (no autograd version bump)
    so that the engines re-cast their bf16 weight shadows."""
    _WEIGHTS_EPOCH[0] += 1


class EngineModule(nn.Module):
    _skip_prefixes = ()

    def _new_step_guard(self):
        """Self-healing for the forward / backward call counters (`_fwd_calls`, `_bwd_calls`: which activation plan a
        differentiated forward takes, whether a backward overwrites or accumulates).  `begin_step()` of the owning model
        resets them; a loop that never calls it, or a grad-enabled forward whose output never receives a gradient,
        would otherwise leave `_fwd_calls > _bwd_calls` for good - every later step allocating one more plan and
        ACCUMULATING into the previous step's gradient buffers.  An optimiser step (it bumps the weights epoch) ends a
        step by definition: the first forward after one starts from zero."""
        ep = _WEIGHTS_EPOCH[0]
        if self.__dict__.get("_calls_epoch") != ep:
            self.__dict__["_calls_epoch"] = ep
            if getattr(self, "_fwd_calls", 0) or getattr(self, "_bwd_calls", 0):
                self._fwd_calls = self._bwd_calls = 0
        if getattr(self, "_fwd_calls", 0) >= 64:
            raise RuntimeError(f"{type(self).__name__}: {self._fwd_calls} differentiated forwards without a backward or an "
                               "optimiser step - call model.begin_step() at the start of every step")

    def _weights_signature(self):
        pairs = self._engine_params()
        return (_WEIGHTS_EPOCH[0], sum(p._version for _, p in pairs), pairs[0][1].data_ptr(), pairs[-1][1].data_ptr())

    def flatten_parameters(self):
        """Re-home every engine parameter into ONE flat fp32 buffer (same order as the flat gradient
        buffer) so that the optimiser and the gradient all-reduce each see a single range."""
        pairs = self._engine_params()
        dev = pairs[0][1].device
        total = sum(p.numel() for _, p in pairs)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        off = 0
        for _, p in pairs:
            v = flat[off:off + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            off += p.numel()
        self._flat_param = flat
        self._gradbuf = None
        for _, p in pairs:
            p.grad = None
        self._grad_views()
        return flat

    def _engine_params(self):
        """[(state_dict name, Parameter)] in flat-buffer order.  Called several times per step: the list is cached
        together with where every Parameter hangs (owner module, attribute) and re-validated by identity - walking
        named_parameters() of the ViT costs 0.4 ms a call."""
        cache = self.__dict__.get("_pairs_cache")
        if cache is not None and all(o._parameters.get(a) is p for o, a, p in cache[1]):
            return cache[0]
        names = getattr(self, "_names", None)
        if names is None:
            names = [n for n, _ in self.named_parameters() if not n.startswith(tuple(self._skip_prefixes) or ("\0",))]
            self._names = names
            self._n_params = len(names)
        d = dict(self.named_parameters())
        pairs = [(n, d[n]) for n in names]
        where = []
        for n, prm in pairs:
            owner_path, _, attr = n.rpartition(".")
            where.append((self.get_submodule(owner_path) if owner_path else self, attr, prm))
        self.__dict__["_pairs_cache"] = (pairs, where)
        return pairs

    def _param_data(self):
        return {n: p.data for n, p in self._engine_params()}

    grad_ready_hook = None      # set by parallel.GradSync: hook(module, lo, hi) once flat_grad()[lo:hi] is enqueued

    def _announce(self, prefixes):
        """Tell the gradient all-reduce that every parameter whose name starts with one of `prefixes` has its
        final gradient enqueued on the current stream.  Only a CONTIGUOUS run of the flat buffer is announced;
        anything else is left to the reduction after backward."""
        hook = self.grad_ready_hook
        if hook is None:
            return
        cache = self.__dict__.setdefault("_announce_cache", {})
        if prefixes not in cache:
            lo = hi = None
            off = 0
            table = []
            for n, p in self._engine_params():
                if n.startswith(prefixes):
                    lo = off if lo is None else lo
                    hi = off + p.numel()
                table.append((n, off, off + p.numel()))
                off += p.numel()
            if lo is None or any(a >= lo and b <= hi and not n.startswith(prefixes) for n, a, b in table):
                lo = hi = None
            cache[prefixes] = (lo, hi)
        lo, hi = cache[prefixes]
        if lo is not None:
            hook(self, lo, hi)

    def flat_grad(self):
        self._grad_views()
        return self._gradbuf

    def _grad_views(self):
        pairs = self._engine_params()
        dev = pairs[0][1].device
        total = sum(p.numel() for _, p in pairs)
        buf = getattr(self, "_gradbuf", None)
        if buf is None or buf.device != dev or buf.numel() != total:
            buf = torch.zeros(total, dtype=torch.float32, device=dev)
            self._gradbuf = buf
        views, off = {}, 0
        for n, p in pairs:
            v = buf[off:off + p.numel()].view_as(p)
            off += p.numel()
            if p.requires_grad and (p.grad is None or p.grad.data_ptr() != v.data_ptr()):
                p.grad = v
            p._oat_engine_grad = True        # written in place by the kernels, overwritten by every backward (optim.zero_grad)
            views[n] = v
        return views
