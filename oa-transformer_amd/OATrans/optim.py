"""Fused AdamW over flat parameter ranges (one HIP launch per contiguous run).

The reference resolves its optimiser by name on the `transformers` namespace
(`config.initialize('optimizer', transformers, trainable_params)`, train_dist_multi.py:66; config
"optimizer": {"type": "AdamW", "args": {"lr": 2e-4}}).  transformers >= 5 no longer ships AdamW, so
the entry points pass THIS module instead; `AdamW` here keeps transformers-4.6 semantics and defaults
(betas (0.9, 0.999), eps 1e-6, weight_decay 0, correct_bias True).
"""
import torch

from .engine.module import bump_weights_epoch
from .ops import hip


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True,
                 hf_style=True, grad_scale=1.0):
        if not correct_bias:
            raise NotImplementedError("correct_bias=False is unused by the shipped configs")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.hf_style = hf_style
        self.grad_scale = grad_scale
        self._runs = None
        self._stream = None          # eager mode: side stream the early launches run on
        self.eager_launches = 0      # launches issued from grad-ready announcements (introspection / tests)
        self._dev = None             # capturable mode: per-group device state {step, lr, coef} (see enable_capture)

    def attach(self, model):
        """EAGER mode (opt-in, single rank): the engine modules announce gradient ranges as soon as the kernels that
        write them are enqueued (`grad_ready_hook`, the same announcements the gradient all-reduce overlaps with);
        each announced range is updated right away on a side stream, under the rest of backward, instead of in
        `step()` after it.  `step()` then updates whatever was not announced and joins the side stream.  The result
        is identical to the plain `backward(); step()` sequence of the reference trainer (trainer_dist.py:163-166);
        do not attach when gradients are inspected or backward is run without a following `step()`.
        With more than one rank the announcements belong to the gradient all-reduce and nothing is attached."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return self
        for m in model.modules():
            if hasattr(m, "flat_grad") and hasattr(m, "_engine_params") and getattr(m, "grad_ready_hook", None) is None:
                m.grad_ready_hook = self._on_ready
        return self

    def _on_ready(self, module, lo, hi):
        """flat_grad()[lo:hi] of `module` is final once the work enqueued so far on the CURRENT stream is done."""
        if hi <= lo or self._runs is None or not self._runs_valid():
            return                               # first step (state not built yet) or buffers moved: step() does it all
        ptr = module.flat_grad().data_ptr() + 4 * lo
        n = hi - lo
        for r in self._runs:
            if r['g0'] <= ptr and ptr + 4 * n <= r['g0'] + 4 * r['n']:
                break
        else:
            return
        off = (ptr - r['g0']) // 4
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        ev = torch.cuda.Event()
        ev.record()
        self._stream.wait_event(ev)
        with torch.cuda.stream(self._stream), torch.no_grad():
            self._launch(r, off, off + n)
        r['done'].append((off, off + n))
        self.eager_launches += 1

    # ---- capturable mode: the step-dependent scalars live on the device -------------------------------------------
    def enable_capture(self):
        """Keep step count, learning rate and bias corrections in device memory (oat_adam_tick / oat_adamw_dev) so that
        step() can be captured once into a hipGraph and replayed: a captured kernel argument could not change from
        step to step.  The arithmetic is the same (transformers.AdamW formula), powf evaluated on the device."""
        if self._dev is not None:
            return self
        if not self._runs_valid():
            self._build_runs()
        self._dev = []
        for gi, g in enumerate(self.param_groups):
            runs = [r for r in self._runs if r['group'] == gi]
            dev = runs[0]['params'][0].device if runs else torch.device('cuda')
            cur = self.state[runs[0]['params'][0]]['step'] if runs else 0
            self._dev.append(dict(step=torch.full((1,), int(cur), dtype=torch.int32, device=dev),
                                  lr=torch.full((1,), float(g['lr']), dtype=torch.float32, device=dev),
                                  coef=torch.zeros(3, dtype=torch.float32, device=dev), lr_host=float(g['lr'])))
        return self

    def sync_device_scalars(self):
        """Before replaying a captured step: push a learning rate the trainer changed (trainer_dist.py:117-122)."""
        for g, d in zip(self.param_groups, self._dev or []):
            if float(g['lr']) != d['lr_host']:
                d['lr'].fill_(float(g['lr']))
                d['lr_host'] = float(g['lr'])

    def note_replayed_step(self):
        """A captured step() was replayed: advance the host-side step counters (state_dict / resume)."""
        for r in self._runs or []:
            step = self.state[r['params'][0]]['step'] + 1
            for p in r['params']:
                self.state[p]['step'] = step

    def _launch(self, r, a, b):
        g = self.param_groups[r['group']]
        if self._dev is not None:
            hip.adamw_dev(r['flat_p'][a:b], r['flat_g'][a:b], r['m'][a:b], r['v'][a:b], self._dev[r['group']]['coef'],
                          g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], hf_style=self.hf_style,
                          gscale=self.grad_scale)
            return
        step = self.state[r['params'][0]]['step'] + 1
        hip.adamw(r['flat_p'][a:b], r['flat_g'][a:b], r['m'][a:b], r['v'][a:b], g['lr'], g['betas'][0], g['betas'][1],
                  g['eps'], g['weight_decay'], step, hf_style=self.hf_style, gscale=self.grad_scale)

    def _build_runs(self):
        """Coalesce parameters that are adjacent in memory in BOTH .data and .grad (the engine modules
        keep them in flat buffers) into single launches; everything else is one launch per tensor."""
        runs = []
        for gi, group in enumerate(self.param_groups):
            cur = None
            for p in group['params']:
                if p.grad is None or not p.is_cuda:
                    continue
                n = p.numel()
                dp, gp = p.data.data_ptr(), p.grad.data_ptr()
                ds, gs = p.data.untyped_storage().data_ptr(), p.grad.untyped_storage().data_ptr()
                # adjacent AND inside the same allocation (neighbouring blocks of the caching allocator
                # can be address-adjacent without sharing a storage)
                if cur is not None and dp == cur['dend'] and gp == cur['gend'] and (ds, gs) == cur['stor']:
                    cur['params'].append(p)
                    cur['n'] += n
                else:
                    cur = dict(group=gi, params=[p], n=n, d0=dp, g0=gp, stor=(ds, gs))
                    runs.append(cur)
                cur['dend'], cur['gend'] = dp + 4 * n, gp + 4 * n
        for r in runs:
            dev = r['params'][0].device
            r['m'] = torch.zeros(r['n'], dtype=torch.float32, device=dev)
            r['v'] = torch.zeros(r['n'], dtype=torch.float32, device=dev)
            off = 0
            for p in r['params']:
                st = self.state[p]
                n = p.numel()
                m_view, v_view = r['m'][off:off + n].view_as(p), r['v'][off:off + n].view_as(p)
                # moments that already exist - a resumed checkpoint (load_state_dict), or runs rebuilt because a
                # buffer moved - are carried over into the flat buffers, not reset
                if torch.is_tensor(st.get('exp_avg')) and st['exp_avg'].shape == p.shape:
                    m_view.copy_(st['exp_avg'])
                if torch.is_tensor(st.get('exp_avg_sq')) and st['exp_avg_sq'].shape == p.shape:
                    v_view.copy_(st['exp_avg_sq'])
                st['exp_avg'], st['exp_avg_sq'] = m_view, v_view
                st['step'] = int(st.get('step', 0))
                off += n
            r['sig'] = tuple((p.data.data_ptr(), p.grad.data_ptr()) for p in r['params'])
            p0 = r['params'][0]
            r['flat_p'] = torch.as_strided(p0.data, (r['n'],), (1,))
            r['flat_g'] = torch.as_strided(p0.grad, (r['n'],), (1,))
            r['done'] = []
        self._runs = runs

    def _runs_valid(self):
        if self._runs is None:
            return False
        return all(tuple((p.data.data_ptr(), p.grad.data_ptr() if p.grad is not None else 0) for p in r['params']) == r['sig']
                   for r in self._runs)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if not self._runs_valid():
            if self._dev is not None:
                raise hip.OatError("AdamW: parameter / gradient buffers moved after enable_capture()")
            self._build_runs()
        if self._dev is not None:
            self.sync_device_scalars()
            for g, d in zip(self.param_groups, self._dev):
                hip.adam_tick(d['step'], d['lr'], g['betas'][0], g['betas'][1], d['coef'])
        for r in self._runs:
            pos = 0
            for a, b in sorted(r['done']):       # ranges already updated under backward (eager mode)
                if a > pos:
                    self._launch(r, pos, a)
                pos = max(pos, b)
            if pos < r['n']:
                self._launch(r, pos, r['n'])
            r['done'] = []
            step = self.state[r['params'][0]]['step'] + 1
            for p in r['params']:
                self.state[p]['step'] = step
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        bump_weights_epoch()     # kernels wrote through raw pointers: tell the engines to re-cast shadows
        return loss

    def load_state_dict(self, state_dict):
        """Resume.  Without flat buffers yet (or after the parameters moved) the loaded per-parameter moments are copied
        into them when the runs are next built.  With valid runs - in particular after enable_capture(), when captured
        graphs hold the addresses of the flat moment buffers and of the device-side step / lr scalars - the loaded
        state is copied INTO the existing buffers and the device scalars are re-synchronised, so the next (captured or
        eager) step continues from the loaded step count."""
        runs = self._runs if self._runs_valid() else None
        super().load_state_dict(state_dict)
        if runs is None:
            if self._dev is not None:
                raise hip.OatError("AdamW.load_state_dict: parameter / gradient buffers moved after enable_capture()")
            self._runs = None
            return
        with torch.no_grad():
            for r in runs:
                off = 0
                for p in r['params']:
                    st, n = self.state[p], p.numel()
                    m_view, v_view = r['m'][off:off + n].view_as(p), r['v'][off:off + n].view_as(p)
                    for key, view in (('exp_avg', m_view), ('exp_avg_sq', v_view)):
                        t = st.get(key)
                        if torch.is_tensor(t) and t.shape == p.shape:
                            if t.data_ptr() != view.data_ptr():
                                view.copy_(t)
                        else:
                            view.zero_()
                        st[key] = view
                    st['step'] = int(st.get('step', 0))
                    off += n
                r['done'] = []
            for gi, (g, d) in enumerate(zip(self.param_groups, self._dev or [])):
                own = [r for r in runs if r['group'] == gi]
                d['step'].fill_(int(self.state[own[0]['params'][0]]['step']) if own else 0)
                d['lr'].fill_(float(g['lr']))
                d['lr_host'] = float(g['lr'])

    def zero_grad(self, set_to_none=False):
        """Engine parameters keep their gradients in persistent flat buffers that every backward OVERWRITES (marked
        `_oat_engine_grad` by EngineModule._grad_views) - nothing to clear there.  Every other parameter (the
        projection heads and the loose parameters of the object-aware variants) is an ordinary autograd leaf whose
        AccumulateGrad ADDS: those gradients are zeroed in place, which keeps the pointers the fused launches were
        built on (the reference calls zero_grad every step, trainer_dist.py:156)."""
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is not None and not getattr(p, '_oat_engine_grad', False):
                    p.grad.zero_()
        return None
