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
                st['exp_avg'] = r['m'][off:off + p.numel()].view_as(p)
                st['exp_avg_sq'] = r['v'][off:off + p.numel()].view_as(p)
                st.setdefault('step', 0)
                off += p.numel()
            r['sig'] = tuple((p.data.data_ptr(), p.grad.data_ptr()) for p in r['params'])
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
            self._build_runs()
        for r in self._runs:
            g = self.param_groups[r['group']]
            p0 = r['params'][0]
            st = self.state[p0]
            step = st['step'] + 1
            flat_p = torch.as_strided(p0.data, (r['n'],), (1,))
            flat_g = torch.as_strided(p0.grad, (r['n'],), (1,))
            hip.adamw(flat_p, flat_g, r['m'], r['v'], g['lr'], g['betas'][0], g['betas'][1], g['eps'],
                      g['weight_decay'], step, hf_style=self.hf_style, gscale=self.grad_scale)
            for p in r['params']:
                self.state[p]['step'] = step
        bump_weights_epoch()     # kernels wrote through raw pointers: tell the engines to re-cast shadows
        return loss

    def zero_grad(self, set_to_none=False):
        """Gradients are persistent buffers that every backward OVERWRITES; nothing to clear."""
        return None
