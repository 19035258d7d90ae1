"""DistilBERT text encoder - parameter container with HF DistilBertModel state_dict names,
executed by OATrans.engine.text.TextEngine (HIP).  Stands in for the `AutoModel.from_pretrained(
text_params['model'])` call of the reference (/root/reference/OATrans/model/oa_model.py:27); the
HF implementation is third-party code, restated in oracle/oatrans_oracle.py."""
import json
import os
from types import SimpleNamespace

import torch
from torch import nn

from ..engine.module import EngineModule
from ..engine.text import TextEngine
from ..ops import hip

# dropout / attention_dropout: distilbert-base-uncased's config.json values; active in training mode only (the reference
# calls text_model.train(), oa_model.py:56, and the trainers model.train()), identity under .eval()
DEFAULT_CONFIG = dict(vocab_size=30522, max_position_embeddings=512, n_layers=6, n_heads=12, dim=768, hidden_dim=3072,
                      dropout=0.1, attention_dropout=0.1)


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.q_lin, self.k_lin, self.v_lin, self.out_lin = (nn.Linear(dim, dim) for _ in range(4))


class _FFN(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.lin1, self.lin2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)


class _Layer(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.attention = _Attn(dim)
        self.sa_layer_norm = nn.LayerNorm(dim, eps=1e-12)
        self.ffn = _FFN(dim, hidden)
        self.output_layer_norm = nn.LayerNorm(dim, eps=1e-12)


class _Embeddings(nn.Module):
    def __init__(self, vocab, max_pos, dim):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab, dim)
        self.position_embeddings = nn.Embedding(max_pos, dim)
        self.LayerNorm = nn.LayerNorm(dim, eps=1e-12)


class _Transformer(nn.Module):
    def __init__(self, n_layers, dim, hidden):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(dim, hidden) for _ in range(n_layers)])


class _HiddenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, input_ids, attention_mask, call_idx, launched, *params):
        # launched: (hidden, plan) of a call whose kernels DistilBertHIP.launch enqueued earlier - handed over explicitly by
        # DistilBertHIP.forward(launched=ticket); a plain forward never looks into module._launched (a ticket left behind by
        # an abandoned step must not be mistaken for this call's result)
        if launched is not None:             # only the graph node is new
            hidden, plan = launched
        else:
            hidden, plan = module._engine.forward(input_ids, attention_mask, module._param_data(),
                                                  module._weights_signature(), slot=call_idx, drop=module._dropout_args(input_ids.device))
        ctx.module, ctx.plan, ctx.call_idx = module, plan, call_idx
        return hidden.clone()

    @staticmethod
    def backward(ctx, d_hidden):
        m = ctx.module
        # a second call in the same step (caption + caption-with-tags, oa_model_global_local.py:161-164)
        # accumulates into the gradients the first call's backward wrote
        accumulate = m._bwd_calls > 0
        m._bwd_calls += 1
        m._engine.backward(ctx.plan, m._param_data(), m._grad_views(), d_hidden.float().contiguous(), accumulate)
        if m._bwd_calls == m._fwd_calls:
            m._announce(("",))               # last backward of the step: the whole flat gradient is final
        return (None, None, None, None, None) + (None,) * m._n_params


class DistilBertHIP(EngineModule):
    def __init__(self, config=None):
        super().__init__()
        cfg = dict(DEFAULT_CONFIG)
        cfg.update(config or {})
        self.config = SimpleNamespace(hidden_size=cfg["dim"], **cfg)
        self.embeddings = _Embeddings(cfg["vocab_size"], cfg["max_position_embeddings"], cfg["dim"])
        self.transformer = _Transformer(cfg["n_layers"], cfg["dim"], cfg["hidden_dim"])
        for mod in self.modules():
            if isinstance(mod, (nn.Linear, nn.Embedding)):
                nn.init.normal_(mod.weight, std=0.02)               # HF initializer_range
                if isinstance(mod, nn.Linear):
                    nn.init.zeros_(mod.bias)
        self._engine = TextEngine(cfg["n_layers"], cfg["dim"], cfg["n_heads"], cfg["hidden_dim"])
        self._bwd_calls = 0
        self._fwd_calls = 0
        self._launched = {}                  # call index -> (hidden, plan) of forwards enqueued by launch() and not yet attached
        self._rng_state = None
        self.dropout_seed = None             # None: drawn from torch's generator (torch.manual_seed governs it) on first use

    def _dropout_args(self, device):
        """(p_hidden, p_attention, device rng state) in training mode, None in eval mode or with both rates 0."""
        ph, pa = float(self.config.dropout), float(self.config.attention_dropout)
        if not self.training or (ph <= 0 and pa <= 0):
            return None
        if self._rng_state is None or self._rng_state.device != device:
            seed = self.dropout_seed if self.dropout_seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
            self._rng_state = hip.new_rng_state(seed, device)
        return ph, pa, self._rng_state

    def set_dropout_seed(self, seed):
        """Restart the mask stream: the same seed, the same sequence of forward calls -> the same masks."""
        self.dropout_seed = int(seed)
        self._rng_state = None

    @classmethod
    def from_pretrained(cls, path):
        """Reads an HF checkpoint directory (config.json + model.safetensors | pytorch_model.bin)."""
        with open(os.path.join(path, "config.json")) as fh:
            raw = json.load(fh)
        m = cls({k: raw[k] for k in DEFAULT_CONFIG if k in raw})
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
        sd = {k[len("distilbert."):] if k.startswith("distilbert.") else k: v for k, v in sd.items()}
        m.load_state_dict(sd, strict=False)
        return m

    def begin_step(self):
        """Called once per optimiser step: the next backward overwrites gradients."""
        self._bwd_calls = 0
        self._fwd_calls = 0
        self._launched.clear()

    def launch(self, input_ids=None, attention_mask=None, **unused):
        """Enqueue the forward kernels of one call NOW (on the current stream) and return a ticket for `forward(..., launched=ticket)`,
        which creates the autograd node LATER.  Why the two are separated: autograd runs ready backward nodes in reverse creation
        order, and the host issues a tower's whole backward from inside its node.  The model classes enqueue the text tower FIRST
        in forward (it runs beneath the first ViT blocks), which made its node the oldest and its backward the LAST thing the host
        issues - behind the video tower's ~350 launches and, with more than one rank, its 13 gradient all-reduce calls; under the
        queue's back-pressure the GPU then reached the text backward only after the video backward had finished (kernel trace of
        the W > 1 path: 1.3 ms of text backward alone on the GPU at the end of every step).  With the node created after the video
        forward, the text backward is issued first and runs beneath the top ViT blocks' backward.  Grad mode only."""
        if not torch.is_grad_enabled():
            return None
        if not input_ids.is_cuda:
            raise hip.OatError("DistilBertHIP runs on MI355X only (no CPU path); use the oracle for CPU")
        hip.lib()
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        self._new_step_guard()
        idx = self._fwd_calls
        if idx == 0:
            self._launched.clear()           # tickets of a step that was abandoned before its nodes were created
        self._fwd_calls += 1
        self._launched[idx] = self._engine.forward(input_ids, attention_mask, self._param_data(), self._weights_signature(),
                                                   slot=idx, drop=self._dropout_args(input_ids.device))
        return (idx, input_ids, attention_mask)

    def forward(self, input_ids=None, attention_mask=None, launched=None, **unused):
        if launched is not None:             # ticket of launch(): same call, kernels already enqueued
            idx, input_ids, attention_mask = launched
            held = self._launched.pop(idx, None)
            if held is None:
                raise RuntimeError("DistilBertHIP: stale launch ticket (begin_step() or an optimiser step came in between)")
            params = [p for _, p in self._engine_params()]
            return SimpleNamespace(last_hidden_state=_HiddenFn.apply(self, input_ids, attention_mask, idx, held, *params))
        if not input_ids.is_cuda:
            raise hip.OatError("DistilBertHIP runs on MI355X only (no CPU path); use the oracle for CPU")
        hip.lib()
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        params = [p for _, p in self._engine_params()]
        if torch.is_grad_enabled():
            self._new_step_guard()
            idx = self._fwd_calls
            if idx == 0:
                self._launched.clear()       # tickets of a step that was abandoned before its nodes were created
            self._fwd_calls += 1
        else:
            idx = -1                 # no backward will follow (validation): one dedicated plan, reused by every such call
        hidden = _HiddenFn.apply(self, input_ids, attention_mask, idx, None, *params)
        return SimpleNamespace(last_hidden_state=hidden)
