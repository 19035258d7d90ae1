"""A whole training step as ONE hipGraph launch.

The step of trainer/step.py is ~1100 kernel launches issued from Python through ctypes: 20-45 ms of host time per
step depending on the host CPU.  Measured on MI355X / ROCm 7: 28 ms eager -> 24 ms replayed (hipGraphLaunch still walks
its ~1100 kernel nodes on the host, so the gain is the Python share only).  The launch sequence of a step is static for a given input signature (all activations live in
per-shape plans, engine/video.py), so it is captured once with HIP stream capture (torch.cuda.graph: the ctypes
launches on torch's current stream, the side streams joined by events, autograd's backward and the optimiser are all
recorded) and replayed with a single hipGraphLaunch afterwards.

What makes the step capturable:
  * inputs are copied into static buffers before every replay (the graph holds raw pointers);
  * AdamW keeps step count / learning rate / bias corrections on the device (optim.AdamW.enable_capture);
  * nothing in the step synchronises with the host (the loss stays on the device).
A new input signature (shapes / dtypes) runs `warmup` eager steps first - real optimiser steps on the incoming
batches - and is then captured; single-process (world size 1) only: with more ranks the collectives of a step are
issued eagerly (trainer/step.py), where RCCL overlaps them with backward.
"""
import torch

from ..engine.module import bump_weights_epoch


def _signature(data):
    """Shapes / dtypes / devices of the TENSOR leaves only.  Everything else in a batch (`meta`: paths and raw captions,
    lists, strings) differs on every batch and is never read by the step, so it must not key - or defeat - the capture."""
    if isinstance(data, dict):
        sub = ((k, _signature(v)) for k, v in sorted(data.items()))
        return tuple((k, s) for k, s in sub if s is not None)
    if torch.is_tensor(data):
        return (tuple(data.shape), str(data.dtype), str(data.device))
    return None


def _clone(data):
    if isinstance(data, dict):
        return {k: _clone(v) for k, v in data.items()}
    return data.clone() if torch.is_tensor(data) else data


def _copy_into(dst, src):
    if isinstance(dst, dict):
        for k in dst:
            if isinstance(dst[k], dict) or torch.is_tensor(dst[k]):
                _copy_into(dst[k], src[k])
    elif torch.is_tensor(dst):
        dst.copy_(src, non_blocking=True)


class GraphedStep:
    MAX_SIGNATURES = 16              # distinct input signatures tracked; past that, new ones simply stay eager

    def __init__(self, step_fn, model_dp, loss_fn, optimizer, args, warmup=3):
        self.step_fn, self.model_dp, self.loss_fn, self.optimizer, self.args = step_fn, model_dp, loss_fn, optimizer, args
        self.warmup = max(1, int(warmup))
        self._seen = {}              # signature -> eager steps taken so far (bounded: MAX_SIGNATURES)
        self._graphs = {}            # signature -> (graph, static data, static loss)
        self.replays = 0
        self.failed = False

    def _eager(self, data):
        return self.step_fn(self.model_dp, self.loss_fn, self.optimizer, data, self.args)

    def __call__(self, data):
        sig = _signature(data)
        ent = self._graphs.get(sig)
        if ent is None:
            n = self._seen.get(sig, 0)
            if sig not in self._seen and len(self._seen) >= self.MAX_SIGNATURES:
                return self._eager(data)
            if n < self.warmup or n < 0:                       # warm-up: plans, workspaces, kernel attributes, optimiser state
                self._seen[sig] = n + 1
                return self._eager(data)
            try:
                ent = self._capture(sig, data)
            except Exception as exc:                  # capture is an optimisation: never let it take the run down
                print(f"### hipGraph capture failed ({type(exc).__name__}: {exc}); this input signature stays eager")
                self._seen[sig] = -(1 << 30)
                self.failed = True
                return self._eager(data)
            first = True
        else:
            first = False
        graph, static, loss = ent
        _copy_into(static, data)
        self.optimizer.sync_device_scalars()
        graph.replay()
        if not first:                                 # the capture pass already advanced the host-side counters once
            self.optimizer.note_replayed_step()
        # the replayed AdamW wrote the fp32 masters through raw pointers: an eager forward that follows (validation)
        # must re-cast the bf16 shadows, exactly as after an eager optimizer.step()
        bump_weights_epoch()
        self.replays += 1
        return loss

    def _capture(self, sig, data):
        self.optimizer.enable_capture()
        static = _clone(data)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = self._eager(static)
        ent = (graph, static, loss)
        self._graphs[sig] = ent
        return ent
