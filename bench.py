#!/usr/bin/env python3
"""Headline benchmark: video-text pairs/s, forward + backward + optimiser step, 8-frame 224^2
ViT-B/16 + DistilBERT-base, per-GPU batch 32, N GPUs of one node (weak scaling).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement).  Synthetic inputs are resident in
HBM before the timed region; weights are random-init of the named architecture.  The `roofline`
object is measured live: one extra step with a HIP event pair around EVERY GEMM launch (forward /
data-gradient GEMMs and weight gradients), grouped by the kernel that serves the launch; the kernel
with the largest summed time is reported, the others follow in `other_gemm_kernels`.  Nothing is read
from files.  `cpu_baseline` times the CPU oracle (oracle/, a port of the reference's arithmetic) on
this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd"))
sys.path.insert(0, ROOT)

# dmabuf IPC: without it RCCL's intra-node transport fails with `hipIpcGetMemHandle: invalid argument` on this driver
# stack.  Must be in the environment before the HIP runtime initialises, i.e. before torch touches the GPU.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and NO launcher around it (no RANK / WORLD_SIZE in the environment):
    become the launcher - start N copies of this command, one rank per GPU, with the env rendezvous the reference's
    entry points read (train_dist_multi.py:35-38,127-132: MASTER_ADDR / MASTER_PORT / WORLD_SIZE / RANK / LOCAL_RANK),
    wait for them, and exit with the first non-zero status.  Rank 0's JSON line goes to this process's stdout.
    Returns None when there is nothing to launch (N == 1, or a launcher - torch.distributed.run - already set the env)."""
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--gpus", type=int, default=1)
    n = pre.parse_known_args()[0].gpus
    if n <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    import signal
    import subprocess
    import tempfile
    # rendezvous through a FILE store in a fresh temp dir (OAT_BENCH_INIT, read by main()): nothing to race for - a port found
    # by bind(0) + close can be taken by another process before the ranks bind it, and N ranks then hang until RCCL's timeout.
    # MASTER_ADDR / MASTER_PORT are still exported (what the reference's entry points read, train_dist_multi.py:35-38,127-132).
    rdv = tempfile.mkdtemp(prefix="oat_bench_rdv_")
    procs = []
    for r in range(n):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29500"), WORLD_SIZE=str(n), RANK=str(r),
                   LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n), OAT_BENCH_SELF_LAUNCHED="1", OAT_BENCH_INIT=f"file://{rdv}/store")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))

    def _stop(signum, frame):            # SIGTERM / SIGINT to the launcher: take exactly the ranks started above along
        for p in procs:
            if p.poll() is None:
                p.terminate()
        raise SystemExit(128 + signum)

    old = {sig: signal.signal(sig, _stop) for sig in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py: rank {r} exited with status {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    for o in pending:
                        procs[o].terminate()          # exactly the processes started above
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for sig, h in old.items():
            signal.signal(sig, h)
        import shutil
        shutil.rmtree(rdv, ignore_errors=True)
    return rc


if __name__ == "__main__":
    _rc = _self_launch()
    if _rc is not None:
        sys.exit(_rc)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0          # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
FP8_DENSE_PEAK_TFLOPS = 5000.0           # same guide: ~5 PF dense fp8 (v_mfma_f32_16x16x128_f8f6f4)


def flops_per_pair(T, N=196, D=768, depth=12, Lt=32, clips=None, text_passes=1):  # noqa: E302
    """Algorithmic fwd+bwd FLOPs per video-text pair (1 MAC = 2 FLOP, bwd = 2x fwd).
    clips: frame counts of the clips one sample sends through the video encoder (default (T,); the object-aware
    variants add a one-frame object clip); text_passes: DistilBERT passes per sample (global_local: caption + tags)."""
    def video(T):
        S = 1 + T * N
        return 2 * T * N * D * D + depth * (32 * S * D * D + 4 * D * (2 * S + N * T * (T + 1) + T * N * (N + 1))) + 2 * D * 256
    text = 6 * (24 * Lt * D * D + 4 * Lt * Lt * D) + 2 * D * 256
    return 3 * (sum(video(t) for t in (clips or (T,))) + text_passes * text)


def pruned_top_gflops(T, N=196, D=768, mlp_ratio=4):
    """GF per pair that VideoEngine.prune_top does not execute: the top block's space projection (D x D), fc1 and fc2
    (D x 4D each) on the T*N patch rows of a sample, forward + data gradient + weight gradient."""
    return 3 * T * N * 2 * D * D * (1 + 2 * mlp_ratio) / 1e9


def build(args, device):
    from OATrans import model as module_arch
    from OATrans.optim import AdamW
    from OATrans.parallel import HipDataParallel
    torch.manual_seed(1234)
    cls = {"frozen": module_arch.FrozenInTime, "region_mem": module_arch.oa_model_region_mem.FrozenInTime,
           "global_local": module_arch.oa_model_global_local.FrozenInTime}[args.variant]
    model = cls(
        video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224",
                          num_frames=args.frames, pretrained=True, time_init="rand", two_outputs=False,
                          object_clip="native",
                          **({"arch_kwargs": {"img_size": args.res}} if args.res != 224 else {})),
        object_params=dict(model="", input_objects=False),
        text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text"),
        projection="minimal", load_checkpoint="")
    # fan-in scaled random init so activations stay O(1) through 12 blocks (random data, not zeros:
    # zero-filled operands clock higher and would flatter the number)
    with torch.no_grad():
        for n, p in model.video_model.named_parameters():
            if p.dim() >= 2 and "embed" not in n and "cls" not in n:
                p.normal_(0, (p[0].numel()) ** -0.5)
    model = model.to(device)
    model.set_device(device)
    for m in (model.video_model, model.text_model):
        m.flatten_parameters()
    dp = HipDataParallel(model)
    opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=args.lr)
    # (optim.AdamW.attach - parameter updates start under backward - measured SLOWER, 56.5 -> 57.2 ms: not used here)
    loss_fn = module_arch.NormSoftmaxLoss()
    return dp, opt, loss_fn


def synthetic_batch(args, rank, device):
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    B, T, L = args.batch, args.frames, 32
    if args.variant != "frozen":
        T += 1                      # frame 0 = the object frame (its 10 / 5 box masks are in patch_masks), then the clip
    video = torch.randn(B, T, 3, args.res, args.res, generator=g).to(torch.bfloat16).to(device)
    ids = torch.randint(1000, 30000, (B, L), generator=g)
    ids[:, 0], ids[:, -1] = 101, 102
    batch = {"video": video, "text": {"input_ids": ids.to(device), "attention_mask": torch.ones(B, L, dtype=torch.int64, device=device)}}
    if args.variant != "frozen":
        from OATrans.data_loader.data_loader import MultiDistTextObjectVideoDataLoader
        O = 5 if args.variant == "region_mem" else 10
        dl = MultiDistTextObjectVideoDataLoader("Synthetic", {"max_length": L}, {"input_res": args.res, "num_frames": 1}, "",
                                                batch_size=B, object_params={"input_objects": True, "num_objects": O})
        extra = dl.make_batch(4321 + rank, device)
        for k in ("patch_masks", "object_token_masks", "object_token_len", "pad_text", "text_region_embedding"):
            batch[k] = extra[k]
    return batch


def other_config_line(base_args, variant, device, steps=10, warmup=3, label=None, **overrides):
    """One more workload of BASELINE.json's `configs`, run AFTER the headline's timed region and reported under
    `other_configs`, outside `value`: a short (warmup + steps) single-GPU measurement with the same step function,
    optimiser and timing brackets as the headline.  variant: frozen | region_mem | global_local, with the suffix
    `_full` for the same model and step with VideoEngine.prune_top OFF (the reference's full graph; the default schedule
    skips the top block's unused patch rows, see pruned_top_gflops); overrides: frames / batch / res / dtype of the run
    (default: the headline's); label: the BASELINE.json config the entry stands for."""
    import copy
    import gc
    from OATrans.trainer.step import global_local_step, hot_step, region_mem_step
    args = copy.copy(base_args)
    for k, v in overrides.items():
        setattr(args, k, v)
    full_graph = variant.endswith("_full")
    args.variant = variant[:-len("_full")] if full_graph else variant
    step_impl = {"region_mem": region_mem_step, "global_local": global_local_step, "frozen": hot_step}[args.variant]
    dp, opt, loss_fn = build(args, device)
    if args.dtype == "fp8":
        dp.module.video_model._engine.fp8 = True
    if full_graph:
        dp.module.video_model._engine.prune_top = False
    data = synthetic_batch(args, 0, device)
    step_args = argparse.Namespace(world_size=1, rank=0, local_rank=device.index or 0)
    for _ in range(warmup):
        step_impl(dp, loss_fn, opt, data, step_args)
    pruned = any(pl.prune_top for pl in dp.module.video_model._engine.plans.values())     # what the engine actually ran
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step_impl(dp, loss_fn, opt, data, step_args)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    value = args.batch * steps / elapsed
    N = (args.res // 16) ** 2
    oa = args.variant != "frozen"
    full = flops_per_pair(args.frames, N=N, clips=(1, args.frames) if oa else None,
                          text_passes=2 if args.variant == "global_local" else 1) / 1e9
    cls_name = {"frozen": "oa_model", "region_mem": "oa_model_region_mem", "global_local": "oa_model_global_local"}[args.variant]
    n_obj = {"region_mem": 5, "global_local": 10}.get(args.variant)
    dtype = "bf16" if args.dtype == "bf16" else "fp8 e4m3 forward linears (per-tensor delayed scaling) + bf16 backward"
    common = {"value": round(value, 2), "unit": "pairs/s", "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps,
              "warmup": warmup, "dtype": dtype, "per_gpu_batch": args.batch, "frames": args.frames, "res": args.res}
    if pruned:
        # the utilisation of a pruned run counts what was EXECUTED; the clips of an OA model are pruned one by one
        gf_pair = full - pruned_top_gflops(args.frames + (1 if oa else 0), N)
        what = ("oa_model_region_mem.FrozenInTime takes the CLS rows of the encoder output and the patch rows of block 6 (region tap), never "
                f"the final patch rows - the top block's space projection / norm2 / fc1 / GELU / fc2 run on the 2 x {args.batch} CLS rows of "
                "the object frame and the clip") if oa else \
               ("oa_model.FrozenInTime consumes only the CLS row of the encoder output, so the top block's space projection / norm2 / fc1 / "
                f"GELU / fc2 run on the {args.batch} CLS rows instead of all {args.batch * (args.frames * N + 1)} (forward, data and weight gradients)")
        clip = f"{args.frames}-frame + {n_obj} obj (one object frame with {n_obj} box masks + the {args.frames}-frame clip, same encoder)" \
            if oa else f"{args.frames}-frame"
        line = {"workload": f"[{label or args.variant}] {clip} {args.res}^2 ViT-B/16 + DistilBERT-base ({cls_name}.FrozenInTime), "
                            f"bs {args.batch}, fwd+bwd+AdamW; default schedule (VideoEngine.prune_top): {what}; same loss, same "
                            "gradients as the full graph (tests/test_prune_gpu.py)",
                **common, "gflop_per_pair": round(gf_pair, 1), "gflop_per_pair_full_graph": round(full, 1),
                "step_mfma_frac": round(value * gf_pair / 1e3 / BF16_DENSE_PEAK_TFLOPS, 4)}      # from the EXECUTED FLOPs
    else:
        clip = f"{args.frames}-frame + {n_obj} obj (one object frame with {n_obj} box masks + the {args.frames}-frame clip, same encoder)" \
            if oa else f"{args.frames}-frame"
        tag = ", full graph (OAT_PRUNE_TOP=0: the top block's discarded patch rows computed as the reference does)" if full_graph else ""
        line = {"workload": f"[{label or variant}] {clip} {args.res}^2 ViT-B/16 + DistilBERT-base ({cls_name}.FrozenInTime), "
                            f"bs {args.batch}, fwd+bwd+AdamW{tag}",
                **common, "gflop_per_pair": round(full, 1),
                "step_mfma_frac": round(value * full / 1e3 / BF16_DENSE_PEAK_TFLOPS, 4)}    # against the bf16 roof in either dtype
    line["final_loss"] = round(float(loss.item()), 4)
    del dp, opt, data, loss
    gc.collect()
    torch.cuda.empty_cache()
    return line


def other_config_plan(args):
    """What a default 1-GPU run appends under `other_configs`: config 3 as BASELINE.json words it (both object-aware
    classes, objects on), the full-graph runs of the two classes whose default schedule prunes the top block, and the
    per-GPU shapes of configs 2, 4 and 5 (config 5 in both dtypes: bf16 is its in-tolerance form, fp8 forward the opt-in,
    DESIGN section 7; in bf16 also at twice the per-GPU batch).  At another geometry
    (--other-configs at test sizes) the shapes scale with the command line: half the frames, twice the batch, and
    twice the frames at 336^2 with a quarter of the batch."""
    b, f = args.batch, args.frames
    c5 = dict(frames=2 * f, res=336 if args.res == 224 else args.res, batch=max(2, b // 4))
    return [("global_local", dict(label="config 3, global_local")),
            ("region_mem", dict(label="config 3, region_mem")),
            ("frozen_full", dict(label="frozen, full graph")),
            ("region_mem_full", dict(label="region_mem, full graph")),
            ("frozen", dict(label="config 2", frames=max(1, f // 2), steps=5, warmup=2)),
            ("frozen", dict(label="config 4, per-GPU shape", batch=2 * b, steps=5, warmup=2)),
            ("global_local", dict(label="config 5 geometry, bf16", dtype="bf16", steps=5, warmup=2, **c5)),
            ("global_local", dict(label="config 5 geometry, bf16, twice the batch", dtype="bf16", steps=4, warmup=2, **dict(c5, batch=2 * c5["batch"]))),
            ("global_local", dict(label="config 5 geometry, fp8 forward", dtype="fp8", steps=5, warmup=2, **c5))]


def _gemm_class(kind, epi, M, N, K):
    """Name of the kernel that serves a launch (the dispatch rules of csrc/gemm_nt.hip / gemm_tn.hip), used to group
    the instrumented launches exactly as rocprofv3 --stats groups them."""
    epis = {0: "EPI_BF16", 1: "EPI_F32", 2: "EPI_GELU_DUAL", 3: "EPI_DGELU", 4: "EPI_F32_BF16", 5: "EPI_GELU_GRAD", 6: "EPI_MUL_AUX"}
    if kind == "tn":
        big = M >= 4096 and N % 256 == 0 and K % 256 == 0          # (N, K) = (N1, N2) here
        return "gemm_tn_pp_kernel (+ tn_reduce)" if big else "gemm_tn_kernel<2,2,4,4> (+ tn_reduce)"
    u8 = bool(epi & 0x100)                     # 8-bit GELU derivative: always the ping-pong kernel
    epi &= 0xff
    big = M >= 4096 and N % 256 == 0 and (u8 or ((M + 255) // 256) * (N // 256) * 5 >= 256 * 2)
    nk = K // 64
    if big and epi in (0, 5, 6) and N <= 4096 and nk >= 2 and nk % 2 == 0:
        return f"gemm_nt_pp_kernel<{epis[epi]}>"
    return f"gemm_nt_kernel<{epis.get(epi, epi)},{'2,4,8,4' if big else '2,2,4,4'}>"


PMC_PASS_TIMEOUT_S = 180          # one rocprofv3 --pmc pass over the two-step child takes 10-20 s
# Optional legs of a default 1-GPU run (traffic, w1_forced, other_configs, cpu_baseline) are skipped - with the reason in the line -
# once the run has used this much wall time: the ONE JSON line is printed at the very end, and a leg that crawls (a cold box, a
# profiler that hangs) must not cost the headline.  A normal run takes 70-120 s in all.
OPTIONAL_LEG_BUDGET_S = float(os.environ.get("OAT_BENCH_BUDGET_S", "900"))
_T_START = time.time()


def _budget_left():
    return OPTIONAL_LEG_BUDGET_S - (time.time() - _T_START)


_ALGO_BYTES = {}      # kernel class -> [algorithmic bytes of each launch of the instrumented step]


def instrumented_gemm_profile(step_fn):
    """Run one step with a HIP event pair around every GEMM launch - data / forward GEMMs (oat_gemm_nt) AND weight
    gradients (oat_gemm_tn) - recorded on the stream the launch goes to.  Returns {kernel class: flops, ms, launches}."""
    from OATrans.ops import hip
    records = []

    # raw HIP events created with hipEventDisableSystemFence ("for events that only measure timing"): torch's timing
    # events carry a system-scope release fence per record, which adds ~30 us around every launch of the instrumented step
    import ctypes
    rt = ctypes.CDLL("libamdhip64.so")
    rt.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    rt.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    rt.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
    rt.hipEventDestroy.argtypes = [ctypes.c_void_p]

    class Ev:
        def __init__(self):
            self.h = ctypes.c_void_p()
            if rt.hipEventCreateWithFlags(ctypes.byref(self.h), 0x20000000) != 0:
                raise RuntimeError("hipEventCreateWithFlags failed")

        def record(self):
            if rt.hipEventRecord(self.h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) != 0:
                raise RuntimeError("hipEventRecord failed")

        def elapsed_time(self, other):
            ms = ctypes.c_float()
            if rt.hipEventElapsedTime(ctypes.byref(ms), self.h, other.h) != 0:
                raise RuntimeError("hipEventElapsedTime failed")
            return ms.value

    orig_nt, orig_tn, orig_f8 = hip.gemm_nt, hip.gemm_tn, hip.gemm_nt_f8

    def timed_f8(A8, B8, M, N, K, epi, out, dq_a, dq_b, **kw):
        s, e = Ev(), Ev()
        s.record()
        orig_f8(A8, B8, M, N, K, epi, out, dq_a, dq_b, **kw)
        e.record()
        records.append((f"gemm_nt_pp_kernel<{'EPI_GELU_GRAD' if epi == 5 else 'EPI_BF16'},fp8>", 2.0 * M * N * K, s, e))

    def timed_nt(A, B, M, N, K, epi, out, **kw):
        s, e = Ev(), Ev()
        s.record()
        orig_nt(A, B, M, N, K, epi, out, **kw)
        e.record()
        records.append((_gemm_class("nt", epi, M, N, K), 2.0 * M * N * K, s, e))
        _ALGO_BYTES.setdefault(records[-1][0], []).append(2.0 * (M * K + N * K + M * N))      # bf16 A + W in, bf16 C out

    def timed_tn(P, Q, M, N1, N2, out, **kw):
        s, e = Ev(), Ev()
        s.record()
        orig_tn(P, Q, M, N1, N2, out, **kw)
        e.record()
        records.append((_gemm_class("tn", 0, M, N1, N2), 2.0 * M * N1 * N2, s, e))

    orig_grp = hip.TnGroup.run

    def timed_grp(self):
        # grouped weight gradients (csrc/gemm_tn_sk.hip): one GEMM launch + its fix-up for several problems
        s, e = Ev(), Ev()
        s.record()
        orig_grp(self)
        e.record()
        fl = sum(2.0 * (r[4] & 0xffffffff) * (r[4] >> 32) * (r[5] & 0xffffffff) for r in self.table.tolist())
        big = (self.table[0, 4].item() & 0xffffffff) >= 4096
        records.append(("gemm_tn_sk_kernel (+ tn_sk_fix)" if big else "gemm_tn_sk_kernel [DistilBERT, 36 problems]", fl, s, e))

    hip.gemm_nt, hip.gemm_tn, hip.gemm_nt_f8, hip.TnGroup.run = timed_nt, timed_tn, timed_f8, timed_grp
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        hip.gemm_nt, hip.gemm_tn, hip.gemm_nt_f8, hip.TnGroup.run = orig_nt, orig_tn, orig_f8, orig_grp
    by = {}
    for name, fl, s, e in records:
        d = by.setdefault(name, dict(flops=0.0, ms=0.0, n=0))
        d["flops"] += fl
        d["ms"] += s.elapsed_time(e)
        d["n"] += 1
    for _, _, s, e in records:
        rt.hipEventDestroy(s.h)
        rt.hipEventDestroy(e.h)
    return by


def traffic_child(args):
    """The process the rocprofv3 --pmc passes wrap: the model of the main run, one warm-up step (tapes are recorded)
    and ONE step - no timing, no instrumentation, no output."""
    from OATrans.trainer.step import hot_step
    device = torch.device("cuda:0")
    dp, opt, loss_fn = build(args, device)
    if args.dtype == "fp8":
        dp.module.video_model._engine.fp8 = True
    data = synthetic_batch(args, 0, device)
    step_args = argparse.Namespace(world_size=1, rank=0, local_rank=0)
    for _ in range(2):
        hot_step(dp, loss_fn, opt, data, step_args)
    torch.cuda.synchronize()


def hbm_traffic(args, kernel_class, algo_bytes):
    """`roofline.traffic`: L2-miss bytes per launch of the roofline kernel from the PMC counters, collected exactly as
    /opt/skills/guides/MI355X_MICROARCH.md (HBM section, rocprofv3 PMC slots) prescribes: FETCH_SIZE and WRITE_SIZE in
    SEPARATE `rocprofv3 --kernel-trace --pmc` passes (they do not fit one pass), counters in KiB, and on gfx950
    FETCH_SIZE tallies 128-byte requests at 64 bytes, so reads = 2 x FETCH_SIZE.  Infinity-Cache hits are counted, so
    the figure is an upper bound on HBM bytes.  Each pass wraps `bench.py --traffic-child` (same model, two steps)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return {"skipped": "rocprofv3 not on PATH"}
    try:
        import pandas as pd
    except ImportError:
        return {"skipped": "pandas missing (needed to read rocprofv3's CSV)"}
    if "gemm_nt_pp_kernel<EPI_BF16" not in kernel_class:
        return {"skipped": f"no rocprofv3 name pattern for {kernel_class}"}
    pattern = "gemm_nt_pp_kernel<0,"          # every <EPI_BF16, flags> instance of the ping-pong GEMM
    tmp = tempfile.mkdtemp(prefix="oat_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--traffic-child", "--batch", str(args.batch), "--frames", str(args.frames),
             "--res", str(args.res), "--dtype", args.dtype, "--lr", str(args.lr)]
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "run", "--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=PMC_PASS_TIMEOUT_S)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"skipped": f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"}
            df = pd.read_csv(files[0])
            df = df[(df.Counter_Name == counter) & df.Kernel_Name.str.replace(" ", "").str.contains(pattern, regex=False)]
            d = df.groupby("Dispatch_Id").Counter_Value.sum()
            per[counter] = (float(d.mean()) * 1024.0, int(d.count()))
    except subprocess.TimeoutExpired:
        return {"skipped": f"rocprofv3 pass timed out ({PMC_PASS_TIMEOUT_S} s)"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    read_b, write_b = 2.0 * per["FETCH_SIZE"][0], per["WRITE_SIZE"][0]
    algo = sum(algo_bytes) / max(1, len(algo_bytes))
    return {"read_mb": round(read_b / 1e6, 1), "write_mb": round(write_b / 1e6, 1), "total_mb": round((read_b + write_b) / 1e6, 1),
            "algorithmic_mb": round(algo / 1e6, 1), "ratio": round((read_b + write_b) / algo, 3), "launches_counted": per["FETCH_SIZE"][1],
            "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) around `bench.py --traffic-child` "
                      "(2 steps of this workload); per-launch mean over every gemm_nt_pp_kernel<EPI_BF16, *> dispatch; reads = 2 x "
                      "FETCH_SIZE KiB (gfx950 correction), writes = WRITE_SIZE KiB; memory-side L2 counters, Infinity-Cache hits included"}


def forced_w1(dp, eager_step, batch, device, steps=5, warmup=2):
    """The W > 1 launch path on ONE GPU: a 1-rank RCCL group, GradSync(force=True) - the ViT block by block and the text
    tower announce their gradient ranges from inside backward and each range goes out as an asynchronous all-reduce on
    RCCL's stream (identity at one rank, but the same kernels, stream edges and tape segments) - and the backward GEMM
    grid of a multi-rank job.  Timed like the headline; reported beside it as `w1_forced`: the per-GPU cost of the
    multi-GPU machinery before any link time, i.e. an upper bound on the scaling efficiency a node can show."""
    import shutil
    import tempfile
    from OATrans.parallel import GradSync
    rdv = tempfile.mkdtemp(prefix="oat_bench_w1_")           # file store: no port to race for
    dist.init_process_group("nccl", init_method=f"file://{rdv}/store", rank=0, world_size=1)
    old_sync = dp.sync
    engines = [m._engine for m in dp.module.modules() if hasattr(getattr(m, "_engine", None), "bwd_nt_grid")]
    old_grids = [e.bwd_nt_grid for e in engines]
    out = {}
    try:
        for name, grid_env in (("default (one workgroup per tile in backward)", None), ("OAT_BWD_NT_GRID=auto (CUs - 16 persistent workgroups)", "auto")):
            prev = os.environ.get("OAT_BWD_NT_GRID")
            if grid_env is not None:
                os.environ["OAT_BWD_NT_GRID"] = grid_env
            try:
                dp.sync = GradSync(dp.module, overlap=True, force=True)
            finally:
                if grid_env is not None:
                    os.environ.pop("OAT_BWD_NT_GRID") if prev is None else os.environ.__setitem__("OAT_BWD_NT_GRID", prev)
            for _ in range(warmup):
                eager_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                eager_step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            out[name] = {"ms_per_step": round(ms, 3), "pairs_per_s": round(batch / ms * 1e3, 2), "bwd_nt_grid": dp.sync.bwd_nt_grid,
                         "async_all_reduces_per_step": dp.sync.started_last_step}
    finally:
        dp.sync = old_sync
        for m in dp.module.modules():
            if getattr(m, "grad_ready_hook", None) is not None:
                m.grad_ready_hook = None
        for e, g in zip(engines, old_grids):
            e.bwd_nt_grid = g
        dist.destroy_process_group()
        shutil.rmtree(rdv, ignore_errors=True)
    first = next(iter(out.values()))
    return {"ms_per_step_w1_forced": first["ms_per_step"], "steps": steps, "warmup": warmup, "variants": out,
            "what": "1-rank RCCL group, GradSync(force=True): per-block asynchronous gradient all-reduces started from inside backward + "
                    "the multi-rank backward GEMM grid; compare with ms_per_step of this line"}


def _cpu_sample(frames, threads, budget, max_iters, min_iters=1):
    """fp32 CPU oracle (port of the reference arithmetic): bs 2, fwd+bwd, `frames` frames; seconds per iteration."""
    from OATrans.utils import seeded_init as si
    from oracle import oatrans_oracle as orc
    torch.set_num_threads(threads)
    p = si.frozen_state_dict(7, dict(num_frames=frames), {})
    for v in p.values():
        v.requires_grad_(True)
    B, L = 2, 32
    video = torch.randn(B, frames, 3, 224, 224)
    ids = torch.randint(1000, 30000, (B, L))
    mask = torch.ones(B, L, dtype=torch.int64)

    def one():
        loss, _, _, _ = orc.train_step_loss(p, video, ids, mask)
        loss.backward()

    tw = time.time()
    one()
    warm = time.time() - tw
    t0 = time.time()
    n = 0
    while n < min_iters or (n < max_iters and (time.time() - t0) + warm < budget):
        one()
        n += 1
    return (time.time() - t0) / n, n


def cpu_baseline(frames):
    """The CPU oracle on this box's host cores, on a bounded sample (about 25 s of CPU work in total): the headline
    shape at 8 threads (the count BASELINE.md quotes for the real reference) plus, as `extra`, config 1's shape (1 frame)
    and the headline shape on more cores.  torch's CPU kernels collapse when every SMT thread of the GPU box is used
    (measured 325 s / iteration on 256 threads vs ~3 s on 8), so the wide run uses a quarter of the logical CPUs."""
    ncpu = os.cpu_count() or 1
    t8 = min(8, ncpu)
    dt, n = _cpu_sample(frames, t8, 14.0, 4)
    out = dict(value=round(2 / dt, 4), unit="pairs/s", cores=t8, kind="port",
               sample=f"oracle fwd+bwd, bs 2, {frames} frames 224^2, Lt 32, {n} timed iterations after 1 warm-up "
                      f"({dt:.2f} s/iter); reference itself measured 0.236 pairs/s on 8 threads (BASELINE.md)")
    extra = []
    dt1, n1 = _cpu_sample(1, t8, 4.0, 3)
    extra.append(dict(workload="config 1: 1 frame 224^2, bs 2", cores=t8, value=round(2 / dt1, 4), unit="pairs/s",
                      sample=f"{n1} iterations, {dt1:.2f} s/iter"))
    wide = max(t8, min(64, ncpu // 4))
    if wide > t8:
        dtw, nw = _cpu_sample(frames, wide, 8.0, 3, min_iters=2)       # at least two timed iterations after the warm-up
        extra.append(dict(workload=f"{frames} frames 224^2, bs 2", cores=wide, value=round(2 / dtw, 4), unit="pairs/s",
                          sample=f"{nw} iterations, {dtw:.2f} s/iter ({ncpu} logical CPUs on this host)"))
    out["extra"] = extra
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--res", type=int, default=224, help="frame size (336 -> 441 patches per frame, BASELINE config 5's geometry)")
    ap.add_argument("--lr", type=float, default=2e-5,
                    help="AdamW step size (the reference config uses 2e-4 on PRETRAINED towers; random-init towers on one repeated "
                         "synthetic batch spike at that value, which says nothing about throughput but makes final_loss useless)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip `roofline.traffic` (two rocprofv3 --pmc passes over one step, run as subprocesses of a 1-GPU run)")
    ap.add_argument("--no-forced-w1", action="store_true",
                    help="skip `w1_forced` (the W > 1 launch path - asynchronous per-block gradient all-reduces from inside backward, "
                         "the backward GEMM grid of a multi-rank job - on a 1-rank RCCL group, 1-GPU run only)")
    ap.add_argument("--force-w1-main", action="store_true",
                    help="dev: run the MAIN timed loop on the W > 1 launch path of a 1-rank RCCL group (what `w1_forced` measures), e.g. under a kernel trace")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)     # the process rocprofv3 wraps (see hbm_traffic)
    ap.add_argument("--full-graph", action="store_true",
                    help="run the MAIN line with VideoEngine.prune_top off (= OAT_PRUNE_TOP=0): the reference's full graph, whose top "
                         "block computes patch rows nothing consumes.  Default: the exact pruned schedule (same loss, same gradients); "
                         "the line then carries gflop_per_pair = EXECUTED and gflop_per_pair_full_graph beside it, and a default "
                         "run reports the full graph as an entry of `other_configs`")
    ap.add_argument("--other-configs", action="store_true",
                    help="append the `other_configs` runs at any --frames / --res (default: only at the headline geometry, 8 x 224^2)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short config-3 runs (object-aware variants) that a default 1-GPU run appends under `other_configs`")
    ap.add_argument("--dtype", choices=["bf16", "fp8"], default="bf16",
                    help="fp8 (BASELINE config 5): the six forward linears of every ViT block on OCP e4m3 MFMA with per-tensor "
                         "delayed scaling; attention, LayerNorm, loss and the whole backward stay bf16 / fp32")
    ap.add_argument("--variant", choices=["frozen", "region_mem", "global_local"], default="frozen",
                    help="frozen = oa_model.FrozenInTime (headline).  The OA variants (BASELINE config 3) take one object "
                         "frame (box masks on its 14x14 patch grid) + a --frames clip per sample, both through the same "
                         "encoder weights ('native' object-clip layout, oa_model_global_local.py docstring)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if os.environ.get("OAT_BENCH_ECHO_RANK") == "1":       # launcher test hook: what this rank was started with
        print(f"bench.py rank {rank}/{world} LOCAL_RANK={local} MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')} "
              f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}", file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path for the product)")
    if os.environ.get("OAT_BENCH_ONE_DEVICE") == "1":      # dry run of the N > 1 path on a one-GPU box (with OAT_BENCH_BACKEND=gloo)
        local = 0
    if args.traffic_child:
        return traffic_child(args)
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local} but only {torch.cuda.device_count()} are visible "
                         f"(--gpus {args.gpus}); OAT_BENCH_ONE_DEVICE=1 OAT_BENCH_BACKEND=gloo runs every rank on GPU 0 as a dry run")
    torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}")
    if world > 1:
        # self-launched ranks meet in a file store (no port to race for); under torch.distributed.run the launcher's env rendezvous
        init = os.environ.get("OAT_BENCH_INIT") or "tcp://{}:{}".format(os.environ.get("MASTER_ADDR", "127.0.0.1"),
                                                                        os.environ.get("MASTER_PORT", "29500"))
        dist.init_process_group(backend=os.environ.get("OAT_BENCH_BACKEND", "nccl"), init_method=init, rank=rank, world_size=world)
    from OATrans.trainer.step import global_local_step, hot_step, region_mem_step
    step_impl = {"frozen": hot_step, "region_mem": region_mem_step, "global_local": global_local_step}[args.variant]
    dp, opt, loss_fn = build(args, device)
    if args.dtype == "fp8":
        dp.module.video_model._engine.fp8 = True
    if args.full_graph:
        dp.module.video_model._engine.prune_top = False
    data = synthetic_batch(args, rank, device)
    step_args = argparse.Namespace(world_size=world, rank=rank, local_rank=local)
    if args.force_w1_main and world == 1:
        from OATrans.parallel import GradSync
        import tempfile
        dist.init_process_group("nccl", init_method=f"file://{tempfile.mkdtemp(prefix='oat_bench_w1_')}/store", rank=0, world_size=1)
        dp.sync = GradSync(dp.module, overlap=True, force=True)
        args.no_forced_w1 = args.no_traffic = True

    def eager_step():
        return step_impl(dp, loss_fn, opt, data, step_args)

    # Launch path.  Default: the encoders' forward / backward schedules replay from launch tapes (csrc/tape.hip: the
    # recorded launches are re-issued from C at ~4 us each), the ~150 remaining launches of a step (loss, projections,
    # optimiser, collectives) are issued eagerly - the same path at every rank count.  OAT_GRAPH_STEP=1 (one rank only)
    # additionally captures the whole step into a hipGraph (trainer/graph_step.py); measured equal on the GPU, and
    # hipGraphLaunch costs more host time per kernel node than the tapes do.
    use_graph = world == 1 and os.environ.get("OAT_GRAPH_STEP", "0") == "1"
    if use_graph:
        from OATrans.trainer.graph_step import GraphedStep
        graphed = GraphedStep(step_impl, dp, loss_fn, opt, step_args, warmup=2)
        step = lambda: graphed(data)
        for _ in range(3):                 # 2 eager warm-up steps + the capture (+ first replay): outside any timed region
            step()
    else:
        step = eager_step

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_elapsed = time.perf_counter() - t0            # host-side enqueue time (GPU still running)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0                     # this rank's own finish time (before the closing barrier)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = [own / args.steps * 1e3]
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        owns = [torch.zeros(1, device=device, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(owns, torch.tensor([own], device=device, dtype=torch.float64))
        rank_ms = [o.item() / args.steps * 1e3 for o in owns]
    loss_val = float(loss.item())
    pairs = world * args.batch * args.steps
    value = pairs / elapsed
    oa = args.variant != "frozen"
    gf_pair = flops_per_pair(args.frames, N=(args.res // 16) ** 2, clips=(1, args.frames) if oa else None,
                             text_passes=2 if args.variant == "global_local" else 1) / 1e9
    gf_full = gf_pair
    pruned = any(pl.prune_top for pl in dp.module.video_model._engine.plans.values())     # what the engine actually ran
    if pruned:                         # the utilisation figures of a pruned run count what was EXECUTED
        gf_pair -= pruned_top_gflops(args.frames + (1 if oa else 0), (args.res // 16) ** 2)
    n_obj = {"region_mem": 5, "global_local": 10}.get(args.variant)
    clip_txt = f"{args.frames}-frame + {n_obj} obj (one object frame with {n_obj} box masks + the {args.frames}-frame clip, same encoder)" \
        if oa else f"{args.frames}-frame"
    if pruned:
        clip_txt += " (default schedule: the top block's space projection / norm2 / fc1 / GELU / fc2 run on the CLS rows only - its patch rows are never consumed, video_transformer.py:349-351; gflop_per_pair = executed; --full-graph runs the rest too)"
    cls_name = {"frozen": "oa_model", "region_mem": "oa_model_region_mem", "global_local": "oa_model_global_local"}[args.variant]
    out = {
        "metric": "video-text pairs/sec fwd+bwd, 8-frame ViT-B/16, 1/2/4/8 MI355X",
        "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "fp8 e4m3 forward linears (per-tensor delayed scaling) + bf16 backward",
        "data": "synthetic",
        "config": {"workload": f"[{args.variant}] {clip_txt} {args.res}^2 ViT-B/16 SpaceTimeTransformer + DistilBERT-base ({cls_name}.FrozenInTime), "
                               f"bs {args.batch}/GPU, Lt 32, fwd+bwd+AdamW, InfoNCE over all-gathered embeddings",
                   "per_gpu_batch": args.batch, "global_batch": world * args.batch, "frames": args.frames,
                   "parallelism": f"dp{world}", "gflop_per_pair": round(gf_pair, 1), "gflop_per_pair_full_graph": round(gf_full, 1)},
        "step_algorithmic_tflops_per_gpu": round(value / world * gf_pair / 1e3, 1),
        "step_mfma_frac": round(value / world * gf_pair / 1e3 / BF16_DENSE_PEAK_TFLOPS, 4),
        "final_loss": round(loss_val, 4),
        "host_loop_ms_per_step": round(host_elapsed / args.steps * 1e3, 2),     # host time per step INSIDE the timed loop: includes waiting for queue space behind the GPU
        "ranks_in_group": dist.get_world_size() if world > 1 else 1,
        "rank_ms_per_step": [round(x, 3) for x in rank_ms],
        "grad_exchange_dtype": os.environ.get("OAT_GRAD_DTYPE", "fp32"),
        "launch_mode": "hipGraph replay (1 launch per step)" if use_graph else
                       ("launch tapes replayed from C (encoders) + eager launches (loss, optimiser)" if os.environ.get("OAT_TAPE", "1") != "0"
                        else "eager (one launch per kernel)"),
    }
    # host_enqueue_ms_per_step: host time to issue ONE step onto an idle queue (best of 3, outside the timed region).
    # The per-step host time inside the timed loop (host_loop_ms_per_step) is NOT the enqueue cost: HIP lets the host run
    # only a few thousand launches ahead, so a host that issues a step in 7 ms spends the rest of the 51 ms waiting.
    singles = []
    for _ in range(3):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        step()
        singles.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    out["host_enqueue_ms_per_step"] = round(min(singles) * 1e3, 2)
    # one more, instrumented, step for the roofline figure.  EVERY rank runs it (its collectives need all of them);
    # rank 0 reports
    engines = [m._engine for m in dp.module.modules() if hasattr(getattr(m, "_engine", None), "use_tape")]
    taped = [e.use_tape for e in engines]
    for e in engines:
        e.use_tape = False                            # launch by launch: the Python wrappers must see every GEMM
    by = instrumented_gemm_profile(eager_step)
    for e, t in zip(engines, taped):
        e.use_tape = t
    if rank == 0:
        if by:
            def entry(name, d):
                ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
                peak = FP8_DENSE_PEAK_TFLOPS if "fp8" in name else BF16_DENSE_PEAK_TFLOPS
                return {"kernel": name, "achieved": round(ach, 1), "frac": round(ach / peak, 4), "peak": peak,
                        "launches_per_step": d["n"], "avg_launch_us": round(d["ms"] / d["n"] * 1e3, 1),
                        "ms_per_step": round(d["ms"], 2), "gflop_per_launch": round(d["flops"] / d["n"] / 1e9, 1)}
            ranked = sorted(by.items(), key=lambda kv: -kv[1]["ms"])
            if args.dtype == "fp8":             # config 5 is quoted on the fp8 kernel: report it first, the bf16 ones beside it
                ranked.sort(key=lambda kv: ("fp8" not in kv[0], -kv[1]["ms"]))
            top = entry(*ranked[0])
            # `traffic` (HBM bytes per launch) needs the PMC counters, which cannot be sampled from inside this process:
            # a 1-GPU run fills it below from two rocprofv3 --pmc passes over a child process (hbm_traffic)
            out["roofline"] = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["achieved"],
                               "peak": top["peak"], "unit": "TFLOP/s", "frac": top["frac"], "traffic": None,
                               "launches_per_step": top["launches_per_step"], "avg_launch_us": top["avg_launch_us"],
                               "ms_per_step": top["ms_per_step"], "gflop_per_launch": top["gflop_per_launch"],
                               "traffic_profile": "profiles/ (rocprofv3 --pmc passes of this command, per round)",
                               "other_gemm_kernels": [entry(n, d) for n, d in ranked[1:] if d["ms"] > 0.2]}
        if world == 1 and by and not args.no_traffic and args.variant == "frozen":
            tr = hbm_traffic(args, out["roofline"]["kernel"], _ALGO_BYTES.get(out["roofline"]["kernel"], [])) \
                if _budget_left() > 2 * PMC_PASS_TIMEOUT_S + 120 else {"skipped": "wall-time budget of the optional legs used up"}
            out["roofline"]["traffic"] = None if "skipped" in tr else tr
            if "skipped" in tr:
                out["roofline"]["traffic_skipped"] = tr["skipped"]
        if world == 1 and not args.no_forced_w1 and args.variant == "frozen" and _budget_left() > 120:
            try:
                out["w1_forced"] = forced_w1(dp, eager_step, args.batch, device)
                out["ms_per_step_w1_forced"] = out["w1_forced"]["ms_per_step_w1_forced"]
            except Exception as exc:                  # the headline line must still be printed
                out["w1_forced"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_other_configs and args.variant == "frozen" and \
                (args.other_configs or (args.frames == 8 and args.res == 224)):
            # config 3 as worded (object regions on): outside the timed region and outside `value`
            import gc
            del dp, opt, data, step, eager_step, engines, by
            gc.collect()
            torch.cuda.empty_cache()
            out["other_configs"] = []
            for variant, kw in other_config_plan(args):
                if _budget_left() < 90:
                    out["other_configs"].append({"workload": f"[{kw.get('label', variant)}]", "skipped": "wall-time budget of the optional legs used up"})
                    continue
                try:
                    out["other_configs"].append(other_config_line(args, variant, device, **kw))
                except Exception as exc:              # the headline line must still be printed
                    out["other_configs"].append({"workload": f"[{kw.get('label', variant)}]", "error": f"{type(exc).__name__}: {exc}"})
                    gc.collect()
                    torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline and args.variant == "frozen":
            out["cpu_baseline"] = cpu_baseline(args.frames)        # bounded by construction (about 25 s): always reported
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
