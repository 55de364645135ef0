#!/usr/bin/env python
"""Headline benchmark: volumes/sec of the CoMA-UNet training step at 128^3, batch 2 per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = forward + GenerativeContrastiveLoss + backward + gradient all-reduce (N > 1) + fused
AdamW on one batch of seeded synthetic volumes already resident in HBM.
Rank 0 prints ONE JSON line.
 * `roofline`: the dominant KERNEL of the step by summed time (by name, as rocprofv3 prints it -- the library reports the
   variant each dispatch launched, coma_last_kernel; normalisation launches are bracketed too): achieved = summed
   algorithmic FLOPs (MFMA-bound) or bytes (HBM-bound) of its launches / summed launch durations, measured live with HIP
   events on the launching stream; `step` carries the step-level figure (U-Net FLOPs x volumes / step time / peak).
   `roofline_mfma` / `roofline_hbm`: the top kernel of EACH regime, so the line always names the dominant convolution and
   the dominant bandwidth kernel.  `traffic` = HBM bytes per launch from the committed PMC passes of the same build
   (profiles/r03_pmc_traffic.json; FETCH_SIZE doubled as the gfx950 note in MI355X_MICROARCH.md prescribes), null if absent.
 * `secondary` (N = 1): the eager `train_step` (what a caller without graph capture gets), the step WITH host->device
   input staging (train.InputStager: pinned host batch -> side-stream copy -> replay), the fp32 mode at 128^3 (the
   reference's own arithmetic: the mode that meets north_star's 1e-3) with its roofline, and BASELINE config C5
   (192x224x192, fp32, batch 1) -- a few seconds each.
 * `cpu_baseline`: the CPU oracle (this repo's restatement of the reference path -- the reference itself cannot be
   imported) timed on the host cores: 1 warm-up + median of 3 fwd+bwd steps, rank 0, N = 1 only.
 * `parity`: rel-L2 and voxel MAE of THIS run's model (same weights, same inputs) against that oracle's forward.
"""
import argparse
import json
import os
import sys
import time

if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("COMA_BENCH_ONE_DEVICE") != "1":
    # N > 1: the exchange's streams (the auxiliary stream behind the graph's bucket events, the process group's own) should
    # not share a hardware queue with the replayed graph's streams, or their work is only SEEN when the graph's packets in
    # front of it have drained (profiles/external_event_probe.py).  The runtime's default is 4 queues; the one-GPU step is
    # insensitive to the setting (17.08 / 17.07 / 17.04 ms at 4 / 8 / 6).  Must be set before HIP initialises.
    # One process per GPU only: TWO processes on one GPU (the gloo rehearsal) oversubscribe the device's hardware queues
    # from 2 x 6 on and the scheduler then time-slices them (replayed step 13 -> 72 / 140 / 160 ms at 6 / 7 / 8).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
FP32_VALU_PEAK_TFLOPS = 157.3     # = the fp32 MFMA peak (v_mfma_f32_32x32x2_f32 runs at the vector rate)
HBM_PEAK_GBPS = 8000.0
UNET_TFLOP_PER_VOLUME_128 = 2.340  # BASELINE.md section 2 (U-Net fwd+bwd, 128^3)


def cpu_baseline(size, threads, batch=2, gpu_model_factory=None):
    """CPU oracle fwd+bwd at `size`^3 on the same per-GPU batch: 1 warm-up + median of 3 steps.
    With `gpu_model_factory` the GPU model is loaded with the oracle's weights and run on the same inputs: -> parity."""
    from oracle.coma_oracle import build_reference_model
    from oracle.criterions_oracle import build_reference_criterion, train_step_loss
    from coma_unet_amd.synthetic import make_batch
    torch.set_num_threads(threads)
    S = (size,) * 3
    torch.manual_seed(0)
    m = build_reference_model(volume_shape=S, double_forward=False)
    m.set_save_attn(None)
    m.train(True)
    b = make_batch(batch, S, seed=0)
    crit = build_reference_criterion()
    sd = {k: v.clone() for k, v in m.state_dict().items()}     # before any step moves the BatchNorm running buffers
    times, out0 = [], None
    for it in range(4):
        m.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        out = m(b["mri"], b["covars"], roi_pred_dicts=b["roi_pred_dicts"], sample_roi_mask=b["roi"])
        train_step_loss(out, b["tau"], b["roi"], b["covars"], crit)[0].backward()
        times.append(time.perf_counter() - t0)
        if out0 is None:
            out0 = out[0].detach().clone()
    dt = sorted(times[1:])[1]
    base = {"value": batch / dt, "unit": "volumes/s", "cores": threads, "kind": "port",
            "sample": f"CPU oracle (torch {torch.__version__}, fp32, single U-Net pass), fwd+bwd of batch {batch} at {size}^3: "
                      f"1 warm-up + median of 3 steps = {dt:.1f} s ({', '.join(f'{t:.1f}' for t in times)})"}
    parity = None
    if gpu_model_factory is not None:
        gm = gpu_model_factory()
        gm.load_state_dict(sd, strict=True)
        gm.train(True)
        with torch.no_grad():
            go = gm(b["mri"].cuda(), b["covars"].cuda(), roi_pred_dicts=b["roi_pred_dicts"], sample_roi_mask=b["roi"].cuda())
        g = go[0].float().cpu()
        parity = {"rel_l2": float((g - out0).norm() / out0.norm()), "voxel_mae": float((g - out0).abs().mean()),
                  "against": f"CPU oracle fp32 forward, same weights and inputs, batch {batch} at {size}^3 (oracle parity unpinned: DESIGN.md section 3)"}
        del gm
    return base, parity


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 --pmc passes (profiles/r03_pmc_traffic.json,
    written by profiles/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command)."""
    path = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        tab = json.load(open(path))
    except Exception:
        return None
    ent = tab.get(kernel_name)
    if ent is not None:
        return float(ent["hbm_bytes_per_launch"])
    if " + " in kernel_name:      # a bracket that covers two launches (normalisation backward: partial sums + apply)
        dt = "__bf16" if "__bf16" in kernel_name else "float"
        tot = 0.0
        for part in kernel_name.split(" + "):
            base = part.split("<")[0].strip()
            cands = [v for k, v in tab.items() if k.startswith(base + "<") and dt in k]
            if not cands:
                return None
            tot += max(float(c["hbm_bytes_per_launch"]) for c in cands)      # (the widest-vector instantiation carries the large tensors)
        return tot
    return None


def rooflines(byk, timer_steps, peak_tf, step_roof, timer_note):
    """(roofline, roofline_mfma, roofline_hbm) from KernelTimer.by_kernel(): per kernel name (launches, ms, flops, bytes)."""
    def entry(kname, n, ms, fl, by, bound):
        common = {"kernel": kname, "launches_per_step": n / timer_steps, "avg_launch_us": round(ms / n * 1e3, 2),
                  "ms_per_step": round(ms / timer_steps, 3), "alg_flops_per_launch": round(fl / n),
                  "alg_bytes_per_launch": round(by / n), "traffic": pmc_traffic(kname)}
        if bound == "hbm":
            ach = by / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBPS, 4), **common}
        ach = fl / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": round(ach, 2), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(ach / peak_tf, 4), **common}

    def bound_of(fl, by):      # below the MFMA/HBM ridge (~300 FLOP/B) by 2x: a bandwidth kernel
        return "hbm" if fl / max(by, 1.0) < 150.0 else "mfma"
    top = sorted(byk.items(), key=lambda kv: -kv[1][1])
    if not top:
        return None, None, None
    ents = [(k, v, bound_of(v[2], v[3])) for k, v in top]
    first = lambda b: next((entry(k, *v, b) for k, v, bb in ents if bb == b), None)
    k0, v0, b0 = ents[0]
    roof = entry(k0, *v0, b0)
    roof.update({"step": step_roof, "measured": timer_note,
                 "top_kernels": [{"kernel": k, "bound": bb, "launches_per_step": v[0] / timer_steps, "ms_per_step": round(v[1] / timer_steps, 3),
                                  "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1), "alg_GBps": round(v[3] / (v[1] * 1e-3) / 1e9)}
                                 for k, v, bb in ents[:10]]})
    return roof, first("mfma"), first("hbm")


def family_roofline(byk, timer_steps, peak_tf, prefix):
    """One roofline entry for ALL instantiations of a kernel template (e.g. conv_mfma_duo_k<0> and <1>, the same kernel
    without / with the fused statistics), named as the ' + '-joined list of the kernels it sums over."""
    sel = sorted((k, v) for k, v in byk.items() if k.startswith(prefix))
    if not sel:
        return None
    n = sum(v[0] for _k, v in sel); ms = sum(v[1] for _k, v in sel); fl = sum(v[2] for _k, v in sel); by = sum(v[3] for _k, v in sel)
    name = " + ".join(k for k, _v in sel)
    ach = fl / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(ach / peak_tf, 4),
            "kernel": name, "launches_per_step": n / timer_steps, "avg_launch_us": round(ms / n * 1e3, 2),
            "ms_per_step": round(ms / timer_steps, 3), "alg_flops_per_launch": round(fl / n), "alg_bytes_per_launch": round(by / n),
            "traffic": (lambda t: None if any(x is None for x in t) else sum(x * v[0] for x, (_k, v) in zip(t, sel)) / n)(
                [pmc_traffic(k) for k, _v in sel]),       # launch-weighted mean of the members' PMC bytes per launch
            "per_kernel": {k: {"launches_per_step": v[0] / timer_steps, "ms_per_step": round(v[1] / timer_steps, 3),
                               "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)} for k, v in sel}}


def timed_steps(fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def secondary_figures(args, dev, headline_ms, eager_ctx):
    """Driver-visible numbers beside the headline (N = 1 only; a few seconds each) -- see the module docstring."""
    import coma_unet_amd as cu
    from coma_unet_amd import ops
    from coma_unet_amd.synthetic import make_batch
    from coma_unet_amd.train import train_step, make_optimizer, GraphedTrainStep, InputStager
    out = {}
    model, crit, opt, batch, step_fn = eager_ctx
    # (1) the eager step of the benched model (what train_dp runs before it has captured, and any caller without capture)
    ops.KernelTimer.enabled = False
    for _ in range(2):
        train_step(model, crit, opt, batch, None)
    out["eager_train_step"] = {"ms_per_step": round(timed_steps(lambda: train_step(model, crit, opt, batch, None), 5), 3),
                               "note": "train.train_step, no graph capture: host launch bound"}
    # (2) the replayed step fed from pinned host memory through train.InputStager (copy of the next batch beside the replay)
    if step_fn is not None:
        host = {k: (v.detach().cpu().pin_memory() if torch.is_tensor(v) else v) for k, v in batch.items()}
        stager = InputStager(step_fn)
        stager.submit(host)

        def staged():
            r = stager.run()
            stager.submit(host)
            return r
        for _ in range(3):
            staged()
        ms = timed_steps(staged, 10)
        stager.run()
        out["staged_input_step"] = {"ms_per_step": round(ms, 3), "volumes_per_s": round(args.batch / ms * 1e3, 2),
                                    "vs_resident_inputs": round(ms / headline_ms, 4),
                                    "note": "hipGraph replay + host->device copy of the next batch (3 volumes + covariates + priors, "
                                            "pinned) on a side stream + device-to-device hand-over"}
    return out


def secondary_config(size_dhw, dtype_name, batch, dev, steps=3, with_roofline=False):
    """One more configuration through the graphed step: ms/step (+ the per-kernel roofline of 1 eager step)."""
    import coma_unet_amd as cu
    from coma_unet_amd import ops
    from coma_unet_amd.synthetic import make_batch
    from coma_unet_amd.train import train_step, make_optimizer, GraphedTrainStep
    dt = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    torch.manual_seed(0)
    model = cu.build_model(volume_shape=size_dhw, compute_dtype=dt, static_prompts=True).to(dev)
    model.set_save_attn(None)
    model.train(True)
    crit = cu.build_reference_criterion(dev)
    opt = make_optimizer(model, 1e-3)
    b = make_batch(batch, size_dhw, seed=2000)
    gb = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    gb["roi_pred_dicts"] = model._priors(b["roi_pred_dicts"], batch, dev)
    step_fn = GraphedTrainStep(model, crit, opt, gb, warmup=2)
    step_fn()
    ms = timed_steps(lambda: step_fn(), steps)
    vox = size_dhw[0] * size_dhw[1] * size_dhw[2]
    unet_tf = UNET_TFLOP_PER_VOLUME_128 * vox / 128.0 ** 3
    peak = MFMA_BF16_PEAK_TFLOPS if dtype_name == "bf16" else FP32_VALU_PEAK_TFLOPS
    res = {"ms_per_step": round(ms, 3), "volumes_per_s": round(batch / ms * 1e3, 3), "dtype": dtype_name, "batch": batch,
           "volume": list(size_dhw), "loss": round(float(step_fn.losses[0]), 4),
           "step": {"unet_tflop_per_volume": round(unet_tf, 4), "achieved_tflops": round(batch / ms * 1e3 * unet_tf, 2), "peak": peak,
                    "frac": round(batch / ms * 1e3 * unet_tf / peak, 4)}}
    if with_roofline:
        ops.KernelTimer.enabled = True
        ops.KernelTimer.records = []
        train_step(model, crit, opt, gb, None)
        torch.cuda.synchronize()
        ops.KernelTimer.enabled = False
        roof, rm, rh = rooflines(ops.KernelTimer.by_kernel(), 1, peak, res["step"], "HIP events around every conv / norm launch of 1 eager step")
        res["roofline"], res["roofline_mfma"], res["roofline_hbm"] = roof, rm, rh
        ops.KernelTimer.records = []
    del step_fn, model, opt, crit
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=2, help="volumes per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (eager step, staged inputs, fp32, C5)")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly even on one GPU")
    ap.add_argument("--dp", default=os.environ.get("COMA_DP", "torch"), choices=["torch", "capi", "capi-sharded"],
                    help="N > 1 gradient exchange: torch = torch.distributed (RCCL) all-reduce after the graph-replayed fwd+bwd; "
                         "capi = the C-ABI RCCL path (coma_allreduce_sum_f32 on a side stream, launched from the gradient sink "
                         "while backward runs, captured INSIDE the step graph); capi-sharded = reduce-scatter + AdamW on 1/N + all-gather")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("COMA_BENCH_ONE_DEVICE"):      # rehearsal of the N > 1 path on a 1-GPU box (all ranks on cuda:0)
        local = 0
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("COMA_DIST_BACKEND", "nccl")      # "gloo" only for rehearsing N > 1 on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import coma_unet_amd as cu
    from coma_unet_amd import ops
    from coma_unet_amd.synthetic import make_batch
    from coma_unet_amd.train import train_step, make_optimizer, GraphedTrainStep
    from coma_unet_amd.data_parallel import GradReducer, StreamedGradExchange, broadcast_module

    S = (args.size,) * 3
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model = cu.build_model(volume_shape=S, compute_dtype=dt, static_prompts=True).to(dev)
    model.set_save_attn(None)
    model.train(True)
    if world > 1:
        broadcast_module(model)
    crit = cu.build_reference_criterion(dev)
    opt = make_optimizer(model, 1e-3, pad_to=world if args.dp == "capi-sharded" else 1)

    def make_reducer():
        if world == 1:
            return None
        if args.dp == "torch":
            return GradReducer(opt)
        if os.environ.get("COMA_DIST_BACKEND", "nccl") != "nccl":
            # rehearsal of the capi modes without RCCL (several ranks on one GPU / gloo): the torch.distributed stand-in with
            # the communicator's interface -- same bucket / shard logic, collectives not capturable (the step runs eagerly)
            from coma_unet_amd.data_parallel import TorchComm
            return StreamedGradExchange(opt, TorchComm(), sharded=args.dp == "capi-sharded")
        from coma_unet_amd.rccl_comm import RcclComm
        return StreamedGradExchange(opt, RcclComm(device=dev), sharded=args.dp == "capi-sharded")
    reducer = make_reducer()
    if world > 1 and args.dp != "torch" and os.environ.get("COMA_DIST_BACKEND", "nccl") != "nccl":
        args.no_graph = True
    b = make_batch(args.batch, S, seed=1000 + rank)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    batch["roi_pred_dicts"] = model._priors(b["roi_pred_dicts"], args.batch, dev)   # (B,36,2) resident table

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # The step is replayed from a captured hipGraph (same kernels, no host launch gaps): one graph for the whole
    # step on one GPU; with N > 1 the graph holds forward + backward and the bucketed RCCL all-reduce + AdamW
    # follow it.  --no-graph runs the eager step (N > 1: all-reduce issued from backward hooks, overlapped).
    use_graph = not args.no_graph
    graph_note = None
    if use_graph:
        try:
            step_fn = GraphedTrainStep(model, crit, opt, batch, warmup=max(args.warmup, 2), reducer=reducer)
            run_step = lambda: step_fn()[0]
        except Exception as e:      # never lose the measurement to a capture problem: fall back to the eager step
            graph_note = f"graph capture failed ({type(e).__name__}: {e}); eager step used"
            print(graph_note, file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            use_graph = False
            args.dp = "torch"                      # the plain, rehearsed path
            reducer = make_reducer()
    if not use_graph:
        for _ in range(args.warmup):
            train_step(model, crit, opt, batch, reducer)
        run_step = lambda: train_step(model, crit, opt, batch, reducer)[0]
        ops.KernelTimer.enabled = not args.no_kernel_timer
        ops.KernelTimer.records = []
    sync_all()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = run_step()
    sync_all()
    elapsed = time.perf_counter() - t0
    ops.KernelTimer.enabled = False
    timer_note = "HIP events around every conv launch of the timed steps"
    if use_graph and not args.no_kernel_timer:
        # kernels inside a graph replay cannot be bracketed individually: time the SAME kernels in two eager steps
        ops.KernelTimer.enabled = True
        ops.KernelTimer.records = []
        for _ in range(2):
            train_step(model, crit, opt, batch, None)
        torch.cuda.synchronize()
        ops.KernelTimer.enabled = False
        timer_note = "HIP events around every conv launch of 2 eager steps run right after the graph-replayed timed region"
    timer_steps = 2 if use_graph else args.steps
    dp_info = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        # how much of the step is the gradient exchange, and how much of the exchange is exposed:
        #   exchange_ms  one exchange of the whole flat gradient buffer with nothing beside it (the mode's own collectives)
        #   local_ms     the same step without any exchange (each rank alone; run AFTER the timed region: the replicas may drift)
        #   exposed_ms   ms_per_step - local_ms: what the exchange adds to the step after overlap
        try:
            sync_all()
            def exchange_once():
                if args.dp == "torch":
                    reducer.reduce_flat()
                elif args.dp == "capi":
                    reducer.comm.all_reduce_(opt.flat_g)
                else:
                    n = opt.flat_g.numel() // world
                    reducer.comm.reduce_scatter(opt.flat_g, opt.flat_g[reducer.comm.rank * n:(reducer.comm.rank + 1) * n])
                    reducer.comm.all_gather(opt.flat_p[reducer.comm.rank * n:(reducer.comm.rank + 1) * n], opt.flat_p)
            exchange_once()
            sync_all()
            t0x = time.perf_counter()
            for _ in range(3):
                exchange_once()
            sync_all()
            exchange_ms = (time.perf_counter() - t0x) / 3 * 1e3
            if use_graph:
                local = GraphedTrainStep(model, crit, opt, batch, warmup=2, reducer=None)
                local_fn = lambda: local()
            else:
                local_fn = lambda: train_step(model, crit, opt, batch, None)
            local_fn()
            sync_all()
            t0l = time.perf_counter()
            for _ in range(5):
                local_fn()
            sync_all()
            local_ms = (time.perf_counter() - t0l) / 5 * 1e3
            tt = torch.tensor([exchange_ms, local_ms], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dp_info = {"mode": args.dp, "exchange_ms": round(float(tt[0]), 3), "local_step_ms": round(float(tt[1]), 3),
                       "exposed_ms": round(elapsed / args.steps * 1e3 - float(tt[1]), 3),
                       "gradient_mb": round(opt.flat_g.numel() * 4 / 2 ** 20, 1),
                       "early_buckets": getattr(reducer, "early", None), "early_mb": getattr(reducer, "early_mb", None),
                       # torch mode under graph replay: were buckets sent from behind external events of the graph
                       # (data_parallel.GraphBucketWatch)?  "node" = yes; anything else = exchange behind the graph, and why
                       "graph_overlap": getattr(reducer, "watch_probe", None)}
        except Exception as e:
            dp_info = {"mode": args.dp, "error": f"{type(e).__name__}: {e}"}
    loss = float(last[0])

    if rank == 0:
        vols = args.batch * world * args.steps
        value = vols / elapsed
        summ = ops.KernelTimer.summary()
        byk = ops.KernelTimer.by_kernel()
        roof = None
        kernels = {}
        unet_tf = UNET_TFLOP_PER_VOLUME_128 * (args.size / 128.0) ** 3
        peak_tf = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else FP32_VALU_PEAK_TFLOPS
        step_ach = value * unet_tf
        step_roof = {"unet_tflop_per_volume": round(unet_tf, 4), "achieved_tflops": round(step_ach, 2), "peak": peak_tf,
                     "frac": round(step_ach / peak_tf, 4)}
        roof_mfma = roof_hbm = roof_thick = None
        if summ:
            for (kind, algo), (n, ms, fl, by) in sorted(summ.items()):
                kernels[f"{kind}/{algo}"] = {"launches": n, "ms_total": round(ms, 3),
                                             "tflops": round(fl / (ms * 1e-3) / 1e12, 2) if ms > 0 else None,
                                             "alg_GBps": round(by / (ms * 1e-3) / 1e9, 1) if ms > 0 else None}
            roof, roof_mfma, roof_hbm = rooflines(byk, timer_steps, peak_tf, step_roof, timer_note)
            # the thick stride-1 forward / data-gradient kernel is two instantiations of one template (without / with the
            # fused statistics): reported once more as a family, next to the per-name entries above
            roof_thick = family_roofline(byk, timer_steps, peak_tf, "conv_mfma_duo_k" if args.dtype == "bf16" else "conv_mfma_halo2_k")
        line = {
            "metric": f"volumes/sec (train fwd+bwd) at 128^3 {args.dtype}", "value": round(value, 4), "unit": "volumes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"CoMA-UNet train step (fwd + RoiMSE/RnC loss + bwd + all-reduce + AdamW), "
                                   f"{args.size}^3 volumes, batch {args.batch}/GPU, 6-dim covariates (BASELINE configs[3])",
                       "global_batch": args.batch * world, "volume": list(S), "parallelism": f"dp{world}",
                       "launch": ("hipGraph replay" + ((" of fwd+bwd, then bucketed RCCL all-reduce + AdamW" if args.dp == "torch" else
                                                        f" of the whole step incl. the overlapped C-ABI RCCL exchange ({args.dp})")
                                                       if world > 1 else " of the whole step"))
                                 if use_graph else "eager (all-reduce overlapped with backward)",
                       "params_M": round(sum(p.numel() for p in model.parameters()) / 1e6, 1)},
            "loss": round(loss, 4), "note": graph_note,
            "unet_tflops_per_s": round(step_ach, 2),
            "roofline": roof, "roofline_mfma": roof_mfma, "roofline_hbm": roof_hbm, "roofline_thick_conv": roof_thick,
            "conv_kernels": kernels,
            "zero_arena_mb": {str(k): round(a.peak / 2 ** 20, 1) for k, a in ops.ZeroArena._arenas.items()},
            "data_parallel": dp_info,
        }
        if world == 1 and not args.no_secondary:
            try:
                line["secondary"] = secondary_figures(args, dev, elapsed / args.steps * 1e3,
                                                      (model, crit, opt, batch, step_fn if use_graph else None))
            except Exception as e:      # never lose the headline to a secondary measurement
                line["secondary"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                threads = len(os.sched_getaffinity(0))
            except Exception:
                threads = os.cpu_count() or 1
            threads = min(threads, 16)   # a 1-GPU box's CPU share is 16 cores; more threads only oversubscribe
            del model, opt, crit
            if use_graph:
                del step_fn, run_step
            torch.cuda.empty_cache()
            if not args.no_secondary and isinstance(line.get("secondary"), dict) and "error" not in line["secondary"]:
                for key, cfg in (("fp32_128", ((args.size,) * 3, "fp32", args.batch, 3, True)),
                                 ("c5_fullres_fp32", ((192, 224, 192), "fp32", 1, 2, False))):
                    if args.size != 128 and key == "c5_fullres_fp32":
                        continue
                    try:
                        line["secondary"][key] = secondary_config(cfg[0], cfg[1], cfg[2], dev, steps=cfg[3], with_roofline=cfg[4])
                    except Exception as e:
                        line["secondary"][key] = {"error": f"{type(e).__name__}: {e}"}
            factory = lambda: (lambda mm: (mm.set_save_attn(None), mm)[1])(cu.build_model(volume_shape=S, compute_dtype=dt).to(dev))
            line["cpu_baseline"], line["parity"] = cpu_baseline(args.size, threads, args.batch, factory)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
