"""bench.py — denoise-step throughput of the quantized SD1.4 / SDXL UNet on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c4|c5]

N > 1 without a torchrun environment: the parent process — before anything touches the GPU — checks that N devices
are visible (fails loudly otherwise) and starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py <same flags>` as a CHILD process (never an exec of a process that
holds the GPU), then exits with its code.  Under torchrun every rank asserts WORLD_SIZE == --gpus.  This is the
reference's only multi-GPU mode (one python process per GPU, rank-sliced prompt list: src/gen4eval_SDXL.py:116,
scripts/gen4eval_SDXL.sh:53-100): ranks share nothing but the timing barrier and a MAX all-reduce of the elapsed time.

Metric (BASELINE.json): UNet denoise steps/sec @ SD1.4 512² W4A8 g16 (config c2, the default).  A "step" = one
QuantModel.forward on a CFG pair ([2,4,64,64] latents, [2,77,768] context) inside the DDIM loop (CFG combine +
scheduler update included), synthetic name-keyed weights and a synthetic reference-format cali_ckpt (time-aware act
tables, 16 DGQ groups, log2-quantised softmax with real-time δ and start-peak — the reference's own preset for G>1,
scripts/quantize_act.sh:16-19).  Inputs are resident in HBM when the timed region starts.
  c4 = SDXL-turbo W4A8 g16, 1024² (128² latents), batch 1, 4-step schedule, no CFG (BASELINE.json configs[3])
  c5 = SDXL-turbo W4A6 g=1 (scalar scales, uniform softmax quantiser), 8 prompts per GPU (configs[4]: 64 prompts / 8 GPUs)

Timing: W untimed warm-up steps, then `--windows` (default 5) windows of EXACTLY K steps, each bracketed by a barrier +
torch.cuda.synchronize() on both sides, max over ranks per window; `value` / `ms_per_step` come from the MEDIAN window,
the minimum and every window are printed beside it.

The JSON line also carries
  roofline     — the W4A8 MFMA GEMM path (dgq_gemm_wxa8): algorithmic int8 ops of all quantized layers of one step
                 (2·M·N·K, SURVEY.md §8(d): 1.354 Top for c2) ÷ the HIP-event time of those launches, against the
                 dense int8 MFMA peak (≈5 Pop/s = 2× the 2.5 PF bf16 peak, MI355X_MICROARCH.md § Matrix cores);
  cpu_baseline — the reference's op sequence (oracle/, a pinned CPU port: re-quantised weights each call, F.unfold,
                 fp32 GEMMs, materialised attention) on this box's host cores: 1 warm-up + 3 timed steps (c2 only).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

INT8_PEAK_TOPS = 5000.0

CONFIGS = {
    # SD v1.4 W4A8 g=16, DDIM-50 schedule, CFG pair per prompt
    "c2": dict(arch="sd", res=64, cfg=dict(wbits=4, abits=8, use_aq=True, G=16, log=True, rt=True, sp=True, time_aware=True, steps=50),
               guidance=7.5, prompts=1,
               metric="UNet denoise steps/sec @ SD1.4 512^2 W4A8 g16",
               workload="SD v1.4 UNet W4A8 g=16 (time-aware, log2 softmax real-time δ, start-peak), DDIM 50-step schedule, "
                        "512x512 (64x64 latents), CFG pair per step per GPU"),
    # SD v1.4 W4A6 g=8, the reference's preset for G > 1 (scripts/quantize_act.sh:16-19: log2 softmax quantiser, real-time δ, start-peak,
    # time-aware) — BASELINE.json configs[2]
    "c3": dict(arch="sd", res=64, cfg=dict(wbits=4, abits=6, use_aq=True, G=8, log=True, rt=True, sp=True, time_aware=True, steps=50),
               guidance=7.5, prompts=1,
               metric="UNet denoise steps/sec @ SD1.4 512^2 W4A6 g8",
               workload="SD v1.4 UNet W4A6 g=8 + t2i_log_quant (real-time δ, start-peak) + time_aware_aqtizer, DDIM 50-step schedule, "
                        "512x512 (64x64 latents), CFG pair per step per GPU"),
    # SDXL-turbo W4A8 g=16, 4 steps, 1024x1024, batch 1, no CFG
    "c4": dict(arch="sdxl", res=128, cfg=dict(wbits=4, abits=8, use_aq=True, G=16, log=True, rt=True, sp=True, time_aware=True, steps=4),
               guidance=0.0, prompts=1,
               metric="UNet denoise steps/sec @ SDXL-turbo 1024^2 W4A8 g16",
               workload="SDXL-turbo UNet W4A8 g=16 (time-aware, log2 softmax real-time δ, start-peak), 4-step EulerAncestral (trailing) schedule, "
                        "1024x1024 (128x128 latents), batch = prompts per GPU, no CFG"),
    # SDXL-turbo W4A6 g=1 (scalar scales), prompts sharded over the GPUs (64 prompts over 8 GPUs = 8 per GPU)
    "c5": dict(arch="sdxl", res=128, cfg=dict(wbits=4, abits=6, use_aq=True, G=1, log=False, rt=False, sp=False, time_aware=True, steps=4),
               guidance=0.0, prompts=8,
               metric="UNet denoise steps/sec @ SDXL-turbo 1024^2 W4A6 g1, prompts sharded over GPUs",
               workload="SDXL-turbo UNet W4A6 g=1 (scalar activation scales, uniform softmax quantiser, time-aware), 4-step "
                        "EulerAncestral (trailing) schedule, 1024x1024 (128x128 latents), batch = prompts per GPU, no CFG"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each (median reported)")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", action="store_true", help="force the CPU leg for c4/c5 too (minutes per SDXL step)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per timestep slot")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-launch GEMM replay pass (profiling runs)")
    ap.add_argument("--prompts-per-gpu", type=int, default=0,
                    help="prompts denoised together on each GPU (default: the config's own: c2/c4 1, c5 8)")
    ap.add_argument("--dev-no-guidance", action="store_true",
                    help="development: run the SD configs without the CFG pair (batch 1 per prompt) — the line says so in "
                         "config.workload / config.dev_override and is never the headline")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp16", "bf16"],
                    help="inter-layer activation dtype (fp32 = the reference's default .float() mode)")
    return ap.parse_args(argv)


def launch_command(args, argv, port):
    """The torchrun command the parent starts for --gpus N > 1 (one rank per GPU over RCCL)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(args, argv):
    """Parent side of --gpus N > 1.  Nothing here initialises the GPU (torch.cuda.device_count() does not, on this
    image); the ranks are child processes."""
    import torch
    visible = torch.cuda.device_count()
    if visible < args.gpus and not dry_run():
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) visible; refusing to report a smaller run as "
                         "n_gpus=%d\n" % (args.gpus, visible, args.gpus))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(launch_command(args, argv, port), env=env)


def dry_run():
    """DGQ_BENCH_BACKEND=gloo: the LAUNCHER REHEARSAL — the same rank path (torchrun child, rank-seeded prompts, rank-0 ckpt
    write + mmap read on the others, barriers, MAX all-reduce, one JSON line with n_gpus = N) on CPU ranks with the FP tiny
    UNet, because the quantized kernels need a GPU.  The line is marked ``dry_run`` and is not a measurement."""
    return os.environ.get("DGQ_BENCH_BACKEND", "nccl") == "gloo"


class _FpDryModel:
    """FP tiny UNet behind the few QuantModel methods main() calls (launcher rehearsal only)."""

    def __init__(self, unet, laps):
        self.unet, self._build_laps, self._graphs = unet.eval(), laps, None

    def prepare_slots(self, slots):
        pass

    def enable_graphs(self, on):
        pass

    def __call__(self, x, t, ctx, **kw):
        import torch
        return self.unet(x, torch.tensor(int(t)), encoder_hidden_states=ctx)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        return 2
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args, argv)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: WORLD_SIZE=%d but --gpus %d; launch as `python bench.py --gpus N` or with "
                         "--nproc-per-node equal to --gpus\n" % (world, args.gpus))
        return 2
    dist = None
    DRY = dry_run()
    if not DRY:
        torch.cuda.set_device(local_rank)
    # host-side work of the load (planning, packing tables, the synthetic ckpt): a GPU box shows every host thread but grants a
    # share of the cores — torch's default oversubscribes it (synthetic build 35.6 s against 25.1 s with 16 threads)
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    if "RANK" in os.environ:          # launched by torch.distributed.run: one rank per GPU over RCCL (timing only)
        import torch.distributed as dist
        if DRY:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group(os.environ.get("DGQ_BENCH_BACKEND", "nccl"), device_id=torch.device("cuda", local_rank))
    dev = torch.device("cpu") if DRY else torch.device("cuda", local_rank)

    from dgq_amd import ops, synth
    from dgq_amd.runtime import build_synthetic_qnn, synthetic_fp_unet, DDIMScheduler
    from dgq_amd._lib import require_gpu
    if not DRY:
        require_gpu()

    def sync():
        if not DRY:
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    C = CONFIGS[args.config]
    arch, res, qcfg, guidance = C["arch"], C["res"], C["cfg"], C["guidance"]
    if args.dev_no_guidance:                                 # development: the conditional half alone (batch 1); labelled in the line below
        guidance = 0.0
    if DRY:
        arch, res, guidance = "tiny", 16, 7.5
        args.no_graph = args.no_roofline = args.no_cpu_baseline = True
    K, W = args.steps, args.warmup
    P = args.prompts_per_gpu or C["prompts"]
    nsteps = qcfg["steps"]
    sch = DDIMScheduler(nsteps)
    sched_ts = sch.timesteps if arch in ("sd", "tiny") else [999, 749, 499, 249]     # SDXL-turbo: src/inference_qmodel.py:49-54 trailing spacing
    n_ts = min(K + W, len(sched_ts))
    timesteps = [sched_ts[i % n_ts] for i in range(W + K)]
    slots = sorted({(1000 - t) // (1000 // nsteps) for t in timesteps})
    batch = (2 if guidance > 0 else 1) * P
    t_model = time.perf_counter()
    if DRY:
        unet, ckpt_path, laps, _ = synthetic_fp_unet(arch, qcfg, res, batch, max(slots) + 1, ckpt_dir=os.environ.get("DGQ_BENCH_CKPT_DIR", "/tmp"),
                                                     rank=rank, barrier=barrier)
        qnn = _FpDryModel(unet, laps)
    else:
        qnn, ckpt_path = build_synthetic_qnn(arch, qcfg, res, batch, max(slots) + 1, rank=local_rank, barrier=barrier, device=dev)
    if args.dtype == "fp16":
        qnn.half()
    elif args.dtype == "bf16":
        qnn.to(torch.bfloat16)
    adt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    t_prep = time.perf_counter()
    qnn.prepare_slots(slots)
    sync()
    t_prep = time.perf_counter() - t_prep
    model_ready_s = time.perf_counter() - t_model
    if not args.no_graph:
        qnn.enable_graphs(True)

    # this rank's prompts (seeded by rank: rank-sliced prompt list), resident on the device
    ctx_dim = {"sd": 768, "sdxl": 2048, "tiny": 64}[arch]
    lat = synth.named_randn("latent", (P, 4, res, res), 1 + rank).to(dev, adt)
    ctx = synth.named_randn("ctx", (batch, 77, ctx_dim), 100 + rank).to(dev, adt)
    extra = {}
    if arch == "sdxl":
        te = synth.named_randn("text_embeds", (batch, 1280), 200 + rank).to(dev, adt)
        tid = torch.tensor([[float(res * 8)] * 2 + [0.0, 0.0] + [float(res * 8)] * 2]).repeat(batch, 1).to(dev, adt)
        extra = {"added_cond_kwargs": {"text_embeds": te, "time_ids": tid}}

    euler, noise = {}, None
    if arch == "sdxl":
        from dgq_amd.scheduler import EulerAncestralDiscreteScheduler
        es = EulerAncestralDiscreteScheduler(nsteps)
        for i, t in enumerate(int(v) for v in es.timesteps):
            s_from, s_to = float(es.sigmas[i]), float(es.sigmas[i + 1])
            s_up = (s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2) ** 0.5
            s_down = (s_to ** 2 - s_up ** 2) ** 0.5
            euler[t] = (s_from, s_down - s_from, s_up)    # x' = x + eps·(σ_down − σ) + noise·σ_up  (derivative = eps for ε-prediction)
        assert sorted(euler) == sorted(sched_ts), (sorted(euler), sched_ts)
        noise = synth.named_randn("euler_noise", (P, 4, res, res), 300 + rank).to(dev, adt)
        lat = (lat.float() * float(es.init_noise_sigma)).to(adt)              # the pipeline's initial latent scale

    def one_step(x, t):
        if arch == "sdxl":
            x_in = x * (1.0 / (euler[t][0] ** 2 + 1.0) ** 0.5)                  # scale_model_input
            eps = qnn(x_in.to(x.dtype), t, ctx, **extra)[0]
            sg, dsg, sup = euler[t]
            return ((x + eps * dsg) + noise * sup).to(x.dtype)
        if guidance > 0:
            eps = qnn(torch.cat([x, x], dim=0), t, ctx, **extra)[0]
            if arch in ("sd", "tiny") and hasattr(sch, "step_guided"):
                return sch.step_guided(eps, t, x, guidance)        # as dgq_amd/pipeline.py: guidance + DDIM update
            e_u, e_c = eps.chunk(2)
            eps = e_u + guidance * (e_c - e_u)
        else:
            eps = qnn(x, t, ctx, **extra)[0]
        if arch in ("sd", "tiny"):
            return sch.step(eps, t, x)
        raise AssertionError("unreachable: SDXL-turbo steps return above")

    windows = []
    with torch.no_grad():
        if not args.no_graph:
            for t in sorted(set(timesteps), reverse=True):     # capture one graph per slot used (amortised over images)
                one_step(lat, t)
        x = lat
        for t in timesteps[:W]:
            x = one_step(x, t)
        x_w = x
        for _ in range(max(1, args.windows)):
            x = x_w
            sync()
            barrier()
            sync()
            t0 = time.perf_counter()
            for t in timesteps[W:]:
                x = one_step(x, t)
            sync()
            barrier()
            sync()
            el = time.perf_counter() - t0
            if dist is not None:
                tt = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            windows.append(el)
    assert torch.isfinite(x).all()
    elapsed = statistics.median(windows)
    ready_per_rank = [round(model_ready_s, 1)]
    if dist is not None:                   # every rank's load time (they run concurrently; rank 0 also writes the ckpt)
        tt = torch.zeros(world, device=dev, dtype=torch.float64)
        tt[rank] = model_ready_s
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        ready_per_rank = [round(float(v), 1) for v in tt.tolist()]

    # ---- roofline of the dominant north-star kernel (W4A8 GEMM), measured live with HIP events ------------------
    # One eager step with the GEMM entry point wrapped: every launch (its split-K reduction included) is replayed REP
    # times back to back from a hipGraph on the current stream and bracketed by HIP events on that stream — eager
    # per-launch events would time the host-side launch gaps, not the kernels.  Sum over the step = the kernel time
    # rocprofv3 reports for the GEMM kernels (profiles/).
    roofline, roofline_extra = None, {}
    if rank == 0 and not args.no_roofline:
        roofline, roofline_extra = measure_gemm_roofline(torch, ops, qnn, lambda: one_step(lat, timesteps[W]), args)

    # ---- CPU baseline: the reference's op sequence (oracle port) on this box's host cores ----------------------------
    cpu_baseline = None
    want_cpu = (args.config == "c2" and not args.no_cpu_baseline) or args.cpu_baseline
    if rank == 0 and world == 1 and want_cpu:
        from oracle import dgq_oracle as orc
        ck = torch.load(ckpt_path, map_location="cpu")
        cfg = orc.OracleConfig(arch, qcfg["wbits"], qcfg["abits"], True, True, qcfg["abits"], qcfg["log"], qcfg["rt"], qcfg["sp"],
                               True, nsteps, qcfg["G"] > 1)
        om = orc.OracleModel(ck, cfg, synth.synth_state_dict(arch, 0))
        lat_c = synth.named_randn("latent", (1, 4, res, res), 1)
        ctx_c = synth.named_randn("ctx", (2 if guidance > 0 else 1, 77, ctx_dim), 100)
        okw = {}
        if arch == "sdxl":
            okw = dict(text_embeds=synth.named_randn("text_embeds", (1, 1280), 200),
                       time_ids=torch.tensor([[float(res * 8)] * 2 + [0.0, 0.0] + [float(res * 8)] * 2]))
        t = timesteps[W]
        xin = torch.cat([lat_c, lat_c]) if guidance > 0 else lat_c
        # Thread count: a one-GPU box shows every host thread but grants a share of the cores, and torch's default (all of them)
        # oversubscribes that share — measured 4x slower than 16 threads on the 16-core share.  One untimed-for-the-result probe
        # step per candidate (which doubles as the warm-up), the faster one is used and reported.
        cands = sorted({min(os.cpu_count() or 1, n) for n in (16, 64)})
        probe = {}
        for n_thr in cands:
            torch.set_num_threads(n_thr)
            tc0 = time.perf_counter()
            om.forward(xin, t, ctx_c, **okw)
            probe[n_thr] = time.perf_counter() - tc0
        ncores = min(probe, key=probe.get)
        torch.set_num_threads(ncores)
        n_cpu = 3 if args.config == "c2" else 1
        secs = []
        for _ in range(n_cpu):
            tc0 = time.perf_counter()
            om.forward(xin, t, ctx_c, **okw)
            secs.append(time.perf_counter() - tc0)
        cpu_s = statistics.median(secs)
        cpu_baseline = {"value": round(1.0 / cpu_s, 5), "unit": "steps/s", "cores": ncores, "kind": "port",
                        "seconds_per_step": [round(s, 2) for s in secs],
                        "thread_probe_seconds": {str(k): round(v, 2) for k, v in probe.items()},
                        "sample": "one probe step per thread-count candidate (also the warm-up), then %d timed UNet denoise steps "
                                  "(batch %d, t=%d) of the same workload, fp32, torch.set_num_threads(%d) = the faster candidate; "
                                  "median reported" % (n_cpu, xin.shape[0], t, ncores)}

    if rank == 0:
        n = max(world, 1)
        # kernel-class shares of the step: a STATIC record from the rocprofv3 kernel trace of this command
        # (tools/profile_step.sh + tools/kernel_classes.py), attached only to the configuration it was profiled on and
        # stamped with the commit it was measured at
        glue = None
        import glob
        for gj in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_step_kernel_classes.json")), reverse=True):
            gd = json.load(open(gj))
            if (gd.get("config") == args.config and gd.get("dtype") == args.dtype and gd.get("graph", True) == (not args.no_graph)
                    and (args.prompts_per_gpu or C["prompts"]) == C["prompts"]):
                glue = gd
                break
        import resource
        # anything that departs from the configuration's own definition is said in the line itself: the workload string
        # and a dev_override list
        workload_label, overrides = C["workload"], []
        if P != C["prompts"]:
            overrides.append("--prompts-per-gpu %d (the configuration runs %d): batch %d per GPU" % (P, C["prompts"], batch))
            workload_label = workload_label.replace("CFG pair per step per GPU", "%d prompts x CFG pair per step per GPU (NOT the headline batch)" % P)
        if args.dev_no_guidance and C["guidance"] > 0:
            overrides.append("--dev-no-guidance: conditional half only, batch %d" % batch)
            workload_label = workload_label.replace("CFG pair per step per GPU", "NO CFG pair: conditional half only (development run)")
        load = dict(getattr(qnn, "_build_laps", {}))
        load.update({"prepare_slots (plan + pack %d slots)" % len(slots): round(t_prep, 2),
                     "model_ready_s": round(sum(v for k, v in load.items() if k != "write cali_ckpt") + t_prep, 1),
                     "note": "model_ready_s = FP UNet + weights + get_qmodel (wrap, load_cali_model) + prepare_slots; the "
                             "synthetic ckpt's generation ('write cali_ckpt') is test-data synthesis, not a load cost; the reference's "
                             "load_cali_model alone is 91-107 s on CPU (SURVEY.md §6)",
                     "host_max_rss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1),
                     "model_ready_s_per_rank": ready_per_rank,
                     "device_mem_gb": None if DRY else round(torch.cuda.max_memory_allocated() / 1e9, 1)})
        out = {
            "metric": C["metric"], "value": round(K * n * P / elapsed, 4),
            "unit": "steps/s", "n_gpus": n, "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / (K * P), 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8 (W%dA%d MFMA, int32 accumulate; %s between layers)" % (qcfg["wbits"], qcfg["abits"], args.dtype),
            "data": "synthetic",
            "config": {"workload": workload_label, "config_id": args.config,
                       "prompts_per_gpu": P, "batch_per_gpu": batch, "parallelism": "replicas x%d (no collectives)" % n},
            "windows": {"n": len(windows), "steps_each": K, "value_from": "median",
                        "ms_per_step_min": round(1e3 * min(windows) / (K * P), 3),
                        "ms_per_step_all": [round(1e3 * w / (K * P), 3) for w in windows]},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "load": load,
            "non_hip_kernels": glue,
        }
        if overrides:
            out["config"]["dev_override"] = overrides        # (a line that carries this key is not the configuration BASELINE.json names)
        out.update(roofline_extra)                       # roofline_quant_act, roofline_attention, roofline_step
        if DRY:
            out["dry_run"] = "DGQ_BENCH_BACKEND=gloo: launcher rehearsal on CPU ranks with the FP tiny UNet — NOT a measurement"
            out["config"]["workload"] = "dry run: FP tiny UNet 16x16 on CPU (gloo)"
            out["dtype"] = "fp32 (dry run)"
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    return 0


HBM_PEAK_TBS = 8.0            # MI355X_MICROARCH.md: HBM3E peak (spec); ~6.3 TB/s is what a streaming copy achieves
BF16_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA peak: what the attention matmuls are priced against


def measure_gemm_roofline(torch, ops, qnn, run_step, args):
    """One eager step with the three launch hooks of dgq_amd.ops set: every GEMM launch (dgq_gemm_wxa8 / _batch, its split-K
    combine included), every quantise-on-load launch (dgq_quant_act_batch) and every attention call (dgq_attention: pre-pass +
    statistics + P·V) is replayed REP times from a hipGraph on the launch stream and bracketed by HIP events on that stream.
    Returns the ``roofline`` object of the JSON line: the W4A8 GEMM family against the int8 MFMA peak as before (``frac``), plus
    per launch the BINDING roof min(P_int8, AI·BW) -> ``ideal_ms`` = Σ max(ops/P, bytes/BW), the split of the family's time by
    which roof binds, and the same accounting for the quantise-on-load class (HBM) and the attention class (bf16 MFMA)."""
    REP = 5
    rec = {"gemm": [], "quant": [], "attn": []}
    layers = [0]

    def timed(issue):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(REP):
                issue()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / REP

    def gemm_hook(issue, problems):
        ms = timed(issue)
        layers[0] += len(problems)
        # a layer's bytes = its input once (un-unfolded, at its dtype) + int4 weights + output at its dtype.  The input
        # belongs to the launch that READS it: this one when it quantises its own operand (xb > 0), else the dgq_quant_act launch in front
        # of it (quant hook below) — the int8 code matrix between the two is overhead, not algorithmic traffic
        o = sum(2.0 * M * ab.pw.N * ab.pw.K for M, ab, _, _ in problems)
        by = sum(xb + ab.pw.N * ab.pw.K * ab.pw.bits / 8 + M * ab.pw.N * es for M, ab, es, xb in problems)
        rec["gemm"].append((ms, o, by))
        if os.environ.get("DGQ_BENCH_GEMM_DUMP"):            # per-launch table for tools (shape, scale mode, time)
            with open(os.environ["DGQ_BENCH_GEMM_DUMP"], "a") as f:
                f.write("%s %.2f\n" % (";".join("%d,%d,%d,%d,%s,%d" % (M, ab.pw.N, ab.pw.K, ab.Kp, ab.mode, es) for M, ab, es, _ in problems), 1e3 * ms))

    ops.GEMM_LAUNCH_HOOK = gemm_hook
    def dump(kind, ms, work):                                # per-launch tables for tools: DGQ_BENCH_QUANT_DUMP / _ATTN_DUMP=<file>
        path = os.environ.get("DGQ_BENCH_%s_DUMP" % kind)
        if path:
            with open(path, "a") as f:
                f.write("%.0f %.2f\n" % (work, 1e3 * ms))
        return ms

    overhead = [0.0]

    def quant_hook(issue, by, over):
        overhead[0] += over
        rec["quant"].append((dump("QUANT", timed(issue), by + over), 0.0, float(by)))

    ops.QUANT_LAUNCH_HOOK = quant_hook
    ops.ATTN_LAUNCH_HOOK = lambda issue, fl, by: rec["attn"].append((dump("ATTN", timed(issue), fl), float(fl), float(by)))
    graphs_were = qnn._graphs
    qnn._graphs = None
    try:
        with torch.no_grad():
            run_step()
        torch.cuda.synchronize()
    finally:
        qnn._graphs = graphs_were
        ops.GEMM_LAUNCH_HOOK = ops.QUANT_LAUNCH_HOOK = ops.ATTN_LAUNCH_HOOK = None

    def family(rows, peak_ops_tps):
        """time, work and the per-launch roofline of one kernel family: ideal = Σ max(ops / P, bytes / BW)"""
        ms = sum(r[0] for r in rows)
        ideal = [max(r[1] / (peak_ops_tps * 1e12), r[2] / (HBM_PEAK_TBS * 1e12)) * 1e3 for r in rows]
        compute = [r[1] / (peak_ops_tps * 1e12) >= r[2] / (HBM_PEAK_TBS * 1e12) for r in rows]
        return {"launches": len(rows), "ms_per_step": round(ms, 3), "ideal_ms_per_step": round(sum(ideal), 4),
                "frac_of_binding_roof": round(sum(ideal) / ms, 4) if ms else None,
                "compute_bound": {"launches": sum(compute), "ms": round(sum(r[0] for r, c in zip(rows, compute) if c), 3),
                                  "ideal_ms": round(sum(i for i, c in zip(ideal, compute) if c), 4)},
                "hbm_bound": {"launches": len(rows) - sum(compute), "ms": round(sum(r[0] for r, c in zip(rows, compute) if not c), 3),
                              "ideal_ms": round(sum(i for i, c in zip(ideal, compute) if not c), 4)},
                "ops": sum(r[1] for r in rows), "bytes": sum(r[2] for r in rows)}

    G, Q, A = family(rec["gemm"], INT8_PEAK_TOPS), family(rec["quant"], INT8_PEAK_TOPS), family(rec["attn"], BF16_PEAK_TFLOPS)
    gemm_ms = G["ms_per_step"]
    tops = G["ops"] / (gemm_ms * 1e-3) / 1e12
    n = G["launches"]
    # HBM-side bytes per launch come from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py); a STATIC
    # figure, attached only to the configuration it was profiled on AND only while no kernel source changed after the commit it
    # was measured at (a stale file is refused, not quoted)
    traffic, traffic_src = None, None
    import glob
    for tj in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_hbm_traffic.json")), reverse=True):
        tjd = json.load(open(tj))
        if tjd.get("config") == args.config and tjd.get("dtype") == args.dtype and tjd.get("prompts_per_gpu", 1) == (args.prompts_per_gpu or CONFIGS[args.config]["prompts"]):
            if traffic_file_is_current(tjd):
                traffic = round(tjd["traffic_bytes_per_launch"] / 1e6, 3)
                traffic_src = "static: profiles/%s (rocprofv3 --pmc passes of this command at commit %s, not this run)" % (
                    os.path.basename(tj), tjd.get("measured_at_commit"))
            else:
                traffic_src = "refused: profiles/%s (commit %s) was measured on other kernel sources than this build's (csrc digest %s)" % (
                    os.path.basename(tj), tjd.get("measured_at_commit"), csrc_digest())
            break
    hbm_gemm = G["hbm_bound"]["ideal_ms"] >= G["compute_bound"]["ideal_ms"]
    out = {"kernel": "dgq_gemm_wxa8 / dgq_gemm_wxa8_batch (gemm_wxa8_kernel<...> tile family, gemm_big_kernel<...>, gemm_panel_kernel<...> / gemm_convq_kernel<...> with the quantiser inside, "
                     "split-K combine where used); the 22 time_emb_proj layers (2 rows each, 0.1 Gop) run in dgq_linear_smallm_batch and are not counted",
           "layers_covered": layers[0],
           # which roof the family's IDEAL time mostly sits under (per launch: argmin(P_int8, AI·BW)); `achieved` / `frac` stay
           # the int8-MFMA accounting of the north star either way
           "bound": "hbm" if hbm_gemm else "mfma",
           "achieved": round(tops, 2), "peak": INT8_PEAK_TOPS, "unit": "TOP/s", "frac": round(tops / INT8_PEAK_TOPS, 4),
           "traffic": traffic, "traffic_unit": "MB of HBM-side reads+writes per launch", "traffic_source": traffic_src,
           "launches_per_step": n, "avg_launch_us": round(1e3 * gemm_ms / n, 2), "kernel_ms_per_step": round(gemm_ms, 3),
           "algorithmic_Gop_per_launch": round(G["ops"] / n / 1e9, 3),
           "algorithmic_MB_per_launch": round(G["bytes"] / n / 1e6, 3),
           "per_launch_roofline": {k: v for k, v in G.items() if k not in ("ops", "bytes")},
           "achieved_hbm_TBs": round(G["bytes"] / (gemm_ms * 1e-3) / 1e12, 3)}
    extra = {}
    if Q["launches"]:
        q_ms = Q["ms_per_step"]
        extra["roofline_quant_act"] = {"kernel": "dgq_quant_act_batch (quantise-on-load: im2col gather, GroupNorm/LayerNorm/SiLU prologue, codes + row sums)",
                                       "bound": "hbm", "achieved": round(Q["bytes"] / (q_ms * 1e-3) / 1e12, 3), "peak": HBM_PEAK_TBS,
                                       "unit": "TB/s", "frac": round(Q["bytes"] / (q_ms * 1e-3) / 1e12 / HBM_PEAK_TBS, 4),
                                       "launches_per_step": Q["launches"], "kernel_ms_per_step": q_ms, "ideal_ms_per_step": Q["ideal_ms_per_step"],
                                       "algorithmic_MB_per_launch": round(Q["bytes"] / Q["launches"] / 1e6, 3),
                                       "algorithmic": "the layer input read once, un-unfolded (SURVEY.md §8(d)); the layer's weights and output are the GEMM launch's",
                                       "overhead_MB_per_launch": round(overhead[0] / Q["launches"] / 1e6, 3),
                                       "overhead": "int8 code matrix [M][Kp] + row sums written for the GEMM launch behind: traffic the algorithm does not need",
                                       "moved_TBs": round((Q["bytes"] + overhead[0]) / (q_ms * 1e-3) / 1e12, 3)}
    if A["launches"]:
        a_ms = A["ms_per_step"]
        extra["roofline_attention"] = {"kernel": "dgq_attention (attn3_prep + attn3_stats + attn3_pv): aqtizer_q/k/v, QK^T, softmax, log2 / uniform aqtizer_w, P·V",
                                       "bound": "mfma (bf16 dense; the int8 Q·K^T form is priced at the bf16 rate too)",
                                       "achieved": round(A["ops"] / (a_ms * 1e-3) / 1e12, 2), "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": round(A["ops"] / (a_ms * 1e-3) / 1e12 / BF16_PEAK_TFLOPS, 4),
                                       "calls_per_step": A["launches"], "kernel_ms_per_step": a_ms, "ideal_ms_per_step": A["ideal_ms_per_step"],
                                       "algorithmic_flops": "4*T*S*D per (batch, head)"}
    tot_ms = gemm_ms + Q["ms_per_step"] + A["ms_per_step"]
    tot_ideal = G["ideal_ms_per_step"] + Q["ideal_ms_per_step"] + A["ideal_ms_per_step"]
    extra["roofline_step"] = {"covers": "GEMM family + quantise-on-load + attention (replayed per launch; GroupNorm finalisers, FP conv_in/out and "
                                        "torch glue are not replayed)",
                              "measured_ms": round(tot_ms, 3), "ideal_ms": round(tot_ideal, 4), "frac_of_binding_roofs": round(tot_ideal / tot_ms, 4),
                              "ideal": "sum over launches of max(ops / P, algorithmic bytes / 8 TB/s), P = 5000 TOP/s int8 (GEMM) or 2500 TFLOP/s bf16 (attention)"}
    return out, extra


GEMM_SOURCES = ("gemm_wxa8.hip", "gemm_wxa8_big.hip", "gemm_panel.hip", "gemm_convq.hip", "gemm_tile.h", "gemm_device.h", "quant_common.h")


def csrc_digest(names=GEMM_SOURCES):
    """sha256 over the CODE of the named kernel sources under dgq_amd/csrc (``//`` comments and blank lines dropped, so a comment
    fix does not orphan a measurement): what a PMC traffic record is stamped with (tools/pmc_traffic.py) — it describes THESE
    kernels and their launch planner or it is refused.  Works on a box that has no git history."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(names):
        h.update(name.encode())
        for line in open(os.path.join(ROOT, "dgq_amd", "csrc", name), encoding="utf-8"):
            code = line.split("//", 1)[0].strip()
            if code:
                h.update(code.encode() + b"\n")
    return h.hexdigest()[:16]


def traffic_file_is_current(record):
    """True when the record's ``csrc_digest`` equals the digest of the kernel sources this run was built from."""
    return bool(record.get("csrc_digest")) and record["csrc_digest"] == csrc_digest(tuple(record.get("digest_files") or GEMM_SOURCES))


if __name__ == "__main__":
    sys.exit(main())
