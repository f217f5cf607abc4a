"""bench.py — denoise-step throughput of the quantized SD1.4 UNet on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Metric (BASELINE.json): UNet denoise steps/sec @ SD1.4 512² W4A8 g16.  A "step" = one QuantModel.forward on a CFG
pair ([2,4,64,64] latents, [2,77,768] context) inside the DDIM loop (CFG combine + scheduler update included),
synthetic name-keyed weights and a synthetic reference-format cali_ckpt (time-aware act tables, 16 DGQ groups,
log2-quantised softmax with real-time δ and start-peak — the reference's own preset for G>1,
scripts/quantize_act.sh:16-19).  Inputs are resident in HBM when the timed region starts.  Each rank denoises
its own prompts (weak scaling, no data-path collective); value = (steps × ranks) / max-over-ranks time.

The JSON line also carries
  roofline     — the W4A8 MFMA GEMM (dgq_gemm_wxa8): algorithmic int8 ops of all 280 quantized layers of one
                 step (2·M·N·K, SURVEY.md §8(d): 1.354 Top) ÷ the HIP-event time of those launches, against the
                 dense int8 MFMA peak (≈5 Pop/s = 2× the 2.5 PF bf16 peak, MI355X_MICROARCH.md § Matrix cores);
  cpu_baseline — the reference's op sequence (oracle/, a pinned CPU port: re-quantised weights each call, F.unfold,
                 fp32 GEMMs, materialised attention) timed for ONE step on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG_C2 = dict(wbits=4, abits=8, use_aq=True, G=16, log=True, rt=True, sp=True, time_aware=True, steps=50)
INT8_PEAK_TOPS = 5000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per timestep slot")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-launch GEMM replay pass (profiling runs)")
    ap.add_argument("--prompts-per-gpu", type=int, default=1,
                    help="prompts denoised together on each GPU (default 1 = the CFG-pair step the metric is defined on)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp16", "bf16"],
                    help="inter-layer activation dtype (fp32 = the reference's default .float() mode)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch.cuda.set_device(local_rank)
    if "RANK" in os.environ:          # launched by torch.distributed.run: one rank per GPU over RCCL (timing only)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    from dgq_amd import ops, synth
    from dgq_amd.runtime import build_synthetic_qnn, DDIMScheduler
    from dgq_amd._lib import require_gpu
    require_gpu()

    def barrier():
        if dist is not None:
            dist.barrier()

    K, W = args.steps, args.warmup
    sch = DDIMScheduler(50)
    n_ts = min(K + W, 50)
    timesteps = [sch.timesteps[i % n_ts] for i in range(W + K)]
    slots = sorted({(1000 - t) // 20 for t in timesteps})
    P = args.prompts_per_gpu
    qnn, ckpt_path = build_synthetic_qnn("sd", CFG_C2, 64, 2 * P, max(slots) + 1, rank=local_rank, barrier=barrier, device=dev)
    if args.dtype == "fp16":
        qnn.half()
    elif args.dtype == "bf16":
        qnn.to(torch.bfloat16)
    adt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    qnn.prepare_slots(slots)
    if not args.no_graph:
        qnn.enable_graphs(True)

    # this rank's prompt (seeded by rank: rank-sliced prompt list), resident on the device
    lat = synth.named_randn("latent", (P, 4, 64, 64), 1 + rank).to(dev, adt)
    ctx = synth.named_randn("ctx", (2 * P, 77, 768), 100 + rank).to(dev, adt)
    guidance = 7.5

    def one_step(x, t):
        inp = torch.cat([x, x], dim=0)
        eps = qnn(inp, t, ctx)[0]
        e_u, e_c = eps.chunk(2)
        return sch.step(e_u + guidance * (e_c - e_u), t, x)

    x = lat
    with torch.no_grad():
        if not args.no_graph:
            for t in sorted(set(timesteps), reverse=True):     # capture one graph per slot used (amortised over images)
                one_step(lat, t)
        for t in timesteps[:W]:
            x = one_step(x, t)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in timesteps[W:]:
            x = one_step(x, t)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    assert torch.isfinite(x).all()
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- roofline of the dominant north-star kernel (W4A8 GEMM), measured live with HIP events ------------------
    # One eager step with ops.gemm_wxa8 wrapped: every one of the 280 launches (its split-K reduction included) is
    # replayed REP times back to back from a hipGraph on the current stream and bracketed by HIP events — eager
    # per-launch events would time the host-side launch gaps, not the kernels.  Sum over the step = the kernel time
    # rocprofv3 reports for gemm_wxa8_kernel + splitk_epilogue_kernel (profiles/).
    roofline = None
    if rank == 0 and not args.no_roofline:
        REP = 5
        times_ms, algo_ops, algo_bytes = [], [], []
        orig = ops.gemm_wxa8

        def timed(codes, rowsum, M, ab, out_dtype, out=None, extra=None):
            y = orig(codes, rowsum, M, ab, out_dtype, out, extra)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(REP):
                    orig(codes, rowsum, M, ab, out_dtype, y, extra)
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            e1.synchronize()
            times_ms.append(e0.elapsed_time(e1) / REP)
            algo_ops.append(2.0 * M * ab.pw.N * ab.pw.K)
            # un-unfolded input counted once at 1 B/code, int4 weights, output at its dtype
            algo_bytes.append(M * ab.pw.K / max(1, ab.pw.taps) + ab.pw.N * ab.pw.K / 2 + M * ab.pw.N * y.element_size())
            return y
        ops.gemm_wxa8 = timed
        graphs_were = qnn._graphs
        qnn._graphs = None
        with torch.no_grad():
            one_step(lat, timesteps[W])
        torch.cuda.synchronize()
        qnn._graphs = graphs_were
        ops.gemm_wxa8 = orig
        gemm_ms = sum(times_ms)
        tops = sum(algo_ops) / (gemm_ms * 1e-3) / 1e12
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r01_gemm_hbm_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        if os.path.exists(tj):
            traffic = round(json.load(open(tj))["traffic_bytes_per_launch"] / 1e6, 3)
        n = len(times_ms)
        roofline = {"kernel": "gemm_wxa8_kernel<4,*> + splitk_epilogue_kernel (dgq_gemm_wxa8)", "bound": "mfma",
                    "achieved": round(tops, 2), "peak": INT8_PEAK_TOPS, "unit": "TOP/s", "frac": round(tops / INT8_PEAK_TOPS, 4),
                    "traffic": traffic, "traffic_unit": "MB of HBM-side reads+writes per launch (PMC, profiles/r01_gemm_hbm_traffic.json)",
                    "launches_per_step": n, "avg_launch_us": round(1e3 * gemm_ms / n, 2), "kernel_ms_per_step": round(gemm_ms, 3),
                    "algorithmic_Gop_per_launch": round(sum(algo_ops) / n / 1e9, 3),
                    "algorithmic_MB_per_launch": round(sum(algo_bytes) / n / 1e6, 3)}

    # ---- CPU baseline: the reference's op sequence (oracle port) on this box's host cores, one step --------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import dgq_oracle as orc
        ncores = min(os.cpu_count() or 1, 64)
        torch.set_num_threads(ncores)
        ck = torch.load(ckpt_path, map_location="cpu")
        cfg = orc.OracleConfig("sd", 4, 8, True, True, 8, True, True, True, True, 50, True)
        om = orc.OracleModel(ck, cfg, synth.synth_state_dict("sd", 0))
        lat_c = synth.named_randn("latent", (1, 4, 64, 64), 1)[:1]
        ctx_c = synth.named_randn("ctx", (2, 77, 768), 100)
        t = timesteps[W]
        tc0 = time.perf_counter()
        om.forward(torch.cat([lat_c, lat_c]), t, ctx_c)
        cpu_s = time.perf_counter() - tc0
        cpu_baseline = {"value": round(1.0 / cpu_s, 5), "unit": "steps/s", "cores": ncores, "kind": "port",
                        "sample": "1 UNet denoise step (CFG pair, t=%d) of the same SD1.4 W4A8 g16 workload, fp32, "
                                  "torch.set_num_threads(%d); no warm-up (the port keeps no state)" % (t, ncores)}

    if rank == 0:
        n = max(world, 1)
        out = {
            "metric": "UNet denoise steps/sec @ SD1.4 512^2 W4A8 g16", "value": round(K * n * P / elapsed, 4),
            "unit": "steps/s", "n_gpus": n, "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / (K * P), 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8 (W4A8 MFMA, int32 accumulate; %s between layers)" % args.dtype, "data": "synthetic",
            "config": {"workload": "SD v1.4 UNet W4A8 g=16 (time-aware, log2 softmax real-time δ, start-peak), "
                                   "DDIM 50-step schedule, 512x512 (64x64 latents), CFG pair per step per GPU",
                       "prompts_per_gpu": P, "cfg_batch": 2 * P, "parallelism": "replicas x%d (no collectives)" % n},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
