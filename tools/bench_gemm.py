"""GEMM microbenchmark (GPU box): dgq_gemm_wxa8 on representative SD1.4 / SDXL shapes,
back-to-back launches timed with HIP events on the launch stream; prints algorithmic TOP/s and the fraction
of the dense int8 MFMA peak (5000 TOP/s).   usage: python tools/bench_gemm.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
PEAK = 5000.0
# (name, M, N, C, taps, mode)
SHAPES_ALL = [
    ("sd conv3x3 64x64 320->320", 8192, 320, 320, 9, "perK"),
    ("sd conv3x3 32x32 640->640", 2048, 640, 640, 9, "perK"),
    ("sd conv3x3 16x16 1280->1280", 512, 1280, 1280, 9, "perK"),
    ("sd conv3x3 8x8 2560->1280", 128, 1280, 2560, 9, "perK"),
    ("sd conv3x3 16x16 2560->1280", 512, 1280, 2560, 9, "perK"),
    ("sd conv3x3 64x64 960->320", 8192, 320, 960, 9, "perM"),
    ("sd geglu 64x64 320->2560", 8192, 2560, 320, 1, "perK"),
    ("sd geglu 32x32 640->5120", 2048, 5120, 640, 1, "perK"),
    ("sd geglu 16x16 1280->10240", 512, 10240, 1280, 1, "perK"),
    ("sd ff.2 64x64 1280->320", 8192, 320, 1280, 1, "perK"),
    ("sd to_q 64x64 320->320", 8192, 320, 320, 1, "perM"),
    ("sd to_q 16x16 1280->1280", 512, 1280, 1280, 1, "perK"),
    ("sd to_k ctx 768->320", 154, 320, 768, 1, "perK"),
    ("xl geglu 32x32 1280->10240 (B=1)", 1024, 10240, 1280, 1, "perK"),
    ("xl geglu 64x64 640->5120 (B=1)", 4096, 5120, 640, 1, "perK"),
    ("xl ff.2 32x32 5120->1280", 1024, 1280, 5120, 1, "perK"),
    ("big 8192x8192x8192 perM", 8192, 8192, 8192, 1, "perM"),
    ("big 8192x8192x8192 perK g16", 8192, 8192, 8192, 1, "perK"),
]
# SURVEY.md §8(d) headline shapes: GEGLU-class Linear layers at B = 1 and at the batch a GPU holds in config C5 (8 prompts),
# the compute-bound convs, 8192^3 — per-K g16 and per-M, fp32 and bf16 outputs
HEADLINE = [
    ("sd geglu 16x16 1280->10240", 512, 10240, 1280, 1), ("xl geglu 32x32 1280->10240 B=1", 1024, 10240, 1280, 1),
    ("xl geglu 32x32 1280->10240 B=8", 8192, 10240, 1280, 1), ("xl ff.2 32x32 5120->1280 B=8", 8192, 1280, 5120, 1),
    ("sd conv3x3 64x64 320->320", 8192, 320, 320, 9), ("sd conv3x3 16x16 1280->1280", 512, 1280, 1280, 9),
    ("xl conv3x3 32x32 1280->1280 B=8", 8192, 1280, 1280, 9), ("big 8192^3", 8192, 8192, 8192, 1),
]
OUT_DTYPES = {"fp32": torch.float32, "bf16": torch.bfloat16}
if os.environ.get("HEADLINE"):
    SHAPES = [(n, M, N, C, t, mode, od) for (n, M, N, C, t) in HEADLINE for mode in ("perK", "perM") for od in ("fp32", "bf16")]
else:
    SHAPES = [s + ("fp32",) for s in (SHAPES_ALL[-2:] if os.environ.get("BIG_ONLY") else SHAPES_ALL)]
# SHAPE="M,N,C,taps[;M,N,C,taps...]": custom shapes instead (per-K g16 and per-M, fp32 and bf16 outputs)
if os.environ.get("SHAPE"):
    SHAPES = [("custom %s" % sh, *(int(x) for x in sh.split(",")), mode, od) for sh in os.environ["SHAPE"].split(";")
              for mode in ("perK", "perM") for od in ("fp32", "bf16")]
# VARIANTS="default;128,128,1;256,256,1": each shape under several launch plans (DGQ_GEMM_FORCE, read per call) in ONE process
VARIANTS = os.environ.get("VARIANTS", "default").split(";")
if os.environ.get("ONLY"):
    SHAPES = [s for s in SHAPES if any(k in s[0] for k in os.environ["ONLY"].split(";"))]
if os.environ.get("MODES"):
    SHAPES = [s for s in SHAPES if s[5] in os.environ["MODES"].split(";")]
if os.environ.get("OUTS"):
    SHAPES = [s for s in SHAPES if s[6] in os.environ["OUTS"].split(";")]
print("%-40s %6s %6s %6s %5s %5s %12s %9s %9s %7s" % ("shape", "M", "N", "K", "mode", "out", "plan", "us", "TOP/s", "frac"))
for name, M, N, C, taps, mode, od in SHAPES:
    odt = OUT_DTYPES[od]
    K = C * taps
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) * 0.05
    wd, wz = synth.channel_minmax(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, C, taps)
    if mode == "perK":
        d, z = synth._group_params(K, 16, 8, "bench|" + name, 0)
        lay = plan_act(d.view(1, -1, 1) if taps > 1 else d.view(1, 1, -1), z.view(1, -1, 1) if taps > 1 else z.view(1, 1, -1),
                       "conv" if taps > 1 else "linear", C, taps, 8)
    else:
        d, z = synth._group_params(64, 16, 8, "bench|" + name, 0)
        lay = plan_act(d.view(1, 1, -1) if taps > 1 else d.view(1, -1, 1), z.view(1, 1, -1) if taps > 1 else z.view(1, -1, 1),
                       "conv" if taps > 1 else "linear", C, taps, 8)
    ab = ops.ActBinding(lay, pw, 8)
    codes = torch.randint(-128, 128, (M, ab.Kp), dtype=torch.int8, device=dev)
    rowsum = torch.randn(M, device=dev)
    out = torch.empty(M, N, device=dev, dtype=odt)
    for var in VARIANTS:
        if var == "default":
            os.environ.pop("DGQ_GEMM_FORCE", None)
        else:
            os.environ["DGQ_GEMM_FORCE"] = var
        for _ in range(3):
            ops.gemm_wxa8(codes, rowsum, M, ab, odt, out)
        torch.cuda.synchronize()
        # capture `iters` launches in a hipGraph so that host launch overhead (python + ctypes ~15 us/call) is excluded
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(iters):
                ops.gemm_wxa8(codes, rowsum, M, ab, odt, out)
        graph.replay()
        torch.cuda.synchronize()
        best = 1e30
        for _rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
        us = best
        tops = 2.0 * M * N * K / (us * 1e-6) / 1e12
        print("%-40s %6d %6d %6d %5s %5s %12s %9.1f %9.1f %6.1f%%  Kp=%d" % (name, M, N, K, mode, od, var, us, tops, 100 * tops / PEAK, ab.Kp), flush=True)
        del graph
    os.environ.pop("DGQ_GEMM_FORCE", None)
    del pw, ab, codes, out
