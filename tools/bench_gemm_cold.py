"""Weight-streaming launches (M <= 512) timed the way the step runs them: COLD weights.

Inside a UNet step every layer's weights are read once and the step's 430 MB of int4 do not stay in the 256 MB Infinity Cache, so a
launch streams its weights from HBM; a back-to-back replay of ONE layer (tools/tile_sweep.py) finds them in L2 / MALL instead and
ranks launch plans for a machine state the step never has.  Here a graph cycles through R private copies of the packed weights
(R x bytes >= 320 MB), activations and tables shared, so every launch misses on its weights.
usage: python tools/bench_gemm_cold.py ["M,N,C,taps,mode" ...]      (DGQ_GEMM_FORCE candidates: TILES / SPLITS env, ';' separated)"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act

dev = torch.device("cuda:0")
SHAPES = [(128, 1280, 1280, 9, "perM"), (128, 1280, 1280, 9, "perK"), (512, 1280, 1280, 9, "perK"), (512, 1280, 1280, 9, "perM"),
          (128, 1280, 2560, 9, "perM"), (512, 1280, 2560, 9, "perK"), (512, 1280, 1280, 1, "perK"), (512, 1280, 1280, 1, "perM"),
          (512, 1280, 5120, 1, "perK"), (512, 10240, 1280, 1, "perK"), (512, 10240, 1280, 1, "perM"), (128, 1280, 1280, 1, "perK"),
          (2048, 640, 640, 9, "perK"), (2048, 640, 640, 9, "perM"), (2048, 1280, 1280, 9, "perK")]
if len(sys.argv) > 1:
    SHAPES = []
    for s in sys.argv[1:]:
        f = s.split(",")
        SHAPES.append((int(f[0]), int(f[1]), int(f[2]), int(f[3]), f[4]))
TILES = [tuple(int(v) for v in t.split("x")) for t in os.environ.get("TILES", "32x64;64x64;128x64;64x128").split(";")]
SPLITS = [int(v) for v in os.environ.get("SPLITS", "1;2;3;4;6;8;12;16;24;32").split(";")]
COLD_BYTES = 160 << 20          # per image (row-major + fragment-major copies: 320 MB per cycle)


def replay_us(fns):
    for f in fns[:2]:
        f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / len(fns))
    del g
    return best


print("%6s %6s %6s %6s %5s %4s | %8s %8s | %-20s | cold us per (tile/S)" % ("M", "N", "K", "Kp", "mode", "R", "auto warm", "auto cold", "best cold"))
for (M, N, C, taps, mode) in SHAPES:
    K = C * taps
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) * 0.05
    wd, wz = synth.channel_minmax(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, C, taps)
    if mode == "perK":
        d, z = synth._group_params(K, 16, 8, "cold|%d|%d" % (N, K), 0)
        lay = plan_act(d.view(1, -1, 1) if taps > 1 else d.view(1, 1, -1), z.view(1, -1, 1) if taps > 1 else z.view(1, 1, -1),
                       "conv" if taps > 1 else "linear", C, taps, 8)
    else:
        d, z = synth._group_params(64, 16, 8, "cold|%d|%d" % (N, K), 0)
        lay = plan_act(d.view(1, 1, -1) if taps > 1 else d.view(1, -1, 1), z.view(1, 1, -1) if taps > 1 else z.view(1, -1, 1),
                       "conv" if taps > 1 else "linear", C, taps, 8)
    ab = ops.ActBinding(lay, pw, 8)
    R = max(4, min(400, -(-COLD_BYTES // ab.wpacked.numel())))
    abs_ = []
    for r in range(R):
        c = copy.copy(ab)
        c.wpacked = ab.wpacked.clone()
        c.wfrag = ab.wfrag.clone() if ab.wfrag is not None else None
        abs_.append(c)
    codes = torch.randint(-128, 128, (M, ab.Kp), dtype=torch.int8, device=dev)
    rowsum = torch.randn(M, device=dev)
    out = torch.empty(M, N, device=dev)
    os.environ.pop("DGQ_GEMM_FORCE", None)
    warm = replay_us([lambda: ops.gemm_wxa8(codes, rowsum, M, ab, torch.float32, out)] * 20)
    cold = replay_us([(lambda c=c: ops.gemm_wxa8(codes, rowsum, M, c, torch.float32, out)) for c in abs_])
    res = []
    nk = ab.Kp // 128
    for bm, bn in TILES:
        grid = -(-M // bm) * -(-N // bn)
        for s in SPLITS:
            if s > 1 and (s * 2 > nk or grid * s > 4096 or s * M * N * 4 > ops.WORKSPACE_BYTES):
                continue
            os.environ["DGQ_GEMM_FORCE"] = "%d,%d,%d" % (bm, bn, s)
            res.append((replay_us([(lambda c=c: ops.gemm_wxa8(codes, rowsum, M, c, torch.float32, out)) for c in abs_]), bm, bn, s))
    os.environ.pop("DGQ_GEMM_FORCE", None)
    best = min(res)
    wbytes = ab.wpacked.numel()
    print("%6d %6d %6d %6d %5s %4d | %8.1f %8.1f | %3dx%-3d/%-2d %7.1f us | %s   [weights %.1f MB: %.1f us at 6 TB/s]" % (
        M, N, K, ab.Kp, mode, R, warm, cold, best[1], best[2], best[3], best[0],
        " ".join("%dx%d/%d:%.1f" % (r[1], r[2], r[3], r[0]) for r in sorted(res)[:10]), wbytes / 1e6, wbytes / 6e6), flush=True)
    del abs_, pw, ab, codes, out
    torch.cuda.empty_cache()
