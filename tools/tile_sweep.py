"""Tile-shape / K-split sweep of dgq_gemm_wxa8 (GPU box): for every distinct layer shape of the SD1.4 step (and the SDXL /
headline GEGLU shapes) time each (BM, BN, splits) candidate through the DGQ_GEMM_FORCE development hook, next to what the
library's own plan picks.  hipGraph replay of ITERS launches, HIP events on the launch stream.
usage: python tools/tile_sweep.py [sd|xl|all] > gpurun_out/tile_sweep.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act

dev = torch.device("cuda:0")
ITERS = 10
# (M, N, K, taps, mode, count per SD step)
SD = [(2048, 640, 640, 1, "perK", 16), (8192, 320, 320, 1, "perK", 15), (512, 1280, 1280, 1, "perK", 25),
      (8192, 2560, 320, 1, "perK", 5), (128, 1280, 1280, 9, "perM", 8), (2048, 640, 640, 1, "perM", 17),
      (8192, 320, 320, 1, "perM", 17), (512, 1280, 1280, 1, "perM", 14), (2048, 640, 640, 9, "perM", 4),
      (8192, 320, 320, 9, "perK", 4), (154, 1280, 768, 1, "perK", 7), (2, 1280, 1280, 1, "perM", 13),
      (512, 1280, 1280, 9, "perK", 3), (2048, 640, 2560, 1, "perK", 3), (2048, 640, 640, 9, "perK", 2),
      (512, 1280, 1280, 9, "perM", 3), (2048, 5120, 640, 1, "perM", 4), (512, 1280, 5120, 1, "perM", 3),
      (128, 1280, 1280, 9, "perK", 3), (512, 10240, 1280, 1, "perK", 2), (8192, 640, 640, 9, "perK", 1),
      (8192, 320, 960, 9, "perK", 1), (2048, 1280, 1280, 9, "perM", 1), (8192, 320, 1280, 1, "perK", 2),
      (512, 10240, 1280, 1, "perM", 3), (2048, 640, 1920, 9, "perK", 1), (512, 1280, 2560, 9, "perK", 1),
      (154, 320, 768, 1, "perM", 4), (128, 1280, 2560, 9, "perK", 1), (8192, 320, 640, 9, "perM", 1)]
XL = [(1024, 10240, 1280, 1, "perK", 1), (1024, 1280, 5120, 1, "perK", 1), (1024, 1280, 1280, 1, "perK", 1),
      (4096, 5120, 640, 1, "perK", 1), (4096, 640, 640, 1, "perK", 1), (4096, 640, 2560, 1, "perK", 1),
      (1024, 1280, 1280, 9, "perK", 1), (8192, 10240, 1280, 1, "perK", 1), (8192, 1280, 5120, 1, "perK", 1),
      (8192, 10240, 1280, 1, "perM", 1)]
which = sys.argv[1] if len(sys.argv) > 1 else "sd"
shapes = SD if which == "sd" else XL if which == "xl" else SD + XL
TILES = [(128, 128), (128, 64), (64, 128), (64, 64), (32, 128), (32, 64)]


def replay_us(fn):
    for _ in range(2):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(ITERS):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / ITERS)
    return best


print("%6s %6s %6s %6s %5s | %9s | %-22s | all (BMxBN/S us)" % ("M", "N", "K", "Kp", "mode", "auto us", "best"))
tot_auto = tot_best = 0.0
for (M, N, C, taps, mode, cnt) in shapes:
    K = C * taps
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) * 0.05
    wd, wz = synth.channel_minmax(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, C, taps)
    if mode == "perK":
        d, z = synth._group_params(K, 16, 8, "sweep|%d|%d" % (N, K), 0)
        lay = plan_act(d.view(1, -1, 1) if taps > 1 else d.view(1, 1, -1), z.view(1, -1, 1) if taps > 1 else z.view(1, 1, -1),
                       "conv" if taps > 1 else "linear", C, taps, 8)
    else:
        d, z = synth._group_params(64, 16, 8, "sweep|%d|%d" % (N, K), 0)
        lay = plan_act(d.view(1, 1, -1) if taps > 1 else d.view(1, -1, 1), z.view(1, 1, -1) if taps > 1 else z.view(1, -1, 1),
                       "conv" if taps > 1 else "linear", C, taps, 8)
    ab = ops.ActBinding(lay, pw, 8)
    codes = torch.randint(-128, 128, (M, ab.Kp), dtype=torch.int8, device=dev)
    rowsum = torch.randn(M, device=dev)
    out = torch.empty(M, N, device=dev)
    os.environ.pop("DGQ_GEMM_FORCE", None)
    auto = replay_us(lambda: ops.gemm_wxa8(codes, rowsum, M, ab, torch.float32, out))
    res = []
    nk = ab.Kp // 128
    for bm, bn in TILES:
        grid = -(-M // bm) * -(-N // bn)
        cands = [1]
        for s in (2, 3, 4, 6, 8, 12, 16):
            if s * 2 <= nk and grid * s <= 2048 and s * M * N * 4 <= ops.WORKSPACE_BYTES:
                cands.append(s)
        for s in cands:
            os.environ["DGQ_GEMM_FORCE"] = "%d,%d,%d" % (bm, bn, s)
            res.append((replay_us(lambda: ops.gemm_wxa8(codes, rowsum, M, ab, torch.float32, out)), bm, bn, s))
    os.environ.pop("DGQ_GEMM_FORCE", None)
    res.sort()
    b = res[0]
    tot_auto += auto * cnt; tot_best += b[0] * cnt
    print("%6d %6d %6d %6d %5s | %9.1f | %3dx%-3d/%-2d %7.1f us | %s" % (
        M, N, K, ab.Kp, mode, auto, b[1], b[2], b[3], b[0],
        " ".join("%dx%d/%d:%.1f" % (r[1], r[2], r[3], r[0]) for r in res[:8])), flush=True)
    del pw, ab, codes, out
print("weighted total: auto %.1f us, best %.1f us" % (tot_auto, tot_best))
