"""3x3 convolution layers of the 64 x 64 level (C = 320): dgq_quant_act + dgq_gemm_wxa8 (two launches, the int8 code matrix through HBM)
against the quantiser inside the GEMM launch (csrc/gemm_convq.hip), per scale mode; hipGraph replay of 20 calls -> us per layer.
Where the fused launch spends its time: tools/convq_timeline.py (in-kernel stamps of the diagnostic build).
usage: python tools/bench_convq.py ["B,C,H,W,N" ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act
dev = torch.device("cuda:0")
shapes = [(2, 320, 64, 64, 320), (2, 320, 32, 32, 320), (8, 320, 64, 64, 320)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
ITERS = 20


def timed(f):
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(ITERS): f()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / ITERS)
    return best


for B, C, H, W, N in shapes:
    gen = torch.Generator().manual_seed(1)
    w = (torch.randn(N, C, 3, 3, generator=gen) * 0.05).to(dev)
    x = (torch.randn(B, C, H, W, generator=gen) * 1.3).to(dev).contiguous(memory_format=torch.channels_last)
    wd, wz = synth.channel_minmax(w.cpu(), 4)
    pw = ops.PackedWeight(w, wd.to(dev), wz.to(dev), None, torch.randn(N, generator=gen).to(dev), 4, C, 9)
    res = torch.randn(B, N, H, W, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    sc, sh = torch.rand(B, C, device=dev) + 0.5, torch.randn(B, C, device=dev) * 0.1
    for mode in ("perK", "perM"):
        if mode == "perK":
            d, z = synth._group_params(C * 9, 16, 8, "cq|%d" % C, 0)
            lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "conv", C, 9, 8)
        else:
            d, z = synth._group_params(H * W, 16, 8, "cq|%d" % C, 0)
            lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "conv", C, 9, 8)
        ab = ops.ActBinding(lay, pw, 8)
        orig = ops.groupnorm_scale_shift
        ops.groupnorm_scale_shift = lambda *a, **k: (sc, sh)            # (the statistics pass is not what is timed)
        norm = (32, 1e-5, None, None, 1)
        t = {}
        for fuse in (False, True):
            ops.CONV_FUSE = fuse
            t[fuse] = timed(lambda: ops.quant_conv2d(x, ab, 3, 3, 1, 1, norm=norm, residual=res))
        ops.groupnorm_scale_shift = orig
        M = B * H * W
        print("B=%d C=%d %dx%d -> N=%d %s (M=%d, Kp=%d): quantise + GEMM %6.1f us | one launch %6.1f us   [ideal: %.1f us MFMA, %.1f us HBM]"
              % (B, C, H, W, N, mode, M, ab.Kp, t[False], t[True], 2.0 * M * N * C * 9 / 5e15 * 1e6, (2 * M * C * 4 + N * C * 9 / 2) / 8e12 * 1e6), flush=True)
