"""dgq_quant_act on the conv3x3 shapes of the SD1.4 step (GroupNorm + SiLU prologue folded in), hipGraph replay.
The library picks the path (block-staged patch kernel where conv_block_pays, else the row-wise scatter).  usage: python tools/bench_qact_conv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth, _lib
from dgq_amd.plan import plan_act
import ctypes

dev = torch.device("cuda:0")
ITERS = 10


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


tot = 0.0
for (H, C, stride, cnt) in ((64, 320, 1, 7), (64, 640, 1, 2), (64, 960, 1, 1), (64, 320, 2, 1), (32, 640, 1, 6), (32, 320, 1, 1), (32, 1280, 1, 2),
                            (32, 960, 1, 1), (32, 1920, 1, 1), (32, 640, 2, 1), (16, 1280, 1, 8), (16, 640, 1, 1), (16, 2560, 1, 2), (16, 1920, 1, 1),
                            (16, 1280, 2, 1), (8, 1280, 1, 6), (8, 2560, 1, 3)):
    B, taps = 2, 9
    x = torch.randn(B, H, H, C, device=dev)
    pw = ops.PackedWeight(torch.randn(32, C, 3, 3, device=dev) * 0.05, torch.full((32, 1), 0.01, device=dev), torch.full((32, 1), 8.0, device=dev),
                          None, None, 4, C, taps)
    sc, sh = torch.rand(B, C, device=dev) + 0.5, torch.randn(B, C, device=dev) * 0.1
    line = "H=%2d C=%4d s=%d x%d |" % (H, C, stride, cnt)
    for mode in ("perK", "perM"):
        Ho = (H + 2 - 3) // stride + 1
        if mode == "perK":
            d, z = synth._group_params(C * taps, 16, 8, "qc|%d" % C, 0)
            lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "conv", C, taps, 8, kw=3)
        else:
            d, z = synth._group_params(Ho * Ho, 16, 8, "qc|%d" % C, 0)
            lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "conv", C, taps, 8, kw=3)
        ab = ops.ActBinding(lay, pw, 8)
        us = replay_us(lambda: ops.quant_act(x, B, H, H, C, 3, 3, stride, 1, ab, pre=(sc, sh, 1)))
        line += "  %s Kp=%5d %6.1f us" % (mode, ab.Kp, us)
        tot += us * cnt * (0.6 if mode == "perK" else 0.4)
    print(line, flush=True)
print("weighted total (60 %% per-K / 40 %% per-M, counts of the SD graph): %.1f us" % tot)
