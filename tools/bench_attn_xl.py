"""Fused attention on the SDXL shapes (D = 64): hipGraph replay of 10 calls -> us per call (prep + stats + pv), for the two
configurations bench.py times: C5 (W4A6 g=1: scalar aqtizer_q/k/v, uniform softmax quantiser with a static δ, 8 prompts) and C4
(W4A8 g16: per-head-dim aqtizer tables, real-time log2 softmax quantiser, 1 prompt).
usage: bench_attn_xl.py [c5|c4 ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops
dev = torch.device("cuda:0")
D, iters = 64, 10
which = sys.argv[1:] or ["c5", "c4"]


def timed(f):
    for _ in range(2): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): f()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


for cfg in which:
    B = 8 if cfg == "c5" else 1
    bits = 6 if cfg == "c5" else 8
    for H, T, S, n in ((10, 4096, 4096, 10), (10, 4096, 77, 10), (20, 1024, 1024, 60), (20, 1024, 77, 60)):
        q, k, v = (torch.randn(B, m, H * D, device=dev) for m in (T, S, S))
        lo = 2 ** (bits - 1) - 2 ** (bits - 3)
        tab = lambda m: (torch.rand(m, device=dev) * 0.02 * 2 ** (8 - bits) + 0.02 * 2 ** (8 - bits),
                         torch.randint(lo, lo + 2 ** (bits - 2), (m,), device=dev).float())
        if cfg == "c5":
            fq = tuple((0,) + tab(1) + (0, bits) for _ in range(3))
            delta = torch.full((1,), 1.0 / (2 ** bits - 1), device=dev)
            f = lambda: ops.attention(q, k, v, H, D, D ** -0.5, 3, 0, delta, bits, fq=fq)
        else:
            skip = 1 if S == 77 else 0
            fq = ((2,) + tab(D) + (0, 8), (2,) + tab(D) + (skip, 8), (2,) + tab(D) + (0, 8))
            f = lambda: ops.attention(q, k, v, H, D, D ** -0.5, 1, skip, None, 8, fq=fq)
        us = timed(f)
        flops = 4.0 * B * H * T * S * D
        print("%s  B=%d H=%2d T=%5d S=%5d  %8.1f us per call  x %2d calls per step = %7.2f ms   (%.1f TF/s algorithmic)"
              % (cfg, B, H, T, S, us, n, us * n / 1e3, flops / us / 1e6), flush=True)
