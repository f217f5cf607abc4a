"""Fused attention per SD1.4 shape (B=2, H=8): hipGraph replay of 20 calls -> us per call (prep + stats + pv)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops
dev = torch.device("cuda:0")
shapes = [(40, 4096, 4096), (40, 4096, 77), (80, 1024, 1024), (80, 1024, 77), (160, 256, 256), (160, 256, 77), (160, 64, 64), (160, 64, 77)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
B, H, iters = 2, 8, 20
for D, T, S in shapes:
    q, k, v = (torch.randn(B, n, H * D, device=dev) for n in (T, S, S))
    skip = 1 if S == 77 else 0
    f = lambda: ops.attention_f32(q, k, v, H, D, D ** -0.5, 1, skip, None, 8)
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    flops = 2.0 * B * H * T * S * D * 2
    print("D=%3d T=%5d S=%5d  %8.1f us/call   %.1f TF/s algorithmic (QK^T + PV once each)" % (D, T, S, us, flops / us / 1e6), flush=True)
