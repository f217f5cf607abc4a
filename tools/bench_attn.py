"""Fused attention per SD1.4 shape (B=2, H=8): hipGraph replay of 20 calls -> us per call (prep + stats + pv), for the
four operand paths: no fused quantizers (bf16x3 on raw fp32 q/k/v), aqtizer_q/k/v fused with per-token tables on
bf16x3 (DGQ_ATTN_I8=0), the int8 score path (per-token q/k: V_MFMA_I32_32X32X32_I8), and Q1K3 (per-head-dim aqtizer_q)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops
dev = torch.device("cuda:0")
shapes = [(40, 4096, 4096), (40, 4096, 77), (80, 1024, 1024), (80, 1024, 77), (160, 256, 256), (160, 256, 77), (160, 64, 64), (160, 64, 77)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
B, H, iters = 2, 8, 20


def timed(f):
    for _ in range(3): f()
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


for D, T, S in shapes:
    q, k, v = (torch.randn(B, n, H * D, device=dev) for n in (T, S, S))
    skip = 1 if S == 77 else 0
    tab = lambda n: (torch.rand(n, device=dev) * 0.02 + 0.02, torch.randint(100, 156, (n,), device=dev).float())
    fq = ((1,) + tab(T) + (0, 8), (1,) + tab(S - skip) + (skip, 8), (2,) + tab(D) + (0, 8))
    raw = timed(lambda: ops.attention_f32(q, k, v, H, D, D ** -0.5, 1, skip, None, 8))
    os.environ["DGQ_ATTN_I8"] = "0"
    f3 = timed(lambda: ops.attention_f32(q, k, v, H, D, D ** -0.5, 1, skip, None, 8, fq=fq))
    os.environ.pop("DGQ_ATTN_I8")
    i8 = timed(lambda: ops.attention_f32(q, k, v, H, D, D ** -0.5, 1, skip, None, 8, fq=fq))
    # per-head-dim aqtizer_q (the table most calibrated SD slots carry): one exact Q code plane against three K planes
    fq1 = ((2,) + tab(D) + (0, 8), (1,) + tab(S - skip) + (skip, 8), (2,) + tab(D) + (0, 8))
    q1 = timed(lambda: ops.attention_f32(q, k, v, H, D, D ** -0.5, 1, skip, None, 8, fq=fq1))
    flops = 2.0 * B * H * T * S * D * 2
    print("D=%3d T=%5d S=%5d  raw %8.1f us | fused-fq bf16x3 %8.1f us | int8 scores %8.1f us | Q1K3 %8.1f us  (%.1f TF/s algorithmic, int8)"
          % (D, T, S, raw, f3, i8, q1, flops / i8 / 1e6), flush=True)
