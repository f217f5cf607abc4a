"""conv_out (FP 3x3 conv 320 -> 4 with conv_norm_out + SiLU folded into its load) through dgq_conv2d_f32w's N <= 8 kernel: us per launch
at the SD (B = 2, 64x64) and SDXL C5 (B = 8, 128x128) sizes.   usage: bench_conv_out.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops
dev = torch.device("cuda:0")
for (B, H, C) in ((2, 64, 320), (1, 128, 320), (8, 128, 320)):
    x = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(4, 3, 3, C, device=dev) * 0.05).reshape(4, 9 * C).contiguous()      # [N][tap·C + c]
    bias = torch.randn(4, device=dev)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    f = lambda: ops.conv2d_f32w(x, w, bias, 3, 3, 1, 1, norm=(32, 1e-5, gamma, beta, True))
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("B=%d %dx%d C=%d -> 4: %.1f us" % (B, H, H, C, e0.elapsed_time(e1) * 1e3 / 10), flush=True)
