"""K sweep at fixed tile grid: slope = time per K tile, intercept = fixed per-launch cost (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act
dev = torch.device("cuda:0")
iters = 20
for (M, N) in ((8192, 320), (8192, 2560), (512, 1280), (128, 128), (256, 256)):
    for K in (128, 256, 512, 1024, 2048, 4096):
        w = torch.randn(N, K) * 0.05
        wd, wz = synth.channel_minmax(w, 4)
        pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, K, 1)
        d, z = synth._group_params(64, 16, 8, "sweep", 0)
        lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "linear", K, 1, 8)
        ab = ops.ActBinding(lay, pw, 8)
        codes = torch.randint(-128, 128, (M, ab.Kp), dtype=torch.int8, device=dev)
        rowsum = torch.randn(M, device=dev)
        out = torch.empty(M, N, device=dev)
        for _ in range(3):
            ops.gemm_wxa8(codes, rowsum, M, ab, torch.float32, out)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(iters):
                ops.gemm_wxa8(codes, rowsum, M, ab, torch.float32, out)
        graph.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); graph.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print("M=%5d N=%5d K=%5d nk=%3d  %8.1f us" % (M, N, K, ab.Kp // 128, us), flush=True)
