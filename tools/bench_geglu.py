"""ff.net.0 with the GEGLU in the GEMM epilogue (dgq_gemm_extra_t.geglu) on the SD / SDXL shapes: quantise-on-load + GEMM, us per layer
(hipGraph replay of 20 calls).  usage: python tools/bench_geglu.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act
dev = torch.device("cuda:0")
for (M, K, N) in ((8192, 320, 2560), (2048, 640, 5120), (512, 1280, 10240), (8192, 1280, 10240)):
    for mode in ("perK", "perM"):
        g = torch.Generator().manual_seed(M + N)
        x = (torch.randn(M, K, generator=g) * 1.2).to(dev)
        w = torch.randn(N, K, generator=g) * 0.05
        wd, wz = synth.channel_minmax(w, 4)
        if mode == "perK":
            d, z = synth._group_params(K, 16, 8, "geglu|%d" % N, 0)
            lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "linear", K, 1, 8)
        else:
            d, z = synth._group_params(min(M, 1024), 16, 8, "geglu|%d" % N, 0)
            lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "linear", K, 1, 8)
        rp = torch.stack([torch.arange(N // 2), torch.arange(N // 2) + N // 2], 1).flatten()
        pwi = ops.PackedWeight(w[rp].to(dev), wd[rp].to(dev), wz[rp].to(dev), None, torch.zeros(N, device=dev), 4, K, 1)
        ab = ops.ActBinding(lay, pwi, 8)
        xx = x.view(M // min(M, 1024), min(M, 1024), K)
        f = lambda: ops.quant_linear(xx, ab, geglu=True)
        for _ in range(3): f()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20): f()
        gr.replay(); torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
        print("%5d x %5d x %5d %s  %7.1f us (quantise + GEMM with GEGLU epilogue)" % (M, N, K, mode, best), flush=True)
