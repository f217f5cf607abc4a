"""dgq_linear_smallm_batch on the 23 time_emb_proj layers of an SD step (M = 2·prompts rows, K = 1280): hipGraph replay, us per launch.
usage: bench_smallm.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act
dev = torch.device("cuda:0")
K = 1280
binds = []
g = torch.Generator().manual_seed(0)
for N in [320] * 5 + [640] * 6 + [1280] * 12:
    w = torch.randn(N, K, generator=g) * 0.05
    wd, wz = synth.channel_minmax(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, K, 1)
    binds.append(ops.ActBinding(plan_act(torch.tensor(0.03), torch.tensor(120.0), "linear", K, 1, 8), pw, 8))
for M in [int(a) for a in sys.argv[1:]] or [1, 2, 8, 16]:
    x = torch.randn(M, K, device=dev)
    f = lambda: ops.linear_smallm_batch(x, binds, pre_act=1)
    for _ in range(3): f()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10): f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    print("M = %2d: %.1f us per launch (23 layers, 12 MB of int4 weights)" % (M, e0.elapsed_time(e1) * 1e3 / 10), flush=True)
