"""One per-M GEMM shape with and without the residual epilogue, per launch plan (DGQ_GEMM_FORCE): hipGraph replay, us per launch.
usage: bench_gemm_res.py M N K [plan ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act
dev = torch.device("cuda:0")
M, N, K = (int(v) for v in sys.argv[1:4])
plans = sys.argv[4:] or ["default", "64,128,1", "128,128,1", "256,256,1"]
w = torch.randn(N, K) * 0.05
wd, wz = synth.channel_minmax(w, 4)
pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, K, 1)
lay = plan_act(torch.tensor(0.03), torch.tensor(120.0), "linear", K, 1, 8)
ab = ops.ActBinding(lay, pw, 8)
codes = torch.randint(-128, 128, (M, ab.Kp), dtype=torch.int8, device=dev)
rowsum = torch.randn(M, device=dev)
res = torch.randn(M, N, device=dev)
out = torch.empty(M, N, device=dev)


def timed(f, iters=20):
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


for plan in plans:
    if plan == "default": os.environ.pop("DGQ_GEMM_FORCE", None)
    else: os.environ["DGQ_GEMM_FORCE"] = plan
    a = timed(lambda: ops.gemm_wxa8(codes, rowsum, M, ab, torch.float32, out))
    b = timed(lambda: ops.gemm_wxa8(codes, rowsum, M, ab, torch.float32, out, extra=ops.make_extra(res)))
    print("%d x %d x %d  plan %-10s  plain %7.1f us   + residual %7.1f us" % (M, N, K, plan, a, b), flush=True)
