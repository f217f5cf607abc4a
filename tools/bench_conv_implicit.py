"""Scalar-δ 3x3 convolutions (config C5 shapes): quantise pass + GEMM, materialised operand against the implicit-im2col GEMM, under
several tile plans (DGQ_GEMM_FORCE, read per call).   usage: python tools/bench_conv_implicit.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
SHAPES = [(8, 320, 128, 128, 320), (8, 640, 64, 64, 640), (8, 1280, 32, 32, 1280), (8, 1920, 32, 32, 1280), (8, 960, 64, 64, 640), (8, 640, 128, 128, 320)]
if os.environ.get("ONLY_SHAPE"):
    SHAPES = [SHAPES[int(os.environ["ONLY_SHAPE"])]]
PLANS = os.environ.get("PLANS", "default;64,64,1;64,128,1;128,128,1").split(";")
print("%-28s %-10s %-10s %10s %10s %10s" % ("B,C,H,W->N", "operand", "plan", "quant us", "gemm us", "total us"))
for (B, C, H, W, N) in SHAPES:
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(N, C, 3, 3, generator=g) * 0.03).to(dev)
    wd, wz = synth.channel_minmax(w.cpu(), 4)
    pw = ops.PackedWeight(w, wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, C, 9)
    ab = ops.ActBinding(plan_act(torch.tensor(0.05), torch.tensor(31.0), "conv", C, 9, 6), pw, 6)
    x = torch.randn(B, C, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    for implicit in (False, True):
        ops.CONV_IMPLICIT = implicit
        for plan in PLANS:
            if plan == "default":
                os.environ.pop("DGQ_GEMM_FORCE", None)
            else:
                os.environ["DGQ_GEMM_FORCE"] = plan
            tq, tg = [], []
            ops.QUANT_LAUNCH_HOOK = lambda issue, by: tq.append(issue)
            ops.GEMM_LAUNCH_HOOK = lambda issue, pr: tg.append(issue)
            ops.quant_conv2d(x, ab, 3, 3, 1, 1)
            ops.QUANT_LAUNCH_HOOK = ops.GEMM_LAUNCH_HOOK = None
            res = []
            for fn in (tq[0], tg[0]):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) * 1e3 / iters)
            print("%-28s %-10s %-10s %10.1f %10.1f %10.1f" % ("%d,%d,%d,%d->%d" % (B, C, H, W, N), "implicit" if implicit else "unfolded", plan, res[0], res[1], sum(res)), flush=True)
    os.environ.pop("DGQ_GEMM_FORCE", None)
