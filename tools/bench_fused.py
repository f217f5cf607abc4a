"""dgq_linear_fused_batch against dgq_quant_act + dgq_gemm_wxa8 per Linear shape of the SD1.4 step (hipGraph replay of 20
calls -> us per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act

dev = torch.device("cuda:0")
iters = 20


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


shapes = [(8192, 320, 320, "perK", "ln"), (8192, 320, 320, "perM", "ln"), (8192, 320, 2560, "perK", "ln"), (8192, 1280, 320, "perK", "geglu"),
          (2048, 640, 640, "perK", "ln"), (2048, 640, 640, "perM", "ln"), (2048, 640, 5120, "perK", "ln"), (512, 1280, 1280, "perK", "ln")]
gen = torch.Generator().manual_seed(0)
for M, K, N, mode, pro in shapes:
    x = torch.randn(M, 2 * K if pro == "geglu" else K, generator=gen).to(dev)
    w = torch.randn(N, K, generator=gen) * 0.05
    wd, wz = synth.channel_minmax(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, K, 1)
    if mode == "perK":
        d, z = synth._group_params(K, 16, 8, "bf", 0)
        lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "linear", K, 1, 8)
    else:
        d, z = synth._group_params(M // 2, 16, 8, "bf", 0)
        lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "linear", K, 1, 8)
    ab = ops.ActBinding(lay, pw, 8)
    ln = (torch.ones(K, device=dev), torch.zeros(K, device=dev), 1e-5) if pro == "ln" else None
    pa = 2 if pro == "geglu" else 0
    ops.FUSED_LINEAR = 0
    t_sep = timed(lambda: ops.quant_linear(x, ab, pre_act=pa, ln=ln))
    ops.FUSED_LINEAR, ops.FUSED_MIN_M = 2, 1
    ts = []
    for dbg in ("0", "1", "2", "3"):
        os.environ["DGQ_FUSED_DEBUG"] = dbg
        ts.append(timed(lambda: ops.quant_linear(x, ab, pre_act=pa, ln=ln)))
    os.environ["DGQ_FUSED_DEBUG"] = "0"
    t_fused = ts[0]
    print("%5d x %4d -> %5d %s %-5s Kp=%4d | quant+gemm %6.1f us | fused %6.1f us (quantise phase only %.1f, without it %.1f, without the tile loop %.1f)" % (M, K, N, mode, pro or "", ab.Kp, t_sep, t_fused, ts[1], ts[2], ts[3]), flush=True)
