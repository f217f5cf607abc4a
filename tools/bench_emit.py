"""Row-owning GEMM + quantise-on-store (dgq_gemm_wxa8_emit) against the two launches it replaces (dgq_gemm_wxa8, then dgq_quant_act
behind a LayerNorm) on the SD1.4 shapes it applies to; hipGraph replay, us per pair.   usage: python tools/bench_emit.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def binding(N, K, mode, name, T=64):
    g = torch.Generator().manual_seed(len(name))
    w = torch.randn(N, K, generator=g) * 0.05
    wd, wz = synth.channel_minmax(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, K, 1)
    if mode == "perK":
        d, z = synth._group_params(K, 16, 8, "be|" + name, 0)
        lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "linear", K, 1, 8)
    else:
        d, z = synth._group_params(T, 16, 8, "be|" + name, 0)
        lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "linear", K, 1, 8)
    return ops.ActBinding(lay, pw, 8)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


print("%-34s %-6s %-6s %5s %10s %10s %10s %10s %10s" % ("M x N x K (producer)", "prod", "cons", "ncons", "gemm us", "quant us", "pair us", "emit us", "its GEMM"))
for (M, N, K) in ((8192, 320, 320), (8192, 320, 1280), (2048, 640, 640), (2048, 640, 2560)):
    for pmode in ("perK", "perM"):
        for cmode, ncons in (("perK", 1), ("perM", 1)):
            prod = binding(N, K, pmode, "p%d%d%s" % (N, K, pmode))
            cons = [(binding(N, N, cmode, "c%d%s%d" % (N, cmode, i)), None) for i in range(ncons)]
            ln = (torch.ones(N, device=dev), torch.zeros(N, device=dev), 1e-5)
            cons = [(ab, ln) for ab, _ in cons]
            codes = torch.randint(-128, 128, (M, prod.Kp), dtype=torch.int8, device=dev)
            rowsum = torch.randn(1, M, device=dev)
            res = torch.randn(M, N, device=dev)
            ex = ops.make_extra(res)
            y = ops.gemm_wxa8(codes, rowsum, M, prod, torch.float32, extra=ex)
            t_g = timed(lambda: ops.gemm_wxa8(codes, rowsum, M, prod, torch.float32, out=y, extra=ex))
            if ncons == 1:
                t_q = timed(lambda: ops.quant_act(y, M, 1, 1, N, 1, 1, 1, 0, cons[0][0], None, ln))
            else:
                t_q = timed(lambda: ops.quant_linear_multi(y, [c[0] for c in cons], ln=ln)) - timed(lambda: None)
            t_e = timed(lambda: ops.gemm_wxa8_emit(codes, rowsum, M, prod, torch.float32, cons, extra=ex))
            os.environ["DGQ_EMIT_SKIP"] = "1"
            t_e0 = timed(lambda: ops.gemm_wxa8_emit(codes, rowsum, M, prod, torch.float32, cons, extra=ex))
            os.environ.pop("DGQ_EMIT_SKIP")
            note = " (quant figure includes the 3 consumer GEMMs)" if ncons > 1 else ""
            print("%-34s %-6s %-6s %5d %10.1f %10.1f %10.1f %10.1f %10.1f%s" % ("%d x %d x %d" % (M, N, K), pmode, cmode, ncons, t_g, t_q, t_g + t_q, t_e, t_e0, note), flush=True)
