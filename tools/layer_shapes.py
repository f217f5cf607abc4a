"""Per-shape cost table of one SD1.4 W4A8 g16 denoise step: every quantized layer's quantise-on-load pass and GEMM
re-timed in isolation (hipGraph replay of 10 launches), aggregated by (M, N, Kp, taps, mode).  GPU box only."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops
from dgq_amd.runtime import build_synthetic_qnn
from dgq_amd import synth
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
qnn, _ = build_synthetic_qnn("sd", bench.CONFIGS["c2"]["cfg"], 64, 2, 1, device=dev)
qnn.prepare_slots([0])
lat = synth.named_randn("latent", (2, 4, 64, 64), 1).to(dev)
ctx = synth.named_randn("ctx", (2, 77, 768), 100).to(dev)
ITERS = 10


def replay_us(fn):
    for _ in range(2):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(ITERS):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / ITERS


rows = collections.OrderedDict()
orig_g, orig_q = ops.gemm_wxa8, ops.quant_act


def q_wrapped(x, B, H, W, C, kh, kw, stride, pad, ab, *a, **k):
    r = orig_q(x, B, H, W, C, kh, kw, stride, pad, ab, *a, **k)
    pre, ln = k.get("pre"), k.get("ln")
    q_wrapped.pro = ("gn" if (pre and pre[0] is not None) else "") + ("ln" if ln else "") + \
        ({0: "", 1: "+silu", 2: "+geglu"}[pre[2]] if pre else "")
    us = replay_us(lambda: orig_q(x, B, H, W, C, kh, kw, stride, pad, ab, *a, **k))
    q_wrapped.last = us
    return r


def g_wrapped(codes, rowsum, M, ab, out_dtype, out=None, extra=None):
    y = orig_g(codes, rowsum, M, ab, out_dtype, out, extra)
    us = replay_us(lambda: orig_g(codes, rowsum, M, ab, out_dtype, y, extra))
    key = (M, ab.pw.N, ab.Kp, ab.pw.taps, ab.mode)
    r = rows.setdefault(key, [0, 0.0, 0.0, 2.0 * M * ab.pw.N * ab.pw.K, q_wrapped.pro])
    r[0] += 1; r[1] += q_wrapped.last; r[2] += us
    return y


ops.quant_act, ops.gemm_wxa8 = q_wrapped, g_wrapped
with torch.no_grad():
    qnn(lat, 981, ctx)
torch.cuda.synchronize()
print("%6s %5s %6s %4s %6s %4s %9s %9s %9s %9s %8s %s" % ("M", "N", "Kp", "taps", "mode", "n", "qact_us", "gemm_us", "qact_tot", "gemm_tot", "TOP/s", "prologue"))
tq = tg = 0
for k, r in sorted(rows.items(), key=lambda kv: -kv[1][2]):
    tq += r[1]; tg += r[2]
    print("%6d %5d %6d %4d %6s %4d %9.1f %9.1f %9.1f %9.1f %8.1f %s" % (*k, r[0], r[1] / r[0], r[2] / r[0], r[1], r[2], r[3] * r[0] / r[2] / 1e6, r[4]))
print("total quant_act %.1f us, gemm %.1f us" % (tq, tg))
