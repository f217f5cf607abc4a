"""What does a short launch pay for COLD CODE and for COLD DATA?  (round 5: a replayed launch does not predict its in-step time)

For each target launch (real library kernels at in-step shapes) four hipGraphs of R iterations are timed with events:
    hot        target x R back to back                           (its code in the instruction caches, its data in L2)
    code-cold  (target + ~40 distinct small torch kernels) x R    (the other kernels' code evicts the 64 KB instruction caches;
                                                                  their data is a 1 KB tensor) minus the same graph without the target
    data-cold  (target + a 384 MB copy) x R  minus copies alone   (L2 and Infinity Cache swept; the copy kernel is a few hundred bytes)
    both       (target + small kernels + copy) x R  minus the rest
The differences are what ONE target launch adds to the chain in that machine state.
usage: python tools/cold_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act

dev = torch.device("cuda:0")
R = 12
small = torch.rand(256, device=dev) * 0.5 + 0.25
big_a = torch.empty(96 << 20, device=dev, dtype=torch.float32)        # 384 MB
big_b = torch.empty_like(big_a)
UNARY = [torch.sin, torch.cos, torch.exp, torch.log, torch.tanh, torch.sigmoid, torch.erf, torch.sqrt, torch.rsqrt, torch.abs, torch.neg,
         torch.floor, torch.ceil, torch.round, torch.trunc, torch.relu, torch.atan, torch.asin, torch.acos, torch.sinh, torch.cosh,
         torch.expm1, torch.log1p, torch.log2, torch.log10, torch.exp2, torch.reciprocal, torch.sign, torch.square, torch.erfinv,
         torch.lgamma, torch.digamma, torch.erfc, torch.tan, torch.asinh, torch.atanh, torch.frac, torch.i0, torch.special.i1,
         torch.special.ndtri, torch.special.erfcx, torch.nn.functional.gelu, torch.nn.functional.silu, torch.nn.functional.softplus,
         torch.nn.functional.mish, torch.nn.functional.hardswish]


def evict_code():
    for f in UNARY:
        f(small)


def evict_data():
    big_b.copy_(big_a)


def graph_us(fns):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(R):
            for f in fns:
                f()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / R)
    del g
    return best


def binding(N, C, taps, mode, rows):
    K = C * taps
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) * 0.05
    wd, wz = synth.channel_minmax(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, C, taps)
    if mode == "perK":
        d, z = synth._group_params(K, 16, 8, "probe|%d|%d" % (N, K), 0)
        lay = plan_act(d.view(1, -1, 1) if taps > 1 else d.view(1, 1, -1), z.view(1, -1, 1) if taps > 1 else z.view(1, 1, -1),
                       "conv" if taps > 1 else "linear", C, taps, 8)
    else:
        d, z = synth._group_params(rows, 16, 8, "probe|%d|%d" % (N, K), 0)
        lay = plan_act(d.view(1, 1, -1) if taps > 1 else d.view(1, -1, 1), z.view(1, 1, -1) if taps > 1 else z.view(1, -1, 1),
                       "conv" if taps > 1 else "linear", C, taps, 8)
    return ops.ActBinding(lay, pw, 8)


targets = []
# GroupNorm finaliser (B = 2, 16 x 16, C = 1280): the smallest launch of the step
part = torch.randn(2 * 16, 1280, 2, device=dev).abs()
gam, bet = torch.randn(1280, device=dev), torch.randn(1280, device=dev)
gn = {"parts": [(part, 1280)], "B": 2, "HW": 256, "C": 1280}
targets.append(("gn_from_partials 2x256x1280", lambda: ops.groupnorm_from_partials(gn, 32, 1e-5, gam, bet)))
# quantise-on-load + GEMM of a Linear at the 16 x 16 level
for mode in ("perM", "perK"):
    ab = binding(1280, 1280, 1, mode, 256)
    x = torch.randn(2, 256, 1280, device=dev)
    M = 512
    codes, rs, _ = ops.quant_act(x.view(2, 16, 16, 1280), 2, 16, 16, 1280, 1, 1, 1, 0, ab)
    out = torch.empty(M, 1280, device=dev)
    targets.append(("quant_act 512x1280 %s" % mode, lambda ab=ab, x=x: ops.quant_act(x.view(2, 16, 16, 1280), 2, 16, 16, 1280, 1, 1, 1, 0, ab)))
    targets.append(("gemm 512x1280x1280 %s" % mode, lambda ab=ab, codes=codes, rs=rs, out=out: ops.gemm_wxa8(codes, rs, 512, ab, torch.float32, out)))
# small attention (three launches)
for (D, T, S) in ((160, 256, 77), (40, 4096, 77)):
    q, k, v = (torch.randn(2, n, 8 * D, device=dev) for n in (T, S, S))
    tab = lambda n: (torch.rand(n, device=dev) * 0.02 + 0.02, torch.randint(100, 156, (n,), device=dev).float())
    fq = ((2,) + tab(D) + (0, 8), (1,) + tab(S - 1) + (1, 8), (2,) + tab(D) + (0, 8))
    targets.append(("attention D=%d T=%d S=%d (3 launches)" % (D, T, S), lambda q=q, k=k, v=v, D=D, fq=fq: ops.attention(q, k, v, 8, D, D ** -0.5, 1, 1, None, 8, fq)))

# ---- under rocprofv3 --kernel-trace (COLD_PROBE_TRACE=1): no event arithmetic — every pattern is replayed and the target kernels' own
# durations are read from the trace by tools/cold_probe_trace.py, which tells the machine state of a target dispatch from the marker
# kernel that precedes it: logaddexp = hot run starts, hardswish = code evicted, copy = data evicted, relu6 = both.
def mark_hot():
    torch.logaddexp(small, small)


def mark_both():
    torch.nn.functional.relu6(small)


if os.environ.get("COLD_PROBE_TRACE"):
    for name, f in targets:
        for pat in ([mark_hot] + [f] * 6, [f, evict_code], [f, evict_data], [f, evict_data, evict_code, mark_both]):
            graph_us(pat)
    print("trace patterns done")
    sys.exit(0)

base_code, base_data, base_both = graph_us([evict_code]), graph_us([evict_data]), graph_us([evict_code, evict_data])
print("evictors alone: code %.1f us (%d kernels), data %.1f us, both %.1f us per iteration" % (base_code, len(UNARY), base_data, base_both))
print("%-44s %8s %10s %10s %10s" % ("target", "hot", "code-cold", "data-cold", "both"))
for name, f in targets:
    hot = graph_us([f])
    cc = graph_us([f, evict_code]) - base_code
    dc = graph_us([f, evict_data]) - base_data
    bc = graph_us([f, evict_code, evict_data]) - base_both
    print("%-44s %8.2f %10.2f %10.2f %10.2f" % (name, hot, cc, dc, bc), flush=True)
