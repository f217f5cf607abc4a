"""Where does a per-K launch of dgq_gemm_wxa8 spend more than its per-M twin?  Same shape, same Kp (natural order), tables of
1 / 4 / 16 groups laid out contiguously (no padding: K % (32·G) == 0), against the per-M launch — hipGraph replay.
usage: python tools/gemm_gap.py [M N K] ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act

dev = torch.device("cuda:0")
ITERS = 20


def replay_us(fn):
    for _ in range(2):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(ITERS):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / ITERS)
    return best


shapes = [(512, 1280, 1280), (2048, 640, 640), (8192, 320, 320), (2048, 640, 2560)]
if len(sys.argv) > 3:
    v = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
for (M, N, K) in shapes:
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) * 0.05
    wd, wz = synth.channel_minmax(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, K, 1)
    out = torch.empty(M, N, device=dev)
    res = []
    for G in (0, 1, 4, 16, -16):
        if G == 0:
            lay = plan_act(torch.rand(1, 64, 1) + 0.1, torch.zeros(1, 64, 1), "linear", K, 1, 8)
        else:
            n = abs(G)
            if G > 0:       # contiguous equal groups: chunk aligned, no padding
                lab = torch.arange(K) // (K // n)
            else:           # every 32-chunk its own (δ, z): a flush per chunk
                lab = torch.arange(K) // 32
            d = 0.01 + 0.001 * lab.float()
            lay = plan_act(d.view(1, 1, -1), torch.zeros(1, 1, K), "linear", K, 1, 8)
        ab = ops.ActBinding(lay, pw, 8)
        codes = torch.randint(-128, 128, (M, ab.Kp), dtype=torch.int8, device=dev)
        rowsum = torch.randn(M, device=dev)
        t = replay_us(lambda: ops.gemm_wxa8(codes, rowsum, M, ab, torch.float32, out))
        res.append("%s Kp=%d: %.1f" % ("perM" if G == 0 else ("perK g%d" % G if G > 0 else "perK flush-per-chunk"), ab.Kp, t))
    print("%5d x %5d x %5d | " % (M, N, K) + " | ".join(res), flush=True)
