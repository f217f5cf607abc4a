"""Where the conv-with-quantiser-inside launch (csrc/gemm_convq.hip) spends its time (GPU box; needs `make -C dgq_amd/csrc diag`).
One launch of a C = 320 3x3 layer replayed from a graph out of the stamped library; every wave's stamps (csrc/diag.h) come back through
dgq_diag_fetch_convq.  A stamped build's LENGTH is not the product's: read the shares.
usage: python tools/convq_timeline.py ["B,C,H,W,N" ...]  > profiles/r06_convq_timeline.txt"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DGQ_HIP_LIB", os.path.join(ROOT, "dgq_amd", "csrc", "libdgq_hip_diag.so"))
import numpy as np
import torch
from dgq_amd import ops, synth, _lib
from dgq_amd.plan import plan_act
dev = torch.device("cuda:0")
SLOTS, WAVES = 16, 1 << 16
lib = _lib.load()
shapes = [(2, 320, 64, 64, 320)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]


def fetch():
    buf = np.zeros(WAVES * SLOTS, dtype=np.uint64)
    assert lib.dgq_diag_fetch_convq(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes)) == 0
    r = buf.reshape(WAVES, SLOTS)
    return r[r[:, 0] != 0].astype(np.int64)


def med(a):
    return np.percentile(a, 50), np.percentile(a, 10), np.percentile(a, 90)


print("# tools/convq_timeline.py — stamped build (csrc/diag.h); cycles are shader cycles of the stamped run: read the SHARES")
for B, C, H, W, N in shapes:
    gen = torch.Generator().manual_seed(1)
    w = (torch.randn(N, C, 3, 3, generator=gen) * 0.05).to(dev)
    x = (torch.randn(B, C, H, W, generator=gen) * 1.3).to(dev).contiguous(memory_format=torch.channels_last)
    wd, wz = synth.channel_minmax(w.cpu(), 4)
    pw = ops.PackedWeight(w, wd.to(dev), wz.to(dev), None, torch.randn(N, generator=gen).to(dev), 4, C, 9)
    res = torch.randn(B, N, H, W, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    sc, sh = torch.rand(B, C, device=dev) + 0.5, torch.randn(B, C, device=dev) * 0.1
    ops.groupnorm_scale_shift = lambda *a, **k: (sc, sh)
    for mode in ("perK", "perM"):
        if mode == "perK":
            d, z = synth._group_params(C * 9, 16, 8, "cq|%d" % C, 0)
            lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "conv", C, 9, 8)
        else:
            d, z = synth._group_params(H * W, 16, 8, "cq|%d" % C, 0)
            lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "conv", C, 9, 8)
        ab = ops.ActBinding(lay, pw, 8)
        f = lambda: ops.quant_conv2d(x, ab, 3, 3, 1, 1, norm=(32, 1e-5, None, None, 1), residual=res)
        for _ in range(3): f()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): f()
        g.replay(); torch.cuda.synchronize()
        assert lib.dgq_diag_clear_convq() == 0
        g.replay(); torch.cuda.synchronize()
        r = fetch()
        print("== B=%d C=%d %dx%d -> N=%d %s (M=%d, Kp=%d): %d waves" % (B, C, H, W, N, mode, B * H * W, ab.Kp, len(r)))
        if not len(r):
            print("   no records (the layer did not take the one-launch form)"); continue
        e0 = r[:, 1].min()
        entry, exit_ = (r[:, 1] - e0) * 0.01, (r[:, 11] - e0) * 0.01
        cyc = (r[:, 10] - r[:, 0]).astype(np.float64)
        clk = np.median(cyc / (np.maximum(r[:, 11] - r[:, 1], 1) * 0.01)) / 1e3
        print("   shader clock %.2f GHz; one time axis (us): last wave entered %.2f (p50 %.2f) | first exit %.2f, p50 exit %.2f, last exit %.2f"
              % (clk, entry.max(), np.median(entry), exit_.min(), np.median(exit_), exit_.max()))
        tot = np.median(cyc)
        kloop = (r[:, 6] - r[:, 4]) - r[:, 7] - r[:, 12] - r[:, 13]
        rows = [("entry -> W + table loads issued, patch staged", r[:, 3] - r[:, 0]),
                ("gather table / chunk tables -> LDS, barrier", r[:, 4] - r[:, 3]),
                ("slab loop: this wave's rows quantised (all slabs)", r[:, 7]),
                ("slab loop: waiting for the other waves' rows", r[:, 13]),
                ("slab loop: waiting for the slab image to be free", r[:, 12]),
                ("slab loop: K tiles (MFMA)", kloop),
                ("row sums, barrier, tile transposed in LDS", r[:, 8] - r[:, 6]),
                ("epilogue stores issued", r[:, 9] - r[:, 8]),
                ("stores issued -> done", r[:, 10] - r[:, 9]),
                ("whole wave", r[:, 10] - r[:, 0])]
        for name, d in rows:
            m, lo, hi = med(d.astype(np.float64))
            print("   %-52s %8.0f cyc  (p10 %6.0f, p90 %6.0f) = %5.2f us  %5.1f %%" % (name, m, lo, hi, m / clk / 1e3, 100.0 * m / tot))
