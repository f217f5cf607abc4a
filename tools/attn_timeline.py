"""Where the three kernels of one fused attention call spend their time (GPU box; needs `make -C dgq_amd/csrc diag`): the s_memtime /
s_memrealtime stamps of attn3_prep / attn3_stats / attn3_pv (csrc/diag.h) for the SD step's attention shapes, reduced like
tools/small_launch_timeline.py.  A stamped build's LENGTH is not the product's: read the shares.
usage: python tools/attn_timeline.py ["D,T,S" ...]  > profiles/r05_attention_timeline.txt"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DGQ_HIP_LIB", os.path.join(ROOT, "dgq_amd", "csrc", "libdgq_hip_diag.so"))
import numpy as np
import torch
from dgq_amd import ops, _lib

dev = torch.device("cuda:0")
SLOTS, WAVES = 16, 1 << 16
lib = _lib.load()
NAMES = {1: "pre-pass (aqtizer_k / aqtizer_v, K / V tile images, Q codes)", 2: "statistics pass (row maxima, l, real-time δ)", 3: "P·V pass"}
SEGS = {1: [("whole wave", 0, 10), ("  of it: stores issued -> done", 9, 10)],
        2: [("entry -> Q fragments + first K tiles issued", 0, 3), ("issued -> tile 0 landed, barrier", 3, 4), ("key loop", 4, 5),
            ("merge halves, store statistics, δ maximum", 5, 9), ("stores issued -> done", 9, 10), ("whole wave", 0, 10)],
        3: [("entry -> Q fragments, statistics, first tiles issued", 0, 3), ("issued -> landed, barrier", 3, 4), ("key loop", 4, 5),
            ("V tables staged (2 barriers)", 5, 6), ("output stores issued", 6, 9), ("stores issued -> done", 9, 10), ("whole wave", 0, 10)]}


def fetch():
    buf = np.zeros(WAVES * SLOTS, dtype=np.uint64)
    assert lib.dgq_diag_fetch_attn(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes)) == 0
    r = buf.reshape(WAVES, SLOTS)
    return r[r[:, 0] != 0].astype(np.int64)


def run(D, T, S, B=2, H=8, bits=8):
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, T, H * D, generator=g).to(dev)
    k = torch.randn(B, S, H * D, generator=g).to(dev)
    v = torch.randn(B, S, H * D, generator=g).to(dev)
    # per-token q / k quantizers (int8 scores) and a per-head-dim v quantizer, log2 softmax quantiser with the real-time δ
    mk = lambda n: (torch.full((n,), 0.05, device=dev), torch.full((n,), 128.0, device=dev))
    dq, zq = mk(T); dk, zk = mk(S); dv, zv = mk(D)
    fq = ((1, dq, zq, 0, 8), (1, dk, zk, 0, 8), (2, dv, zv, 0, 8))
    delta = torch.zeros(1, device=dev)
    call = lambda: ops.attention(q, k, v, H, D, D ** -0.5, 1, 0, delta, bits, fq=fq)
    for _ in range(2):
        call()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        call()
    gr.replay(); torch.cuda.synchronize()
    assert lib.dgq_diag_clear_attn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    r = fetch()
    print("== attention D = %d, T = %d, S = %d, B·H = %d: %.1f us for the call (stamped build)" % (D, T, S, B * H, e0.elapsed_time(e1) * 1e3))
    t0 = r[:, 1].min()
    for tag in (1, 2, 3):
        rr = r[r[:, 13] == tag]
        if len(rr) == 0:
            continue
        clk = np.median((rr[:, 10] - rr[:, 0]) / np.maximum((rr[:, 11] - rr[:, 1]) * 0.01, 0.01)) / 1e3
        print("  %s: %d waves, clock %.2f GHz; on the call's time axis: first entry %.2f us, last entry %.2f, last exit %.2f" % (
            NAMES[tag], len(rr), clk, (rr[:, 1].min() - t0) * 0.01, (rr[:, 1].max() - t0) * 0.01, (rr[:, 11].max() - t0) * 0.01))
        tot = np.median(rr[:, 10] - rr[:, 0])
        for name, a, b in SEGS[tag]:
            d = (rr[:, b] - rr[:, a]).astype(np.float64)
            print("    %-56s %8.0f cyc (p10 %6.0f, p90 %6.0f) = %5.2f us %5.1f %%" % (name, np.median(d), np.percentile(d, 10), np.percentile(d, 90),
                                                                                    np.median(d) / clk / 1e3, 100 * np.median(d) / tot))
        if tag == 3:
            print("    output stores: dequantise + LDS writes %.0f cyc, LDS reads + global stores %.0f cyc (medians)" % (np.median(rr[:, 7]), np.median(rr[:, 12])))


shapes = sys.argv[1:] or ["160,256,256", "160,256,77", "80,1024,1024", "80,1024,77", "40,4096,77", "160,64,64", "40,4096,4096"]
print("# tools/attn_timeline.py — stamped build (csrc/diag.h); read the SHARES")
for s in shapes:
    D, T, S = (int(x) for x in s.split(","))
    run(D, T, S)
