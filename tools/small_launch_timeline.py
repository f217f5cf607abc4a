"""Where a SHORT launch of the SD step spends its time (GPU box; needs the diagnostic build: `make -C dgq_amd/csrc diag`).

For each shape one quantise-on-load launch + one GEMM launch of a Linear layer run from the stamped library
(DGQ_HIP_LIB=dgq_amd/csrc/libdgq_hip_diag.so, set here); every wave's s_memtime / s_memrealtime stamps (csrc/diag.h) come back
through dgq_diag_fetch_* and are reduced to
  * the launch on one time axis (100 MHz clock): first workgroup entry -> last entry (dispatch spread) -> last exit,
  * per-wave segment lengths in shader cycles (median, p10, p90 over all waves).
A stamped build's LENGTH is not the product's (the fences around a stamp forbid overlaps): read the shares.
usage: python tools/small_launch_timeline.py ["M,N,K,mode" ...]  > profiles/r05_small_launch_timeline.txt"""
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

GEMM_SEG = [("entry -> tables + first tiles issued", 0, 3), ("issued -> tile 0 + tables landed", 3, 4),
            ("landed -> tables staged, barrier (loop start)", 4, 5), ("K loop", 5, 6),
            ("loop end -> tile transposed in LDS", 6, 8), ("transposed -> stores issued", 8, 9),
            ("stores issued -> stores done", 9, 10), ("whole wave", 0, 10)]
QA_SEG = [("entry -> addresses, row statistics (LayerNorm)", 0, 3), ("load + quantise + store loop", 3, 4),
          ("row sum reduce + store issue", 4, 9), ("stores issued -> done", 9, 10), ("whole wave", 0, 10)]
PANEL_SEG = [("entry -> panel DMA + W ring + table loads issued", 0, 3), ("issued -> everything landed (one wait)", 3, 4),
             ("landed -> tables staged, barrier (loop start)", 4, 5), ("K loop", 5, 6),
             ("loop end -> barrier, tile transposed in LDS", 6, 8), ("transposed -> stores issued", 8, 9),
             ("stores issued -> stores done", 9, 10), ("whole wave", 0, 10)]
FUSED_SEG = [("entry -> W ring + table loads issued", 0, 3), ("chunk tables staged, panel zeroed, barrier (per-K)", 3, 14),
             ("row statistics (LayerNorm; last pass of the rows)", 14, 15), ("load + quantise + write the panel, then all loads landed", 15, 4),
             ("landed -> tables staged, barrier (loop start)", 4, 5), ("K loop", 5, 6),
             ("loop end -> barrier, tile transposed in LDS", 6, 8), ("transposed -> stores issued", 8, 9),
             ("stores issued -> stores done", 9, 10), ("whole wave", 0, 10)]
SC_SEG = [("entry -> tables staged, image zeroed, barrier", 0, 3), ("row statistics (LayerNorm)", 3, 4),
          ("load + quantise + scatter into the LDS image", 4, 5), ("reduce + barrier", 5, 6),
          ("image -> global stores issued", 6, 9), ("stores issued -> done", 9, 10), ("whole wave", 0, 10)]


def fetch(name):
    buf = np.zeros(WAVES * SLOTS, dtype=np.uint64)
    rc = getattr(lib, "dgq_diag_fetch_" + name)(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
    assert rc == 0
    r = buf.reshape(WAVES, SLOTS)
    return r[r[:, 0] != 0]


def clear():
    for n in ("gemm", "quant", "panel"):
        assert getattr(lib, "dgq_diag_clear_" + n)() == 0


def pct(a):
    return np.percentile(a, 50), np.percentile(a, 10), np.percentile(a, 90)


def report(title, r, segs, extra=None):
    if len(r) == 0:
        print("  %s: no records" % title)
        return
    r = r.astype(np.int64)
    e0 = r[:, 1].min()
    entry = (r[:, 1] - e0) * 0.01           # us on the shared 100 MHz clock
    exit_ = (r[:, 11] - e0) * 0.01
    cyc = (r[:, 10] - r[:, 0]).astype(np.float64)
    real = np.maximum((r[:, 11] - r[:, 1]).astype(np.float64), 1.0) * 0.01
    clk = np.median(cyc / real) / 1e3        # GHz
    xcc = (r[:, 2] >> 32) & 0xF
    print("  %s: %d waves; shader clock %.2f GHz" % (title, len(r), clk))
    print("    one time axis (us from the first wave's entry): last wave entered %.2f (p50 %.2f, p90 %.2f) | first exit %.2f, "
          "p50 exit %.2f, last exit %.2f" % (entry.max(), np.median(entry), np.percentile(entry, 90), exit_.min(), np.median(exit_), exit_.max()))
    print("    waves per XCD: %s" % " ".join("%d" % int((xcc == i).sum()) for i in range(8)))
    tot = np.median(cyc)
    for name, a, b in segs:
        d = (r[:, b] - r[:, a]).astype(np.float64)
        m, lo, hi = pct(d)
        print("    %-48s %8.0f cyc  (p10 %6.0f, p90 %6.0f) = %5.2f us  %5.1f %%" % (name, m, lo, hi, m / clk / 1e3, 100.0 * m / tot))
    if extra:
        extra(r, clk, tot)


def gemm_extra(r, clk, tot):
    nk = r[:, 13] & 0xFFFF
    w = r[:, 7].astype(np.float64)
    v = r[:, 12].astype(np.float64)
    print("    K loop detail: %d K tiles; per tile %.0f cyc, of which counted-wait %.0f + barrier %.0f (median over waves)" % (
        int(np.median(nk)), np.median((r[:, 6] - r[:, 5]) / np.maximum(nk, 1)), np.median(v / np.maximum(nk, 1)),
        np.median((w - v) / np.maximum(nk, 1))))


def fused_extra(r, clk, tot):
    print("    quantise rounds: waiting for the round's loads %.0f cyc, arithmetic + panel writes %.0f cyc (median over waves)" % (
        np.median(r[:, 12]), np.median(r[:, 7])))


def run(M, N, K, mode, ln):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) * 0.05
    wd, wz = synth.channel_minmax(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, K, 1)
    if mode == "perK":
        d, z = synth._group_params(K, 16, 8, "timeline|%d" % K, 0)
        lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "linear", K, 1, 8)
    else:
        d, z = synth._group_params(64, 16, 8, "timeline|%d" % K, 0)
        lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "linear", K, 1, 8)
    ab = ops.ActBinding(lay, pw, 8)
    ops.GEMM_FUSE = os.environ.get("TIMELINE_FUSE", "1") == "1"
    x = torch.randn(M, K, device=dev)
    lnp = (torch.ones(K, device=dev), torch.zeros(K, device=dev), 1e-5) if ln else None
    for _ in range(3):
        y = ops.quant_linear(x, ab, ln=lnp)
    torch.cuda.synchronize()
    # the pair replayed from a graph, as in the step (no host gaps between the two launches)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        y = ops.quant_linear(x, ab, ln=lnp)
    gr.replay(); torch.cuda.synchronize()
    clear()
    gr.replay(); torch.cuda.synchronize()
    print("== %d x %d x %d %s (Kp = %d)%s" % (M, N, K, mode, ab.Kp, ", LayerNorm folded" if ln else ""))
    q = fetch("quant")
    nw = len(q)
    if len(q):
        report("quantise-on-load", q, SC_SEG if (q[:, 5] != 0).any() else QA_SEG)
    g_ = fetch("gemm")
    if len(g_):
        report("GEMM (tile family)", g_, GEMM_SEG, gemm_extra)
    pn = fetch("panel")
    if len(pn):
        fused = (pn[:, 14] != 0).any()
        report("GEMM (panel kernel%s, DGQ_GEMM_FORCE=%s)" % (", quantise-on-load" if fused else "", os.environ.get("DGQ_GEMM_FORCE", "auto")),
               pn, FUSED_SEG if fused else PANEL_SEG, fused_extra if fused else None)
    del gr


shapes = sys.argv[1:] or ["8192,320,320,perK", "8192,320,320,perM", "2048,640,640,perM", "2048,640,640,perK", "512,1280,1280,perK", "512,1280,1280,perM"]
print("# tools/small_launch_timeline.py — stamped build (csrc/diag.h); cycles are shader cycles of the stamped run: read the SHARES")
for s in shapes:
    f = s.split(",")
    run(int(f[0]), int(f[1]), int(f[2]), f[3], len(f) > 4 and f[4] == "ln")
