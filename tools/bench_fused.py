"""Linear layers of the SD step as ONE launch (dgq_gemm_act_t: the GEMM quantises its own rows, gemm_panel.hip FUSE) against the
dgq_quant_act + dgq_gemm_wxa8 pair, per shape and fused configuration (DGQ_GEMM_FORCE=F<TM>,<NW>,1,<KW>), hipGraph replay of the
layer call.  Each replay cycles through R private copies of the INPUT (R x bytes >= 64 MB) so that the rows are not L2-resident
from the previous replay — in the step they come from the producing kernel, not from a replay of the same call.
usage: python tools/bench_fused.py ["M,N,K,mode[,ln|geglu]" ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act

dev = torch.device("cuda:0")
SHAPES = ["8192,320,320,perK,ln", "8192,320,320,perM,ln", "8192,320,320,perK", "8192,320,1280,perK", "8192,2560,320,perK,geglu",
          "2048,640,640,perK,ln", "2048,640,640,perM", "2048,5120,640,perK,geglu", "2048,640,2560,perK", "512,1280,1280,perK,ln",
          "512,1280,1280,perM", "512,10240,1280,perK,geglu", "512,1280,5120,perK", "128,1280,1280,perK"]
CONFIGS = ["auto", "F1,10,1,1", "F1,5,1,1", "F1,5,1,2", "F1,8,1,2", "F1,4,1,2", "F1,4,1,4"]
ITERS = 8


def replay_us(fns):
    for f in fns[:2]:
        f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / len(fns))
    del g
    return best


print("%-34s %6s | %9s | %s" % ("shape", "Kp", "2 launches", "fused: config us ..."))
tot2 = totf = 0.0
for sh in (sys.argv[1:] or SHAPES):
    f = sh.split(",")
    M, N, K, mode = int(f[0]), int(f[1]), int(f[2]), f[3]
    fold = f[4] if len(f) > 4 else None
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) * 0.05
    wd, wz = synth.channel_minmax(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, K, 1)
    if mode == "perK":
        d, z = synth._group_params(K, 16, 8, "fusedb|%d" % K, 0)
        lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "linear", K, 1, 8)
    else:
        d, z = synth._group_params(64, 16, 8, "fusedb|%d" % K, 0)
        lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "linear", K, 1, 8)
    ab = ops.ActBinding(lay, pw, 8)
    R = max(2, min(16, (64 << 20) // (M * K * 4)))
    xs = [torch.randn(M, K, device=dev) for _ in range(R)]
    ln = (torch.ones(K, device=dev), torch.zeros(K, device=dev), 1e-5) if fold == "ln" else None
    call = (lambda x: ops.quant_linear(x, ab, geglu=True)) if fold == "geglu" else (lambda x: ops.quant_linear(x, ab, ln=ln))
    fns = [(lambda x=x: call(x)) for x in xs] * (ITERS // 2 if R > 4 else ITERS)
    os.environ.pop("DGQ_GEMM_FORCE", None)
    ops.GEMM_FUSE = False
    two = replay_us(fns)
    ops.GEMM_FUSE = True
    res = []
    if ops.act_fuses(ab, M, K, torch.float32):
        for cfg in CONFIGS:
            if cfg == "auto":
                os.environ.pop("DGQ_GEMM_FORCE", None)
            else:
                os.environ["DGQ_GEMM_FORCE"] = cfg
            try:
                res.append((replay_us(fns), cfg))
            except RuntimeError as e:
                res.append((float("nan"), cfg + "!"))
    os.environ.pop("DGQ_GEMM_FORCE", None)
    auto = res[0][0] if res else float("nan")
    tot2 += two; totf += auto if auto == auto else two
    print("%-34s %6d | %9.1f | %s" % (sh, ab.Kp, two, "  ".join("%s %.1f" % (c, t) for t, c in res)), flush=True)
    del xs, fns, pw, ab
print("sum: two launches %.1f us, fused (auto) %.1f us" % (tot2, totf))
