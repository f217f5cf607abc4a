"""Checker-side aid (GPU box): the 256-row GEMM kernel against exact integer expectations, with a mismatch map per 32x32 tile.
usage: python tests/dev/debug_big_gemm.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dgq_amd import _lib, ops

dev = torch.device("cuda:0")
os.environ["DGQ_GEMM_FORCE"] = os.environ.get("FORCE", "256,256,1")


def case(M, N, Kp, per_m, seed=1, groups=True):
    g = torch.Generator().manual_seed(seed)
    s = torch.randint(-16, 16, (M, Kp), generator=g, dtype=torch.int32)
    q = torch.randint(0, 16, (N, Kp), generator=g, dtype=torch.int32)
    nch = Kp // 32
    cd = torch.tensor([2.0 ** ((i % 5) - 2) for i in range(nch)]) if groups else torch.ones(nch)
    gend = torch.tensor([(i % 3 == 1 or i == nch - 1) for i in range(nch)])
    fl = gend.to(torch.uint8)
    gscale = cd.clone()
    for i in range(nch - 2, -1, -1):
        if not gend[i]:
            gscale[i] = gscale[i + 1]
    ws = ops.workspace(dev)
    y = torch.empty(M, N, dtype=torch.float32, device=dev)
    t = lambda x: x.to(dev).contiguous()
    codes = s.to(torch.int8).to(dev)
    wp = ops.pack_weight(q.to(torch.uint8).to(dev), None, Kp, 4)
    ones, zeros = torch.ones(N), torch.zeros(N)
    rs = torch.zeros(M)
    lib = _lib.load()
    if not per_m:
        acc = torch.zeros(M, N, dtype=torch.float64)
        for c in range(nch):
            acc += gscale[c].double() * (s[:, 32 * c:32 * c + 32].double() @ q[:, 32 * c:32 * c + 32].double().T)
        keep = [t(gscale), t(fl), t(ones), t(zeros), t(zeros), t(rs)]
        rc = lib.dgq_gemm_wxa8(_lib.ptr(codes), _lib.ptr(keep[5]), 1, M, Kp, _lib.ptr(wp), 4, N, 0, _lib.ptr(keep[0]), _lib.ptr(keep[1]),
                               None, None, 1, ctypes.c_float(128.0), _lib.ptr(keep[2]), _lib.ptr(keep[3]), _lib.ptr(keep[4]), None,
                               _lib.ptr(y), 0, N, _lib.ptr(ws), ws.numel(), None, _lib.stream())
    else:
        acc = s.double() @ q.double().T
        keep = [t(torch.ones(1)), t(torch.full((1,), 128.0)), t(ones), t(zeros), t(zeros), t(zeros), t(rs)]
        rc = lib.dgq_gemm_wxa8(_lib.ptr(codes), _lib.ptr(keep[6]), 1, M, Kp, _lib.ptr(wp), 4, N, 1, None, None,
                               _lib.ptr(keep[0]), _lib.ptr(keep[1]), 1, ctypes.c_float(128.0), _lib.ptr(keep[2]), _lib.ptr(keep[3]),
                               _lib.ptr(keep[4]), _lib.ptr(keep[5]), _lib.ptr(y), 0, N, _lib.ptr(ws), ws.numel(), None, _lib.stream())
    _lib.check(rc, "gemm")
    torch.cuda.synchronize()
    got = y.cpu().double()
    bad = got != acc
    print("M=%d N=%d Kp=%d %s groups=%s: %d / %d wrong, max |err| %.0f" % (M, N, Kp, "perM" if per_m else "perK", groups, int(bad.sum()), bad.numel(),
                                                                  float((got - acc).abs().max())), flush=True)
    if bad.any():
        tm, tn = (M + 31) // 32, (N + 31) // 32
        for i in range(tm):
            print("   ", "".join("X" if bad[32 * i:32 * i + 32, 32 * j:32 * j + 32].any() else "." for j in range(tn)))
        i, j = [int(v[0]) for v in torch.nonzero(bad, as_tuple=True)]
        print("    first wrong (%d, %d): got %.1f expect %.1f" % (i, j, got[i, j], acc[i, j]))


for per_m in (True, False):
    for (M, N, Kp) in ((256, 256, 128), (256, 256, 256), (256, 256, 384), (256, 256, 640), (512, 512, 1024), (203, 332, 640)):
        case(M, N, Kp, per_m, groups=False)
        if not per_m:
            case(M, N, Kp, per_m, groups=True)
