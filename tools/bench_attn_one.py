"""Short-key-range attention calls of an SD step (B=2, H=8), three launches (DGQ_ATTN_ONE=0) against the single-launch form
(csrc/attn_one.hip), real-time δ (mode 1: with the in-launch exchange) and static δ (mode 2: without): hipGraph replay of 20 calls ->
us per call incl. the pre-pass.   usage: python tools/bench_attn_one.py ["D,T,S" ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops
dev = torch.device("cuda:0")
shapes = [(40, 4096, 77), (80, 1024, 77), (160, 256, 77), (160, 64, 77), (160, 256, 256), (160, 64, 64)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
B, H, iters = 2, 8, 20


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


for D, T, S in shapes:
    q, k, v = (torch.randn(B, n, H * D, device=dev) for n in (T, S, S))
    skip = 1 if S == 77 else 0
    tab = lambda n: (torch.rand(n, device=dev) * 0.02 + 0.02, torch.randint(100, 156, (n,), device=dev).float())
    dl = torch.tensor([0.8], device=dev)
    for name, fq in (("int8", ((1,) + tab(T) + (0, 8), (1,) + tab(S - skip) + (skip, 8), (2,) + tab(D) + (0, 8))),
                     ("Q1K3", ((2,) + tab(D) + (0, 8), (1,) + tab(S - skip) + (skip, 8), (2,) + tab(D) + (0, 8)))):
        res = []
        for mode, delta in ((1, None), (2, dl)):
            for one in ("0", "1"):
                os.environ["DGQ_ATTN_ONE"] = one
                res.append(timed(lambda: ops.attention(q, k, v, H, D, D ** -0.5, mode, skip, delta, 8, fq=fq)))
        os.environ.pop("DGQ_ATTN_ONE")
        print("D=%3d T=%5d S=%4d %s  real-time δ: pre-pass + statistics + P·V launches %6.1f us, pre-pass + one launch %6.1f | static δ: %6.1f, %6.1f"
              % (D, T, S, name, res[0], res[1], res[2], res[3]), flush=True)
print("sync timeouts:", ops.attention_sync_timeouts())
