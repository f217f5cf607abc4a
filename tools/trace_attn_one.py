"""Kernel names and durations of one short attention call per form (run under rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops
dev = torch.device("cuda:0")
B, H = 2, 8
for D, T, S in ((40, 4096, 77), (160, 256, 77), (160, 256, 256)):
    q, k, v = (torch.randn(B, n, H * D, device=dev) for n in (T, S, S))
    skip = 1 if S == 77 else 0
    tab = lambda n: (torch.rand(n, device=dev) * 0.02 + 0.02, torch.randint(100, 156, (n,), device=dev).float())
    fq = ((1,) + tab(T) + (0, 8), (1,) + tab(S - skip) + (skip, 8), (2,) + tab(D) + (0, 8))
    for one in ("0", "1"):
        os.environ["DGQ_ATTN_ONE"] = one
        for _ in range(5):
            ops.attention(q, k, v, H, D, D ** -0.5, 1, skip, None, 8, fq=fq)
        torch.cuda.synchronize()
print("timeouts", ops.attention_sync_timeouts())
