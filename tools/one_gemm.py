"""Runs ONE dgq_gemm_wxa8 shape a few times (for rocprofv3 --pmc passes).  usage: one_gemm.py M N K mode [iters] [fp32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act
dev = torch.device("cuda:0")
M, N, K, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
w = torch.randn(N, K) * 0.05
wd, wz = synth.channel_minmax(w, 4)
pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, K, 1)
if mode == "perK":
    d, z = synth._group_params(K, 16, 8, "one", 0)
    lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "linear", K, 1, 8)
else:
    d, z = synth._group_params(64, 16, 8, "one", 0)
    lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "linear", K, 1, 8)
ab = ops.ActBinding(lay, pw, 8)
codes = torch.randint(-128, 128, (M, ab.Kp), dtype=torch.int8, device=dev)
rowsum = torch.randn(M, device=dev)
odt = torch.bfloat16 if (len(sys.argv) > 6 and sys.argv[6] == "bf16") else torch.float32
out = torch.empty(M, N, device=dev, dtype=odt)
for _ in range(iters):
    ops.gemm_wxa8(codes, rowsum, M, ab, odt, out)
torch.cuda.synchronize()
print("done", M, N, K, ab.Kp)
