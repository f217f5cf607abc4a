"""Runs ONE dgq_quant_act shape a few times (for rocprofv3 passes). usage: one_quant_act.py M_side C k mode"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act
dev = torch.device("cuda:0")
side, C, k, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
taps = k * k
N = 64
w = torch.randn(N, C * taps) * 0.05
wd, wz = synth.channel_minmax(w, 4)
pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, None, 4, C, taps)
if mode == "perK":
    d, z = synth._group_params(C * taps, 16, 8, "one", 0)
    lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "conv", C, taps, 8)
else:
    d, z = synth._group_params(side * side, 16, 8, "one", 0)
    lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "conv", C, taps, 8)
ab = ops.ActBinding(lay, pw, 8)
x = torch.randn(2, side, side, C, device=dev)
for _ in range(5):
    codes, rs, M = ops.quant_act(x, 2, side, side, C, k, k, 1, k // 2, ab)
torch.cuda.synchronize()
print("done M", M, "Kp", ab.Kp, "codes MB", codes.numel() / 1e6)
