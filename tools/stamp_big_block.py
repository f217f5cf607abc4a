"""Block timeline of the 256-row GEMM from a -DBIG_STAMP=2 build (diagnostic, WRONG output): s_memtime at kernel entry, K-loop
start, K-loop end, epilogue stores issued, stores retired — per wave of workgroups 0 and 8 (the same XCD).
usage: DGQ_HIP_LIB=.../libdgq_stamp2.so python tools/stamp_big_block.py [perM|perK] [M N K] [geglu]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "perM"
M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (8192, 10240, 1280)
os.environ["DGQ_GEMM_FORCE"] = "256,256,1"
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
out = torch.empty(M, N, device=dev, dtype=torch.float32)
for _ in range(3):
    ops.gemm_wxa8(codes, rowsum, M, ab, torch.float32, out)
torch.cuda.synchronize()
st = out[0].view(torch.int64).cpu()[:256].view(2, 8, 16)[:, :, :5]
print(mode, M, N, K, "Kp", ab.Kp, " cycles from the workgroup's earliest entry stamp: entry, loop start, loop end, stores issued, stores retired")
for b in range(2):
    base = int(st[b, :, 0].min())
    for w_ in range(8):
        r = [int(v) - base for v in st[b, w_]]
        print("wg %d wave %d  %s   [prologue %d, loop %d, epilogue issue %d, drain %d]" % (8 * b, w_, " ".join("%7d" % v for v in r), r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3]))
