"""Reads the s_memtime stamps a BIG_STAMP build of gemm_wxa8_big.hip leaves in row 0 of y (diagnostic build, GPU box):
per wave of workgroup 0, K tile 8: cycles between LOAD start / loads issued / waits done / barrier 1 passed / COMPUTE done / barrier 2
passed, for each phase.   usage: DGQ_HIP_LIB=.../libdgq_stamp.so python tools/stamp_big.py [perM|perK] [M N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops, synth
from dgq_amd.plan import plan_act
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "perM"
M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (8192, 8192, 8192)
NPH = int(os.environ.get("NPH", "2"))
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
st = out[0].view(torch.int64).cpu()[:8 * 16].view(8, 16)[:, :6 * NPH]
base = int(st.min())
names = ["load0", "issued", "waited", "bar1", "computed", "bar2"]
print(mode, M, N, K, "Kp", ab.Kp, "  (cycles relative to the earliest stamp; segment lengths in brackets)")
for w_ in range(8):
    row = [int(v) - base for v in st[w_]]
    segs = []
    for ph in range(NPH):
        r = row[6 * ph: 6 * ph + 6]
        segs.append("ph%d: " % ph + " ".join("%s %d" % (n, v) for n, v in zip(names, r)) +
                    "  [issue %d, wait %d, bar1 %d, compute %d, bar2 %d]" % (r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4]))
    print("wave %d  " % w_ + "\n        ".join(segs))
