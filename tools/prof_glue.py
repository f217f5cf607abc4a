"""Attribute the non-DGQ ('glue') GPU kernels of one eager SD1.4 step to Python source lines (torch.profiler stacks)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from dgq_amd import synth
from dgq_amd.runtime import build_synthetic_qnn
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
qnn, _ = build_synthetic_qnn("sd", bench.CONFIGS["c2"]["cfg"], 64, 2, 1, device=dev)
qnn.prepare_slots([0])
lat = synth.named_randn("latent", (2, 4, 64, 64), 1).to(dev)
ctx = synth.named_randn("ctx", (2, 77, 768), 100).to(dev)
with torch.no_grad():
    qnn(lat, 981, ctx); qnn(lat, 981, ctx)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        qnn(lat, 981, ctx)
        torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::") or ev.cpu_children:
        pass
    if not ev.name.startswith("aten::"):
        continue
    if any(c.name.startswith("aten::") for c in ev.cpu_children):
        continue                                   # leaf aten ops only
    dt = sum(k.duration for k in ev.kernels)
    if dt <= 0:
        continue
    where = "?"
    for fr in ev.stack:
        if "dgq_amd" in fr and "ops.py" not in fr:
            where = fr.split("dgq_amd/")[-1]
            break
    agg[(ev.name, where)][0] += 1
    agg[(ev.name, where)][1] += dt
tot = sum(v[1] for v in agg.values())
print("aten leaf ops with GPU time: %.1f us total" % tot)
for (name, where), (n, dt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%8.1f us  n=%3d  %-28s %s" % (dt, n, name, where))
