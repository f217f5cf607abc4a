"""Where the aten glue ops of one eager SD1.4 step are issued from (TorchDispatchMode + Python stacks)."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from dgq_amd import synth
from dgq_amd.runtime import build_synthetic_qnn
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
qnn, _ = build_synthetic_qnn("sd", bench.CONFIGS["c2"]["cfg"], 64, 2, 1, device=dev)
qnn.prepare_slots([0])
lat = synth.named_randn("latent", (2, 4, 64, 64), 1).to(dev)
ctx = synth.named_randn("ctx", (2, 77, 768), 100).to(dev)
agg = collections.defaultdict(lambda: [0, 0])
WATCH = ("copy_", "clone", "_to_copy", "add.Tensor", "mul.Tensor", "gelu", "cat", "native_group_norm", "native_layer_norm", "empty", "zeros")

class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        if any(name.startswith(w) for w in WATCH):
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "dgq_amd" in fr.filename and "_lib.py" not in fr.filename:
                    where = "%s:%d" % (fr.filename.split("dgq_amd/")[-1], fr.lineno)
                    break
            n = out.numel() if isinstance(out, torch.Tensor) else 0
            agg[(name, where)][0] += 1
            agg[(name, where)][1] += n
        return out

with torch.no_grad():
    qnn(lat, 981, ctx)
    with Mode():
        qnn(lat, 981, ctx)
for (name, where), (n, numel) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("n=%3d  %10.2f Melem  %-26s %s" % (n, numel / 1e6, name, where))
