"""Debug aid (GPU box): per-layer comparison of the HIP path against the CPU oracle on a small UNet."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dgq_amd import synth
from oracle import dgq_oracle as orc
from tests.test_gpu_unet import build_qnn, C2

res = int(sys.argv[1]) if len(sys.argv) > 1 else 16
c = dict(C2, steps=2)
qnn, path = build_qnn("sd", c, res, 2, 2, "/tmp")
inp = synth.synth_inputs("sd", 2, 1, res)
ck = torch.load(path)
cfg = orc.OracleConfig("sd", 4, 8, True, True, 8, True, True, True, True, 2, True)
om = orc.OracleModel(ck, cfg, synth.synth_state_dict("sd", 0))
rec_o = {}
for fn in ("linear", "conv", "resnet", "transformer_block", "attention", "fp_conv"):
    orig = getattr(om, fn)
    def wrap(path, *a, _orig=orig, _fn=fn, **k):
        y = _orig(path, *a, **k)
        rec_o.setdefault(path, y.detach().clone())
        return y
    setattr(om, fn, wrap)
rec_p = {}
def hook(name):
    def f(mod, inp_, out):
        if torch.is_tensor(out):
            rec_p.setdefault(name, out.detach().float().cpu())
    return f
for name, m in qnn.model.named_modules():
    m.register_forward_hook(hook(name))
t = 999
ref = om.forward(inp["sample"], t, inp["encoder_hidden_states"])
with torch.no_grad():
    y = qnn(inp["sample"].cuda(), torch.tensor(t), inp["encoder_hidden_states"].cuda())[0].float().cpu()
print("final rel", ((y - ref).norm() / ref.norm()).item())
for k, v in rec_o.items():
    if k in rec_p:
        p = rec_p[k]
        if p.shape != v.shape:
            print("%-70s shape %s vs %s" % (k, tuple(p.shape), tuple(v.shape)))
            continue
        print("%-70s rel %.3e" % (k, ((p - v).norm() / v.norm().clamp_min(1e-30)).item()))
    else:
        print("%-70s (no product record)" % k)
