"""Where the synthetic-model build time goes (GPU box): python tests/dev/time_build.py sdxl 32"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["DGQ_BUILD_TIMING"] = "1"
import torch
from dgq_amd.runtime import build_synthetic_qnn
from dgq_amd import synth
arch = sys.argv[1] if len(sys.argv) > 1 else "sdxl"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 32
C = dict(wbits=4, abits=8, use_aq=True, G=16, log=True, rt=True, sp=True, time_aware=True, steps=4 if arch == "sdxl" else 50)
t0 = time.time()
qnn, path = build_synthetic_qnn(arch, C, res, 1 if arch == "sdxl" else 2, 1, ckpt_dir="/tmp")
print("total build %.1f s" % (time.time() - t0), flush=True)
t0 = time.time(); qnn.prepare_slots([0]); torch.cuda.synchronize(); print("prepare_slots %.1f s" % (time.time() - t0), flush=True)
inp = synth.synth_inputs(arch, 1 if arch == "sdxl" else 2, 1, res)
kw = {}
if arch == "sdxl":
    kw = dict(added_cond_kwargs={"text_embeds": inp["text_embeds"].cuda(), "time_ids": inp["time_ids"].cuda()})
t0 = time.time()
with torch.no_grad():
    y = qnn(inp["sample"].cuda(), torch.tensor(999), inp["encoder_hidden_states"].cuda(), **kw)[0]
torch.cuda.synchronize(); print("first forward %.1f s" % (time.time() - t0), flush=True)
from oracle import dgq_oracle as orc
ck = torch.load(path)
print("torch.load %.1f s" % (time.time() - t0), flush=True)
cfg = orc.OracleConfig(arch, 4, 8, True, True, 8, True, True, True, True, C["steps"], True)
t0 = time.time(); fp = synth.synth_state_dict(arch, 0); print("synth_state_dict %.1f s" % (time.time() - t0), flush=True)
om = orc.OracleModel(ck, cfg, fp)
okw = dict(text_embeds=inp["text_embeds"], time_ids=inp["time_ids"]) if arch == "sdxl" else {}
t0 = time.time(); ref = om.forward(inp["sample"], 999, inp["encoder_hidden_states"], **okw); print("oracle forward %.1f s (threads %d)" % (time.time() - t0, torch.get_num_threads()), flush=True)
print("rel", ((y.float().cpu() - ref).norm() / ref.norm()).item())
