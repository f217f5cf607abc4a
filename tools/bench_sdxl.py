"""SDXL-turbo shaped run (BASELINE.json configs[3]): W4A8 g16, 1024x1024 (128x128 latents), batch 1 (no CFG), 4 steps.
Synthetic name-keyed weights and checkpoint; prints ms per UNet step (hipGraph replay)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import synth
from dgq_amd.runtime import build_synthetic_qnn
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
CFG = dict(wbits=4, abits=8, use_aq=True, G=16, log=True, rt=True, sp=True, time_aware=True, steps=4)
t0 = time.time()
qnn, _ = build_synthetic_qnn("sdxl", CFG, 128, 1, 4, device=dev)
print("build %.1f s" % (time.time() - t0), flush=True)
qnn.prepare_slots([0, 1, 2, 3])
qnn.enable_graphs(True)
inp = synth.synth_inputs("sdxl", 1, 1, 128)
x = inp["sample"].to(dev); ctx = inp["encoder_hidden_states"].to(dev)
extra = {"added_cond_kwargs": {"text_embeds": inp["text_embeds"].to(dev), "time_ids": inp["time_ids"].to(dev)}}
ts = [999, 749, 499, 249]
with torch.no_grad():
    for t in ts:
        qnn(x, t, ctx, **extra)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        for t in ts:
            y = qnn(x, t, ctx, **extra)[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 12
print("SDXL 128x128 latents, batch 1: %.2f ms per UNet step (%.1f steps/s); finite=%s" % (dt * 1e3, 1 / dt, bool(torch.isfinite(y).all())))
