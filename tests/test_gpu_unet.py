"""GPU parity of the whole hot path — QuantModel.forward built through the drop-in API
(get_qmodel -> load_cali_model on a synthetic reference-format cali_ckpt) — against
  (a) the CPU oracle at a small latent size (computed on the spot), and
  (b) the REAL reference's outputs at full 64x64 latents (tests/golden/f5_*.pt).
Tolerance: 1e-3 relative (L2 and max-abs/absmax) on the UNet output — BASELINE.json's bound for the
final latent; weight codes and activation codes are bit-exact by the kernel tests."""
import os
import types

import pytest
import torch

from dgq_amd import synth

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def build_qnn(arch, c, res, batch, slots, tmpdir):
    from dgq_amd.diffusers_rewrite import UNet2DConditionModel
    from dgq_amd.quant import get_qmodel, Scaler
    path = os.path.join(tmpdir, "ck_%s_w%da%dg%d_r%d.pth" % (arch, c["wbits"], c["abits"], c["G"], res))
    if not os.path.exists(path):
        synth.write_cali_ckpt(path, arch, c["wbits"], c["abits"], c["G"], num_slots=slots, seed=0, batch=batch, res=res,
                              start_peak=c["sp"], uniform_softmax=(c["use_aq"] and not c["log"]), with_act=c["use_aq"])
    unet = UNet2DConditionModel(arch)
    synth.load_synth_weights(unet, arch, 0)
    pipe = types.SimpleNamespace(unet=unet)
    wq = {"bits": c["wbits"], "channel_wise": True, "scaler": Scaler.MINMAX}
    aq = {"bits": c["abits"], "channel_wise": False, "scaler": Scaler.MINMAX, "leaf_param": c["use_aq"]}
    sm = {"softmax_a_bit": c["abits"], "t2i_log_quant": c["log"], "t2i_real_time": c["rt"],
          "t2i_start_peak": c["sp"], "log_max_1": False}
    qnn = get_qmodel(arch, pipe, path, wq, c["use_aq"], aq, sm, c["G"] > 1, c["steps"],
                     c["time_aware"] and c["use_aq"])
    qnn.float()
    qnn = qnn.cuda()
    qnn.disable_out_quantization()
    return qnn, path


C2 = dict(wbits=4, abits=8, use_aq=True, G=16, log=True, rt=True, sp=True, time_aware=True, steps=50)
C1 = dict(wbits=8, abits=8, use_aq=False, G=1, log=False, rt=False, sp=False, time_aware=False, steps=50)
C3 = dict(wbits=4, abits=6, use_aq=True, G=8, log=True, rt=True, sp=True, time_aware=True, steps=50)


def test_small_unet_vs_oracle(tmp_path_factory):
    """SD W4A8 g16 + log/real-time/start-peak + time-aware at 16x16 latents: HIP path vs CPU oracle."""
    from oracle import dgq_oracle as orc
    tmp = str(tmp_path_factory.mktemp("ck"))
    c = dict(C2, steps=2)
    qnn, path = build_qnn("sd", c, 16, 2, 2, tmp)
    inp = synth.synth_inputs("sd", 2, 1, 16)
    ck = torch.load(path)
    cfg = orc.OracleConfig("sd", 4, 8, True, True, 8, True, True, True, True, 2, True)
    om = orc.OracleModel(ck, cfg, synth.synth_state_dict("sd", 0))
    for t in (999, 499):
        ref = om.forward(inp["sample"], t, inp["encoder_hidden_states"])
        with torch.no_grad():
            y = qnn(inp["sample"].cuda(), torch.tensor(t), inp["encoder_hidden_states"].cuda())[0]
        y = y.float().cpu()
        e = rel_l2(y, ref)
        m = ((y - ref).abs().max() / ref.abs().max()).item()
        print("t=%d rel_l2=%.3g max/absmax=%.3g" % (t, e, m))
        assert e < 1e-3 and m < 5e-3, (t, e, m)


@pytest.mark.parametrize("name,c", [("c2", C2), ("c1", C1), ("c3", C3)])
def test_full_unet_vs_reference_golden(name, c, tmp_path_factory):
    f = os.path.join(GOLD, "f5_unet_sd_%s_r64.pt" % name)
    if not os.path.exists(f):
        pytest.skip("golden %s not generated" % f)
    g = torch.load(f)
    tmp = str(tmp_path_factory.mktemp("ck"))
    ts = sorted(g["outputs"].keys(), reverse=True)
    slots = 1 + max((1000 - t) // 20 for t in ts) if c["time_aware"] else 1
    qnn, _ = build_qnn("sd", c, 64, 2, slots, tmp)
    inp = synth.synth_inputs("sd", 2, 1, 64)
    for t in ts:
        with torch.no_grad():
            y = qnn(inp["sample"].cuda(), torch.tensor(t), inp["encoder_hidden_states"].cuda())[0]
        y = y.float().cpu()
        ref = g["outputs"][t]
        e = rel_l2(y, ref)
        m = ((y - ref).abs().max() / ref.abs().max()).item()
        print("%s t=%d rel_l2=%.3g max/absmax=%.3g" % (name, t, e, m))
        assert e < 1e-3, (name, t, e, m)
