"""CPU: the oracle restatement against the golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import os

import pytest
import torch

from oracle import dgq_oracle as orc
from tests.golden import recipes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return torch.load(os.path.join(GOLD, name), map_location="cpu")


def same(a, b):
    assert a.shape == b.shape
    assert torch.equal(a, b), "max abs diff %g" % (a - b).abs().max().item()


# ------------------------------------------------------------------------------------------ F2
def test_f2_uniform_affine():
    f2 = load("f2_quantizers.pt")
    n = 0
    for k, v in f2.items():
        if k.startswith("uaq_"):
            same(orc.uaq(v["x"], v["delta"], v["zp"], v["bits"]), v["y"])
            n += 1
    assert n == 11


def test_f2_minmax_init():
    f2 = load("f2_quantizers.pt")
    for bits in (4, 8):
        v = f2["minmax_ch_b%d" % bits]
        d, z = orc.minmax_channel(v["w"], bits)
        same(d, v["delta"])
        same(z, v["zp"])
        same(orc.uaq(v["w"], d, z, bits), v["y"])
    v = f2["minmax_scalar"]
    d, z = orc.minmax_scalar(v["x"], 8)
    same(d, v["delta"])
    same(z, v["zp"].float())
    same(orc.uaq(v["x"], d, z, 8), v["y"])
    v = f2["minmax_always_zero"]
    d, z = orc.minmax_scalar(v["x"], 8, always_zero=True)
    same(d, v["delta"].float())
    same(orc.uaq(v["x"], d, z, 8), v["y"])


def test_f2_vectorised_minmax_matches():
    """dgq_amd.synth.channel_minmax (vectorised) == the reference's per-channel python loop."""
    from dgq_amd.synth import channel_minmax
    f2 = load("f2_quantizers.pt")
    for bits in (4, 8):
        v = f2["minmax_ch_b%d" % bits]
        d, z = channel_minmax(v["w"], bits)
        same(d, v["delta"])
        same(z, v["zp"])


def test_f2_adaround():
    v = load("f2_quantizers.pt")["adaround_b4"]
    same(orc.adaround_hard(v["w"], v["delta"], v["zp"], v["alpha"], 4), v["y"])


def test_f2_log_quant():
    f2 = load("f2_quantizers.pt")
    for bits in (6, 8):
        v = f2["logq_rt_b%d" % bits]
        same(orc.log_quant(v["x"], v["x"].max(), bits), v["y"])
        v = f2["logq_static_b%d" % bits]
        d = orc.log_quant_init_delta(v["x"], bits)
        same(d, v["delta"])
        same(orc.log_quant(v["x"], d, bits), v["y"])
        v = f2["logq_fixed_b%d" % bits]
        same(orc.log_quant(v["x"], v["delta"], bits), v["y"])


# ------------------------------------------------------------------------------------------ F3
def oracle_layer(case, inp, gold):
    cfg = orc.OracleConfig(wbits=case["wbits"], abits=case.get("abits", 8), use_wq=case["state"] != "fp",
                           use_aq=case["state"] == "wa", use_group=True)
    w = {"model.l.w": inp["w"], "model.l.b": inp["b"]}
    if case["state"] != "fp":
        w["model.l.wqtizer.delta"], w["model.l.wqtizer.zero_point"] = gold["wdelta"], gold["wzp"]
    ck = {"weight": w}
    if case["state"] == "wa":
        ck["act_0"] = {"model.l.aqtizer.delta": inp["adelta"], "model.l.aqtizer.zero_point": inp["azp"]}
    om = orc.OracleModel(ck, cfg)
    if case["kind"] == "linear":
        return om.linear("l", inp["x"])
    return om.conv("l", inp["x"], case["stride"], case["padding"])


@pytest.mark.parametrize("case", recipes.f3_cases(), ids=lambda c: c["name"])
def test_f3_layers(case):
    gold = load("f3_layers.pt")[case["name"]]
    inp = recipes.f3_inputs(case)
    y = oracle_layer(case, inp, gold)
    same(y, gold["y"])
    if case["state"] != "fp":      # weight-quantizer self-init restated
        d, z = orc.minmax_channel(inp["w"], case["wbits"])
        same(d, gold["wdelta"])
        same(z, gold["wzp"])


# ------------------------------------------------------------------------------------------ F4
def block_ckpt(fp_sd, gold, prefix="blk"):
    w = {}
    for k, v in fp_sd.items():
        path, leaf = k.rsplit(".", 1)
        if path in gold["wq"]:
            w["model.%s.%s.%s" % (prefix, path, "w" if leaf == "weight" else "b")] = v
        else:
            w["model.%s.%s" % (prefix, k)] = v
    for path, (d, z) in gold["wq"].items():
        w["model.%s.%s.wqtizer.delta" % (prefix, path)] = d
        w["model.%s.%s.wqtizer.zero_point" % (prefix, path)] = z
    act = {}
    for name, (d, z) in gold["act"].items():
        act["model.%s.%s.delta" % (prefix, name)] = d
        act["model.%s.%s.zero_point" % (prefix, name)] = z
    return {"weight": w, "act_0": act}


@pytest.mark.parametrize("case", recipes.f4_tblock_cases(), ids=lambda c: c["name"])
def test_f4_transformer_block(case):
    gold = load("f4_blocks.pt")["tblock_" + case["name"]]
    inp = recipes.f4_tblock_inputs(case)
    cfg = orc.OracleConfig(wbits=4, abits=case["abits"], t2i_log_quant=case["log"], t2i_real_time=case["rt"],
                           t2i_start_peak=case["sp"], use_group=True)
    om = orc.OracleModel(block_ckpt(inp["fp_sd"], gold), cfg)
    y = om.transformer_block("blk", inp["x"], inp["ctx"], heads=8)
    same(y, gold["y"])


def test_f4_resnet_block():
    gold = load("f4_blocks.pt")["resnet_w4a8g8"]
    inp = recipes.f4_resnet_inputs()
    cfg = orc.OracleConfig(wbits=4, abits=8, use_group=True)
    ck = block_ckpt(inp["fp_sd"], gold)
    for name in list(gold["act"]):      # conv/linear aqtizers are keyed '<layer>.aqtizer' in a real ckpt
        ck["act_0"]["model.blk.%s.aqtizer.delta" % name] = ck["act_0"].pop("model.blk.%s.delta" % name)
        ck["act_0"]["model.blk.%s.aqtizer.zero_point" % name] = ck["act_0"].pop("model.blk.%s.zero_point" % name)
    om = orc.OracleModel(ck, cfg)
    same(om.resnet("blk", inp["x"], inp["temb"]), gold["y"])


# ------------------------------------------------------------------------------------------ misc
def test_slot_formula():
    assert [orc.slot_for_timestep(t, 50) for t in (981, 961, 21, 1)] == [0, 1, 48, 49]
    assert [orc.slot_for_timestep(t, 4) for t in (999, 749, 499, 249)] == [0, 1, 2, 3]
    assert orc.DDIM(50).timesteps[:3] == [981, 961, 941] and orc.DDIM(50).timesteps[-1] == 1


# ---------------------------------------------------------------------------------------------- F5 (whole UNet)
def test_f5_oracle_forward_matches_reference_unet_16x16(golden_dir):
    """OracleModel.forward (the whole SD1.4 UNet: 280 quantized layers, W4A8 g16, log2 softmax with real-time δ,
    start-peak, time-aware tables) against the REAL reference's output at 16x16 latents
    (tests/golden/f5_unet_sd_c2_r16.pt, written by make_golden.py `unet c2 16`).  Same thread count as the golden run
    => the same BLAS calls => expected bit-identical; on a host whose BLAS blocks differently the fake-quant graph is
    chaotic (the golden file's own 1-thread vs 8-thread outputs differ by ~1e-1), so the fallback bound is that
    self-deviation."""
    import warnings
    from oracle import dgq_oracle as orc
    from dgq_amd import synth
    g = torch.load(os.path.join(golden_dir, "f5_unet_sd_c2_r16.pt"))
    m = g["meta"]
    assert m["arch"] == "sd" and m["res"] == 16
    ts = sorted(g["outputs"].keys(), reverse=True)
    slots = sorted({(1000 - t) // (1000 // m["steps"]) for t in ts})
    recs = synth.enumerate_act_quantizers("sd", m["batch"], m["res"])
    ck = {"act_%d" % s: synth.synth_act_slot("sd", m["abits"], m["G"], s, 0, m["batch"], m["res"], m["sp"], False, recs)
          for s in slots}
    ck["weight"] = synth.synth_weight_ckpt("sd", m["wbits"], 0)
    cfg = orc.OracleConfig("sd", m["wbits"], m["abits"], True, True, m["abits"], m["log"], m["rt"], m["sp"],
                           m["time_aware"], m["steps"], m["G"] > 1)
    fp_sd = {k[len("model."):].replace(".w", ".weight").replace(".b", ".bias"): v for k, v in ck["weight"].items()
             if k.startswith("model.conv_in.") or k.startswith("model.conv_out.")}
    fp_sd = {k: v for k, v in fp_sd.items() if k.endswith("weight") or k.endswith("bias")}
    om = orc.OracleModel(ck, cfg, fp_sd)
    inp = synth.synth_inputs("sd", m["batch"], m["input_seed"], m["res"])
    nt = torch.get_num_threads()
    torch.set_num_threads(m.get("threads", nt))
    try:
        for t in ts:
            y = om.forward(inp["sample"], t, inp["encoder_hidden_states"])
            ref = g["outputs"][t]
            e = ((y.double() - ref.double()).norm() / ref.double().norm()).item()
            self_dev = ((g["outputs_1thread"][t].double() - ref.double()).norm() / ref.double().norm()).item()
            if e != 0.0:
                warnings.warn("oracle vs reference golden at t=%d: rel-L2 %.3g (not bit-identical on this host; "
                              "reference self-deviation %.3g)" % (t, e, self_dev))
            assert e == 0.0 or e < 2.5 * self_dev, (t, e, self_dev)
    finally:
        torch.set_num_threads(nt)


def test_f5_oracle_forward_matches_reference_unet_sdxl_32x32(golden_dir):
    """The SDXL wiring of the oracle (Linear proj_in / proj_out, add_embedding of text_embeds ‖ Timesteps(256)(time_ids), 2- and
    10-layer transformers, no attention at 320 channels; 792 quantized layers, config C4: W4A8 g16, time-aware 4-step slots)
    pinned DIRECTLY: OracleModel.forward against the REAL reference's output at 32x32 latents
    (tests/golden/f5_unet_sdxl_xl_r32.pt = make_golden.py `unet xl 32` under DIFFUSERS_REWRITE=sdxl).  Same thread count as the
    golden run => the same BLAS calls => expected bit-identical; otherwise bounded by the reference's own 1-thread deviation.
    Every teacher-forced SDXL test on the GPU (792 layers against THIS oracle graph) rests on this."""
    import warnings
    from oracle import dgq_oracle as orc
    from dgq_amd import synth
    g = torch.load(os.path.join(golden_dir, "f5_unet_sdxl_xl_r32.pt"))
    m = g["meta"]
    assert m["arch"] == "sdxl" and m["res"] == 32 and m["batch"] == 1
    t = max(g["outputs"].keys())                                          # one timestep keeps the CPU suite short (slot 0)
    slot = (1000 - t) // (1000 // m["steps"])
    ck = synth.build_cali_ckpt("sdxl", m["wbits"], m["abits"], m["G"], num_slots=[slot], seed=0, batch=m["batch"], res=m["res"],
                               start_peak=m["sp"], uniform_softmax=False, with_act=True)
    cfg = orc.OracleConfig("sdxl", m["wbits"], m["abits"], True, True, m["abits"], m["log"], m["rt"], m["sp"],
                           m["time_aware"], m["steps"], m["G"] > 1)
    om = orc.OracleModel(ck, cfg, synth.synth_state_dict("sdxl", 0))
    inp = synth.synth_inputs("sdxl", m["batch"], m["input_seed"], m["res"])
    nt = torch.get_num_threads()
    torch.set_num_threads(m.get("threads", nt))
    try:
        y = om.forward(inp["sample"], t, inp["encoder_hidden_states"], text_embeds=inp["text_embeds"], time_ids=inp["time_ids"])
    finally:
        torch.set_num_threads(nt)
    ref = g["outputs"][t]
    e = ((y.double() - ref.double()).norm() / ref.double().norm()).item()
    self_dev = ((g["outputs_1thread"][t].double() - ref.double()).norm() / ref.double().norm()).item()
    if e != 0.0:
        warnings.warn("SDXL oracle vs reference golden at t=%d: rel-L2 %.3g (not bit-identical on this host; reference "
                      "self-deviation %.3g)" % (t, e, self_dev))
    assert e == 0.0 or e < 2.5 * self_dev, (t, e, self_dev)


def test_f5_oracle_forward_matches_reference_unet_sdxl_c5_32x32(golden_dir):
    """BASELINE.json configs[4] on ITS OWN graph (VERDICT r4 missing #3): SDXL W4A6 under the G = 1 preset (scripts/quantize_act.sh:19-23 —
    scalar activation scales, so every convolution takes the reference's native F.conv2d branch, quant_layer.py:659; uniform
    always-zero aqtizer_w, quant/quant_block.py:145-156), batch 2 at 32x32 latents: OracleModel.forward against the REAL
    reference's output (tests/golden/f5_unet_sdxl_xl_c5_r32.pt = make_golden.py `unet xl_c5 32` under DIFFUSERS_REWRITE=sdxl).
    Same thread count as the golden run => expected bit-identical; otherwise bounded by the reference's own 1-thread deviation."""
    import warnings
    from oracle import dgq_oracle as orc
    from dgq_amd import synth
    g = torch.load(os.path.join(golden_dir, "f5_unet_sdxl_xl_c5_r32.pt"))
    m = g["meta"]
    assert m["arch"] == "sdxl" and m["res"] == 32 and m["batch"] == 2 and m["G"] == 1 and m["abits"] == 6 and not m["log"]
    t = max(g["outputs"].keys())
    slot = (1000 - t) // (1000 // m["steps"])
    ck = synth.build_cali_ckpt("sdxl", m["wbits"], m["abits"], m["G"], num_slots=[slot], seed=0, batch=m["batch"], res=m["res"],
                               start_peak=m["sp"], uniform_softmax=True, with_act=True)
    cfg = orc.OracleConfig("sdxl", m["wbits"], m["abits"], True, True, m["abits"], m["log"], m["rt"], m["sp"],
                           m["time_aware"], m["steps"], m["G"] > 1)
    om = orc.OracleModel(ck, cfg, synth.synth_state_dict("sdxl", 0))
    inp = synth.synth_inputs("sdxl", m["batch"], m["input_seed"], m["res"])
    nt = torch.get_num_threads()
    torch.set_num_threads(m.get("threads", nt))
    try:
        y = om.forward(inp["sample"], t, inp["encoder_hidden_states"], text_embeds=inp["text_embeds"], time_ids=inp["time_ids"])
    finally:
        torch.set_num_threads(nt)
    ref = g["outputs"][t]
    e = ((y.double() - ref.double()).norm() / ref.double().norm()).item()
    self_dev = ((g["outputs_1thread"][t].double() - ref.double()).norm() / ref.double().norm()).item()
    if e != 0.0:
        warnings.warn("SDXL C5 oracle vs reference golden at t=%d: rel-L2 %.3g (not bit-identical on this host; reference "
                      "self-deviation %.3g)" % (t, e, self_dev))
    assert e == 0.0 or e < 2.5 * self_dev, (t, e, self_dev)


def test_oracle_exact_gemm_mode_is_a_rounding_level_change_per_layer():
    """The float64-GEMM variant of the oracle (the "exact" target of the GPU parity statistics) differs from the
    reference-faithful fp32 one only by the rounding of each contraction: on a single quantized layer the outputs
    agree to ~1e-6; through the whole chaotic UNet they diverge like any two fp32 runs do."""
    from oracle import dgq_oracle as orc
    case = [c for c in recipes.f3_cases() if c["name"] == "linear_w4a8g16_perK"][0]
    inp = recipes.f3_inputs(case)
    wd, wz = orc.minmax_channel(inp["w"], 4)
    ck = {"weight": {"model.l.w": inp["w"], "model.l.b": inp["b"], "model.l.wqtizer.delta": wd,
                     "model.l.wqtizer.zero_point": wz},
          "act_0": {"model.l.aqtizer.delta": inp["adelta"], "model.l.aqtizer.zero_point": inp["azp"]}}
    y32 = orc.OracleModel(ck, orc.OracleConfig(wbits=4, abits=8)).linear("l", inp["x"])
    y64 = orc.OracleModel(ck, orc.OracleConfig(wbits=4, abits=8, exact_gemm=True)).linear("l", inp["x"])
    e = ((y32.double() - y64.double()).norm() / y64.double().norm()).item()
    assert 0.0 <= e < 5e-6, e


# ------------------------------------------------------------------------------------------- scale initialisers (F2b)
@pytest.mark.parametrize("case", [c for c in recipes.SCALER_CASES if c[1] == "MSE"], ids=lambda c: c[0])
def test_f2b_scale_initialisers_vs_reference(case, golden_dir):
    """dgq_amd.quant.quant_layer's MSE range search (the reference's default weight initialiser, quant_layer.py:62-86; ``--fast``
    selects MINMAX) against the reference's own results on the same tensors: bit-identical δ and zero point — per channel through
    the vectorised ``channel_mse`` (what a weight quantizer's first forward runs, src/quantize_weight.py:166-169) and as a scalar."""
    from dgq_amd.quant import quant_layer as ql
    name, sc, shape, level, cw = case
    g = torch.load(os.path.join(golden_dir, "f2b_scale_initialisers.pt"))[name]
    x = recipes.scaler_input(name, shape)
    if cw:
        q = ql.UniformAffineQuantizer(bits={16: 4, 256: 8}[level], channel_wise=True, scaler=ql.Scaler.MSE, leaf_param=False)
        d, z = q._init_quantization_param(x, True)
        # and the scalar form, channel by channel (the reference's loop over channels)
        for c in range(shape[0]):
            dc, zc = ql.mse(x[c], False, level, False)
            assert float(dc) == float(g["delta"].flatten()[c]) and float(zc) == float(g["zero_point"].flatten()[c]), (name, c)
    else:
        sym, az = recipes.scaler_flags(name)
        d, z = ql.mse(x, sym, level, az)
    d, z = torch.as_tensor(d).float(), torch.as_tensor(z).float()
    assert d.shape == g["delta"].shape and torch.equal(d, g["delta"]), (name, (d - g["delta"]).abs().max())
    assert torch.equal(z.reshape(g["zero_point"].shape), g["zero_point"]), name


def test_scalers_outside_the_dgq_recipes_raise():
    """KL / HIST / OMSE / LOGMINMAX keep their enum names (drop-in surface) but no DGQ recipe selects them: they raise."""
    from dgq_amd.quant import quant_layer as ql
    for n in ("KL", "HIST", "OMSE", "LOGMINMAX"):
        with pytest.raises(NotImplementedError):
            getattr(ql.Scaler, n)(torch.randn(4, 4), False, 256, False)
