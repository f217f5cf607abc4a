"""Generates the committed golden vectors by running the REAL reference (imported from
/root/reference on CPU through oracle/ref_harness.py).  Development-container only.

    python tests/golden/make_golden.py small            # F2 quantizers, F3 layers, F4 blocks  (seconds)
    python tests/golden/make_golden.py wide             # F3c wide Linear layers (K = 9216: clears of the running totals in the plan)
    python tests/golden/make_golden.py schema           # F1 ckpt schema + F6 loader side effects
    python tests/golden/make_golden.py calib | recon    # F8 DGQ activation calibration, F9 weight PTQ (mini model)
    python tests/golden/make_golden.py qstats           # F8b statistics of the calibration, quantizer by quantizer
    python tests/golden/make_golden.py scalers          # F2b scale initialisers (MSE / KL / HIST / OMSE / LOGMINMAX)
    python tests/golden/make_golden.py unet c1|c2|c3    # F5 full SD UNet, 64x64 latents (minutes each)
    python tests/golden/make_golden.py ddim [steps]     # F5 N-step DDIM final latent (tens of minutes)
    python tests/golden/make_golden.py ddim_traj [steps] [res]   # F5b every call of that run: ε output + the call's self-deviation
    python tests/golden/make_golden.py pndm | euler     # F7 vendored PNDM / EulerAncestral schedulers on a closed-form ε model
    python tests/golden/make_golden.py pndm_unet [steps] [res]   # F10 the reference QuantModel under the pipeline's PNDM loop
    DIFFUSERS_REWRITE=sdxl python tests/golden/make_golden.py unet xl   # SDXL (separate process)
    DIFFUSERS_REWRITE=sdxl python tests/golden/make_golden.py unet xl_c5 32   # SDXL W4A6 g=1 (config C5), 32x32 latents; xl_c5b8: 8 prompts

Fixtures are DATA: seeded inputs and the reference's outputs (``.pt`` files of plain tensors +
JSON).  Full-UNet inputs are not stored — they are regenerated from ``dgq_amd.synth`` (name-keyed).
"""
import json
import math
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_harness as rh          # noqa: E402
from dgq_amd import synth                      # noqa: E402
from tests.golden import recipes               # noqa: E402

torch.set_grad_enabled(False)


def g(seed):
    return torch.Generator().manual_seed(seed)


def save(name, obj):
    torch.save(obj, os.path.join(HERE, name))
    print("wrote", name)


# --------------------------------------------------------------------------------------- F2
def make_quantizers(ref):
    out = {}
    # UniformAffineQuantizer inference branch incl. exact .5 ties, clamp edges, z outside [0, 2^b-1]
    for bits in (4, 6, 8):
        for tag, delta, zp in (("pos", 0.05, 3.0), ("negz", 0.031, -7.0), ("bigz", 0.02, float(2 ** bits + 9))):
            x = torch.randn(4, 33, generator=g(bits * 10 + len(tag))) * (delta * 2 ** bits / 3)
            ties = (torch.arange(-8, 8, dtype=torch.float32) + 0.5) * delta      # exact half steps
            x = torch.cat([x.flatten(), ties, torch.tensor([0.0, -0.0, 1e-9, 1e6, -1e6])])
            q = ref.ql.UniformAffineQuantizer(bits=bits)
            q.delta, q.zero_point, q.init = torch.tensor(delta), torch.tensor(zp), True
            out["uaq_b%d_%s" % (bits, tag)] = dict(x=x, delta=q.delta, zp=q.zero_point, bits=bits, y=q(x))
    # broadcast layouts (1,1,X) and (1,X,1)
    x = torch.randn(2, 12, 20, generator=g(5))
    for tag, shp in (("lastdim", (1, 1, 20)), ("dim1", (1, 12, 1))):
        d = 0.01 + 0.05 * torch.rand(shp, generator=g(6))
        z = torch.round(128 + 40 * torch.randn(shp, generator=g(7)))
        q = ref.ql.UniformAffineQuantizer(bits=8)
        q.delta, q.zero_point, q.init = d, z, True
        out["uaq_bcast_" + tag] = dict(x=x, delta=d, zp=z, bits=8, y=q(x))
    # per-channel minmax init (quant_layer.py:253-264) incl. all-positive / all-negative / constant rows
    w = torch.randn(9, 3, 3, 3, generator=g(8)) * 0.2
    w[0] = w[0].abs() + 0.01
    w[1] = -w[1].abs() - 0.01
    w[2] = 0.0
    w[3] = -1e-10        # (+1e-10 makes the reference itself raise TypeError at quant_layer.py:37)
    for bits in (4, 8):
        q = ref.ql.UniformAffineQuantizer(bits=bits, channel_wise=True)
        y = q(w)
        out["minmax_ch_b%d" % bits] = dict(w=w, bits=bits, delta=q.delta.clone(), zp=q.zero_point.clone(), y=y)
    # scalar minmax (activation self-init) and always_zero
    x = torch.randn(3, 50, generator=g(9)) * 2 + 0.3
    q = ref.ql.UniformAffineQuantizer(bits=8, leaf_param=True)
    y = q(x)
    out["minmax_scalar"] = dict(x=x, bits=8, delta=q.delta.data.clone(), zp=torch.as_tensor(q.zero_point).clone(), y=y)
    p = torch.softmax(torch.randn(2, 4, 10, 10, generator=g(10)) * 3, -1)
    q = ref.ql.UniformAffineQuantizer(bits=8, always_zero=True)
    y = q(p)
    out["minmax_always_zero"] = dict(x=p, bits=8, delta=torch.as_tensor(q.delta).clone(), y=y)
    # AdaRound hard rounding
    w = torch.randn(6, 16, generator=g(11)) * 0.1
    uq = ref.ql.UniformAffineQuantizer(bits=4, channel_wise=True)
    uq(w)
    ar = ref.ar.AdaRoundQuantizer(uq, w, rmode=ref.ar.RMODE.LEARNED_HARD_SIGMOID)
    alpha = torch.randn(6, 16, generator=g(12))
    ar.alpha = torch.nn.Parameter(alpha)
    out["adaround_b4"] = dict(w=w, delta=uq.delta.clone(), zp=uq.zero_point.clone(), alpha=alpha, bits=4, y=ar(w))
    # T2ILogQuantizer: zeros, x>δ, real-time, a∈{6,8}
    p = torch.softmax(torch.randn(2, 3, 9, 17, generator=g(13)) * 4, -1)
    p[0, 0, 0, :3] = 0.0
    for bits in (6, 8):
        q = ref.qlt.T2ILogQuantizer(bits=bits, real_time=True)
        out["logq_rt_b%d" % bits] = dict(x=p, bits=bits, y=q(p))
        q = ref.qlt.T2ILogQuantizer(bits=bits, real_time=False)
        y = q(p)
        out["logq_static_b%d" % bits] = dict(x=p, bits=bits, delta=q.delta.clone(), y=y)
        q = ref.qlt.T2ILogQuantizer(bits=bits, real_time=False)
        q.delta, q.init = torch.tensor(0.05), True      # x > δ  → negative log2 → clamps to code 0
        out["logq_fixed_b%d" % bits] = dict(x=p, bits=bits, delta=q.delta.clone(), y=q(p))
    save("f2_quantizers.pt", out)


# --------------------------------------------------------------------------------------- F3
def make_layers_wide(ref):
    """F3c: the wide Linear cases of recipes.f3_wide_cases() through the reference's QuantLayer (as make_layers)."""
    make_layers(ref, recipes.f3_wide_cases(), "f3c_layers_wide.pt")


def make_layers(ref, cases=None, fname="f3_layers.pt"):
    """QuantLayer Linear & Conv (3x3 s1, 3x3 s2 p1, 1x1) x act-param layout x W{4,8} x A{6,8} x G{1,8,16}.
    Inputs come from tests/golden/recipes.py (name-keyed, regenerated by the tests); only the
    reference's weight-quantizer params and outputs are stored."""
    import torch.nn as nn
    out = {}
    for case in (cases or recipes.f3_cases()):
        inp = recipes.f3_inputs(case)
        wq = {"bits": case["wbits"], "channel_wise": True, "scaler": ref.ql.Scaler.MINMAX}
        aq = {"bits": case.get("abits", 8), "channel_wise": False, "scaler": ref.ql.Scaler.MINMAX,
              "leaf_param": True}
        if case["kind"] == "linear":
            layer = nn.Linear(inp["w"].shape[1], inp["w"].shape[0])
        else:
            layer = nn.Conv2d(inp["w"].shape[1], inp["w"].shape[0], case["k"], case["stride"], case["padding"])
        layer.weight.data, layer.bias.data = inp["w"].clone(), inp["b"].clone()
        ql = ref.ql.QuantLayer(layer, dict(wq), dict(aq))
        state = case["state"]
        ql.set_quant_state(state != "fp", state == "wa")
        ql(inp["x"])                                             # self-init quantizers
        rec = {}
        if state == "wa":
            ql.aqtizer.delta.data, ql.aqtizer.zero_point = inp["adelta"], inp["azp"]
            if case["kind"] == "conv" and case["layout"] != "scalar":
                ql.use_group_num = True                          # what calibration.py:268-291 does
        if state != "fp":
            rec["wdelta"], rec["wzp"] = ql.wqtizer.delta.clone(), ql.wqtizer.zero_point.clone()
        rec["y"] = ql(inp["x"]).clone()
        out[case["name"]] = rec
    save(fname, out)


# --------------------------------------------------------------------------------------- F4
def make_blocks(ref):
    """Attention_forward self/cross with start_peak on/off inside QuantBasicTransformerBlock (hidden 64;
    the reference hard-codes 8 heads & ctx 768, sd.py:243-245) and QuantResnetBlock2D (temb 1280)."""
    import torch.nn as nn
    out = {}
    for case in recipes.f4_tblock_cases():
        inp = recipes.f4_tblock_inputs(case)
        abits = case["abits"]
        wq = {"bits": 4, "channel_wise": True, "scaler": ref.ql.Scaler.MINMAX}
        aq = {"bits": abits, "channel_wise": False, "scaler": ref.ql.Scaler.MINMAX, "leaf_param": True}
        sm = {"softmax_a_bit": abits, "t2i_log_quant": case["log"], "t2i_real_time": case["rt"],
              "t2i_start_peak": case["sp"], "log_max_1": False}
        blk = ref.dr.BasicTransformerBlock(recipes.F4_HIDDEN)
        blk.load_state_dict(inp["fp_sd"])
        for name, mod in list(blk.named_modules()):              # QuantModel.quant_module by hand
            for cname, child in list(mod.named_children()):
                if isinstance(child, nn.Linear):
                    setattr(mod, cname, ref.ql.QuantLayer(child, dict(wq), dict(aq)))
        qb = ref.qb.QuantBasicTransformerBlock(blk, dict(aq), sm)
        qb.set_quant_state(True, True)
        qb(inp["x"], inp["ctx"])                                 # self-init every quantizer (scalars)
        act = {}
        for name, mod in qb.named_modules():
            if isinstance(mod, ref.ql.UniformAffineQuantizer) and "aqtizer" in name and mod.delta is not None:
                if name in inp["act_override"]:
                    d, z = inp["act_override"][name]
                    mod.delta.data = d
                    mod.zero_point = z
                act[name] = (torch.as_tensor(mod.delta.data).clone(),
                             torch.as_tensor(mod.zero_point).clone().float())
        wqp = {name: (m.wqtizer.delta.clone(), m.wqtizer.zero_point.clone())
               for name, m in qb.named_modules() if isinstance(m, ref.ql.QuantLayer)}
        out["tblock_" + case["name"]] = dict(act=act, wq=wqp, y=qb(inp["x"], inp["ctx"]).clone())
    # ResNet block 64 -> 96 (shortcut), 10x10, grouped conv inputs
    inp = recipes.f4_resnet_inputs()
    rb = ref.dr.ResnetBlock2D(64, 96, conv_shortcut=True)
    rb.load_state_dict(inp["fp_sd"])
    wq = {"bits": 4, "channel_wise": True, "scaler": ref.ql.Scaler.MINMAX}
    aq = {"bits": 8, "channel_wise": False, "scaler": ref.ql.Scaler.MINMAX, "leaf_param": True}
    for cname, child in list(rb.named_children()):
        if isinstance(child, (nn.Linear, nn.Conv2d)):
            setattr(rb, cname, ref.ql.QuantLayer(child, dict(wq), dict(aq)))
    qr = ref.qb.QuantResnetBlock2D(rb, dict(aq))
    qr.set_quant_state(True, True)
    qr(inp["x"], inp["temb"])
    act = {}
    for name in ("conv1", "conv2", "conv_shortcut", "time_emb_proj"):
        ql = getattr(qr, name)
        if name in inp["act_override"]:
            ql.aqtizer.delta.data, ql.aqtizer.zero_point = inp["act_override"][name]
            ql.use_group_num = True
        act[name] = (torch.as_tensor(ql.aqtizer.delta.data).clone(),
                     torch.as_tensor(ql.aqtizer.zero_point).clone().float())
    wqp = {n: (getattr(qr, n).wqtizer.delta.clone(), getattr(qr, n).wqtizer.zero_point.clone())
           for n in ("conv1", "conv2", "conv_shortcut", "time_emb_proj")}
    out["resnet_w4a8g8"] = dict(act=act, wq=wqp, y=qr(inp["x"], inp["temb"]).clone())
    save("f4_blocks.pt", out)


# --------------------------------------------------------------------------------------- F5
UNET_CFG = {
    # name: (wbits, abits, use_aq, G, log, rt, sp, time_aware, steps, timesteps)
    "c1": dict(wbits=8, abits=8, use_aq=False, G=1, log=False, rt=False, sp=False, time_aware=False, steps=50,
               ts=(981,)),
    "c2": dict(wbits=4, abits=8, use_aq=True, G=16, log=True, rt=True, sp=True, time_aware=True, steps=50,
               ts=(981, 481)),
    "c3": dict(wbits=4, abits=6, use_aq=True, G=8, log=True, rt=True, sp=True, time_aware=True, steps=50,
               ts=(981, 21)),
    # c2 without the time-aware reload (the reference's --fp16 mode cannot run WITH it: load_act_ckpt_with_difference_shape puts
    # fp32 deltas back at every forward, the quantizer output is promoted to fp32 and the next matmul raises "expected scalar type
    # Float but found Half", quant_layer.py:562 / F.linear)
    "c2n": dict(wbits=4, abits=8, use_aq=True, G=16, log=True, rt=True, sp=True, time_aware=False, steps=50,
                ts=(981,)),
    "c2u": dict(wbits=4, abits=8, use_aq=True, G=1, log=False, rt=False, sp=False, time_aware=True, steps=50,
                ts=(981,)),
}


def build_ref_unet_qnn(ref, arch, c, res, batch, slots, path=None):
    path = path or "/tmp/golden_%s_%s_r%d.pth" % (arch, "w%da%dg%d" % (c["wbits"], c["abits"], c["G"]), res)
    if not os.path.exists(path):
        synth.write_cali_ckpt(path, arch, c["wbits"], c["abits"], c["G"], num_slots=slots, seed=0, batch=batch,
                              res=res, start_peak=c["sp"], uniform_softmax=(c["use_aq"] and not c["log"]),
                              with_act=c["use_aq"])
    fp_sd = synth.synth_state_dict(arch, 0)
    unet = ref.dr.UNet2DConditionModel()
    unet.load_state_dict(fp_sd)
    sm = {"softmax_a_bit": c["abits"], "t2i_log_quant": c["log"], "t2i_real_time": c["rt"],
          "t2i_start_peak": c["sp"], "log_max_1": False}
    a = synth.ARCH[arch]
    init = [torch.randn(1, 4, res, res), torch.randint(0, 1000, (1,)), torch.randn(1, 77, a["ctx_dim"])]
    if arch == "sdxl":
        # get_qmodel's positional adaptor (load_qmodel_util.py:6-18)
        of = unet.forward
        unet.forward = lambda s, t, e, te=None, ti=None, **kw: of(
            s, t, e, kw.pop("added_cond_kwargs", None) or {"text_embeds": te, "time_ids": ti}, **kw)
        init += [torch.randn(1, 1280), torch.randn(1, 6)]
    qnn = rh.build_reference_qnn(ref, c["wbits"], c["abits"], c["use_aq"], sm, path, tuple(init),
                                 use_group=c["G"] > 1, time_aware=c["time_aware"] and c["use_aq"],
                                 num_inference_steps=c["steps"], unet=unet)
    return qnn, path


def make_unet(ref, arch, name, res=None, batch=2):
    c = UNET_CFG[name if name in UNET_CFG else "c2"]
    if name == "xl":
        c = dict(wbits=4, abits=8, use_aq=True, G=16, log=True, rt=True, sp=True, time_aware=True, steps=4,
                 ts=(999, 249))
        batch = 1
    elif name.startswith("xl_c5"):
        # BASELINE.json configs[4] on its own graph: SDXL W4A6 with the G = 1 preset (scripts/quantize_act.sh:19-23: scalar activation
        # scales -> the native F.conv2d path, quant_layer.py:659; uniform aqtizer_w, quant/quant_block.py:145-156).  "xl_c5b8": 8
        # prompts, the per-GPU batch of the config (the launch plans of the batch differ from batch 2's)
        c = dict(wbits=4, abits=6, use_aq=True, G=1, log=False, rt=False, sp=False, time_aware=True, steps=4, ts=(999, 249))
        batch = 8 if name.endswith("b8") else 2
    res = res or synth.ARCH[arch]["sample_size"]
    need_slots = 1 + max((1000 - t) // (1000 // c["steps"]) for t in c["ts"]) if c["time_aware"] else 1
    t0 = time.time()
    qnn, path = build_ref_unet_qnn(ref, arch, c, res, batch, need_slots)
    print("reference qnn ready in %.0fs" % (time.time() - t0))
    inp = synth.synth_inputs(arch, batch, 1, res)
    outs = {}
    for t in c["ts"]:
        t0 = time.time()
        if arch == "sdxl":
            y = qnn(inp["sample"], torch.tensor(t), inp["encoder_hidden_states"],
                    added_cond_kwargs={"text_embeds": inp["text_embeds"], "time_ids": inp["time_ids"]})[0]
        else:
            y = qnn(inp["sample"], torch.tensor(t), inp["encoder_hidden_states"])[0]
        print("t=%d  %.1fs  absmax %.4f" % (t, time.time() - t0, y.abs().max().item()))
        outs[t] = y.clone()
    meta = dict(c)
    meta.update(arch=arch, res=res, batch=batch, seed=0, input_seed=1)
    blob = dict(meta=meta, outputs=outs)
    if c["use_aq"]:
        # The same reference code on the same inputs with ONE BLAS thread: only the fp32 summation order inside
        # the GEMMs changes.  The difference to `outputs` is the reference's own sensitivity to rounding — the
        # scale against which any other implementation's end-to-end deviation has to be read.
        nt = torch.get_num_threads()
        torch.set_num_threads(1)
        outs1 = {}
        for t in c["ts"]:
            t0 = time.time()
            if arch == "sdxl":
                y = qnn(inp["sample"], torch.tensor(t), inp["encoder_hidden_states"],
                        added_cond_kwargs={"text_embeds": inp["text_embeds"], "time_ids": inp["time_ids"]})[0]
            else:
                y = qnn(inp["sample"], torch.tensor(t), inp["encoder_hidden_states"])[0]
            rel = ((y - outs[t]).norm() / outs[t].norm()).item()
            print("1-thread t=%d  %.1fs  rel-L2 vs %d-thread run: %.4g" % (t, time.time() - t0, nt, rel))
            outs1[t] = y.clone()
        torch.set_num_threads(nt)
        blob["outputs_1thread"] = outs1
        blob["meta"]["threads"] = nt
    save("f5_unet_%s_%s_r%d.pt" % (arch, name, res), blob)


def make_unet_half(ref, arch="sd", name="c2", res=16, batch=2):
    """The reference's own --fp16 mode (src/inference_qmodel.py:96-97 -> QuantModel.half(), quant_model.py:183-192) on the CPU: the
    same model / inputs as f5_unet_<arch>_<name>_r<res>.pt, cast by the reference's half() and fed fp16 tensors.  Recorded: its
    output per timestep (and the fp32 run beside it), so that the HIP path's fp16 mode is compared with what the reference's fp16
    mode does to this network instead of with a tolerance read off our own measurement."""
    c = UNET_CFG[name]
    res = res or synth.ARCH[arch]["sample_size"]
    need_slots = 1 + max((1000 - t) // (1000 // c["steps"]) for t in c["ts"]) if c["time_aware"] else 1
    qnn, path = build_ref_unet_qnn(ref, arch, c, res, batch, need_slots)
    inp = synth.synth_inputs(arch, batch, 1, res)
    out32, out16 = {}, {}
    for t in c["ts"]:
        out32[t] = qnn(inp["sample"], torch.tensor(t), inp["encoder_hidden_states"])[0].clone()
    qnn.half()
    for t in c["ts"]:
        t0 = time.time()
        y = qnn(inp["sample"].half(), torch.tensor(t), inp["encoder_hidden_states"].half())[0]
        assert y.dtype == torch.float16
        out16[t] = y.clone()
        print("fp16 t=%d  %.1fs  rel-L2 vs the fp32 run: %.4g" % (t, time.time() - t0, ((y.float() - out32[t]).norm() / out32[t].norm()).item()))
    meta = dict(c)
    meta.update(arch=arch, res=res, batch=batch, seed=0, input_seed=1, threads=torch.get_num_threads())
    save("f5c_unet_%s_%s_r%d_fp16.pt" % (arch, name, res), dict(meta=meta, outputs_fp32=out32, outputs_fp16=out16))


def make_ddim(ref, steps=50, res=64, name="c2"):
    """C2: SD W4A8 G16, N-step DDIM, CFG 7.5, final latent."""
    from oracle import dgq_oracle as orc
    c = dict(UNET_CFG[name])
    c["steps"] = steps
    qnn, path = build_ref_unet_qnn(ref, "sd", c, res, 2, steps,
                                   path="/tmp/golden_sd_ddim%d_%s_r%d.pth" % (steps, name, res))
    lat = synth.named_randn("latent", (1, 4, res, res), 1)
    ctx = synth.named_randn("ctx", (2, 77, 768), 2)

    def fn(x, t, ctx):
        t0 = time.time()
        y = qnn(x, torch.tensor(t), ctx)[0]
        print("  t=%d %.1fs" % (t, time.time() - t0), flush=True)
        return y
    out = orc.denoise_loop(fn, lat, ctx, steps, guidance=7.5)
    meta = dict(c)
    meta.update(arch="sd", res=res, guidance=7.5)
    save("f5_ddim%d_sd_%s_r%d.pt" % (steps, name, res), dict(meta=meta, final_latent=out))


def make_ddim_traj(ref, steps=50, res=64, name="c2"):
    """F5b: every UNet call of the reference's N-step DDIM run of C2, not only the final latent.  Per call:
    the ε output of the reference QuantModel on the CFG pair [2,4,res,res] and the reference's OWN deviation on that very input
    when only the BLAS thread count changes (a scalar: the noise floor of the call, the scale any other implementation's
    deviation is read against).  The inputs are not stored: latent_in of call i+1 is the DDIM step (elementwise fp32, oracle.DDIM)
    of latent_in of call i and the stored ε — the script asserts that this reconstruction is bit-identical to the trajectory it
    ran."""
    from oracle import dgq_oracle as orc
    c = dict(UNET_CFG[name])
    c["steps"] = steps
    qnn, path = build_ref_unet_qnn(ref, "sd", c, res, 2, steps,
                                   path="/tmp/golden_sd_ddim%d_%s_r%d.pth" % (steps, name, res))
    lat = synth.named_randn("latent", (1, 4, res, res), 1)
    ctx = synth.named_randn("ctx", (2, 77, 768), 2)
    sch = orc.DDIM(steps)
    nt = torch.get_num_threads()
    x = lat.clone()
    eps_all, self_dev, lat_in = [], [], []
    for i, t in enumerate(sch.timesteps):
        t0 = time.time()
        inp = torch.cat([x, x], dim=0)
        y = qnn(inp, torch.tensor(t), ctx)[0].clone()
        t1 = time.time()
        torch.set_num_threads(1)
        y1 = qnn(inp, torch.tensor(t), ctx)[0]
        torch.set_num_threads(nt)
        sd = ((y1.double() - y.double()).norm() / y.double().norm()).item()
        print("  call %d t=%d  %.1fs + %.1fs (1 thread)  self-deviation %.4g" % (i, t, t1 - t0, time.time() - t1, sd), flush=True)
        lat_in.append(x.clone())
        eps_all.append(y)
        self_dev.append(sd)
        e_u, e_c = y.chunk(2)
        x = sch.step(e_u + 7.5 * (e_c - e_u), t, x)
    # the reconstruction the test performs
    xr = lat.clone()
    for i, t in enumerate(sch.timesteps):
        assert torch.equal(xr, lat_in[i]), i
        e_u, e_c = eps_all[i].chunk(2)
        xr = sch.step(e_u + 7.5 * (e_c - e_u), t, xr)
    assert torch.equal(xr, x)
    meta = dict(c)
    meta.update(arch="sd", res=res, guidance=7.5, threads=nt)
    save("f5b_ddim%d_traj_sd_%s_r%d.pt" % (steps, name, res),
         dict(meta=meta, timesteps=list(sch.timesteps), eps=torch.stack(eps_all), self_dev=torch.tensor(self_dev, dtype=torch.float64),
              final_latent=x))


# --------------------------------------------------------------------------------------- F1/F6
def make_schema(ref, arch):
    res = 16
    c = UNET_CFG["c2"]
    qnn, path = build_ref_unet_qnn(ref, arch, dict(c, steps=2), res, 2, 2,
                                   path="/tmp/golden_schema_%s.pth" % arch)
    ck = torch.load(path)
    schema = {k: {kk: [list(v.shape), str(v.dtype)] for kk, v in d.items()} for k, d in ck.items()}
    side = {}
    for name, m in qnn.named_modules():
        if isinstance(m, ref.ql.QuantLayer):
            side[name] = dict(use_group_num=bool(m.use_group_num), use_wq=bool(m.use_wq), use_aq=bool(m.use_aq),
                              disable_aq=bool(m.disable_aq), adelta_shape=(list(m.aqtizer.delta.shape)
                                                                          if m.aqtizer.delta is not None else None))
    slots = {str(t): int((1000 - t) // (1000 // n)) for n in (50, 25, 4) for t in
             ([int(i * (1000 // n)) + 1 for i in range(n)] if n != 4 else (999, 749, 499, 249))}
    with open(os.path.join(HERE, "f1_f6_schema_%s.json" % arch), "w") as f:
        json.dump(dict(n_weight_keys=len(ck["weight"]), n_act_keys=len(ck["act_0"]), schema_res=res,
                       act0=schema["act_0"], weight={k: v for k, v in list(schema["weight"].items())},
                       loader_side_effects=side, slots=slots), f)
    print("wrote schema", arch)


# --------------------------------------------------------------------------------------- calibration producer (f1)
CALIB = dict(wbits=4, abits=8, G=4, n=32, interval=16, res=16, ts=(901, 301), np_seed=123, mode="minmax")


def calib_data():
    c = CALIB
    xs = synth.named_randn("calib_x", (c["n"], 4, c["res"], c["res"]), 3)
    ts = torch.tensor([c["ts"][i // c["interval"]] for i in range(c["n"])], dtype=torch.int64)
    ctx = synth.named_randn("calib_ctx", (c["n"], 77, 768), 4)
    return xs, ts, ctx


def build_ref_mini(ref):
    """ARCH['mini'] of dgq_amd.diffusers_rewrite composed from the REFERENCE's block classes (same state-dict keys)."""
    import torch.nn as nn
    S = sys.modules["diffusers_rewrite.sd"]

    class RefMini(nn.Module):
        def __init__(self):
            super().__init__()
            import types
            self.config = types.SimpleNamespace(in_channels=4, sample_size=16, time_cond_proj_dim=None)
            self.conv_in = nn.Conv2d(4, 64, kernel_size=3, stride=1, padding=1)
            self.time_proj = S.Timesteps(64)
            self.time_embedding = S.TimestepEmbedding(in_features=64, out_features=1280)
            self.down_blocks = nn.ModuleList([S.CrossAttnDownBlock2D(64, 64, n_layers=1, has_shortcut=False),
                                              S.DownBlock2D(64, 64, has_downsamplers=False)])
            self.up_blocks = nn.ModuleList([S.UpBlock2D(in_channels=64, out_channels=64, prev_output_channel=64),
                                            S.CrossAttnUpBlock2D(in_channels=64, out_channels=64, prev_output_channel=64,
                                                                 n_layers=1, has_upsamplers=False)])
            self.mid_block = S.UNetMidBlock2DCrossAttn(64)
            self.conv_norm_out = nn.GroupNorm(32, 64, eps=1e-05, affine=True)
            self.conv_act = nn.SiLU()
            self.conv_out = nn.Conv2d(64, 4, kernel_size=3, stride=1, padding=1)

        def forward(self, sample, timesteps, encoder_hidden_states=None, **kwargs):
            timesteps = timesteps.expand(sample.shape[0])
            emb = self.time_embedding(self.time_proj(timesteps).to(dtype=sample.dtype))
            sample = self.conv_in(sample)
            s0 = sample
            sample, [s1, s2, s3] = self.down_blocks[0](sample, temb=emb, encoder_hidden_states=encoder_hidden_states)
            sample, [s4, s5] = self.down_blocks[1](sample, temb=emb)
            sample = self.mid_block(sample, emb, encoder_hidden_states=encoder_hidden_states)
            sample = self.up_blocks[0](hidden_states=sample, temb=emb, res_hidden_states_tuple=[s3, s4, s5])
            sample = self.up_blocks[1](hidden_states=sample, temb=emb, res_hidden_states_tuple=[s0, s1, s2],
                                       encoder_hidden_states=encoder_hidden_states)
            return [self.conv_out(self.conv_act(self.conv_norm_out(sample)))]
    m = RefMini()
    m.load_state_dict(synth.synth_state_dict("mini", 0))          # strict: the key sets must coincide
    return m


def make_calib(ref):
    """F8: the reference's DGQ activation calibration (quant/calibration_group_quantization.py:44-129 cali_model_aq)
    on the mini model: per quantizer the folded (min, max) vectors that enter done_group_num and the (δ, z) it
    produces, plus the act_<t> dicts act_group_quant would save."""
    import contextlib, io
    import numpy as np
    import quant.calibration_group_quantization as cgq
    c = CALIB
    unet = build_ref_mini(ref)
    wq = {"bits": c["wbits"], "channel_wise": True, "scaler": ref.ql.Scaler.MINMAX}
    aq = {"bits": c["abits"], "channel_wise": False, "scaler": ref.ql.Scaler.MINMAX, "leaf_param": True}
    sm = {"softmax_a_bit": c["abits"], "t2i_log_quant": True, "t2i_real_time": True, "t2i_start_peak": True, "log_max_1": False}
    wpath = "/tmp/golden_mini_weight_only.pth"
    torch.save(synth.synth_weight_ckpt("mini", c["wbits"], 0), wpath)
    xs, ts, ctx = calib_data()
    with contextlib.redirect_stdout(io.StringIO()):
        qnn = ref.qm.QuantModel(model=unet, wq_params=wq, aq_params=aq, softmax_aq_params=sm,
                                aq_mode=[ref.ql.QMODE.NORMAL.value, ref.ql.QMODE.QDIFF.value], tib_recon=False).eval()
        ref.cal.load_cali_model(qnn, init_data=(xs[:1], ts[:1], ctx[:1]), use_aq=False, path=wpath)
    names = {id(m): n for n, m in qnn.model.named_modules()}
    ranges = {}
    cur = {"t": 0}
    orig_done = ref.ql.UniformAffineQuantizer.done_group_num

    def spy(self, group_num, mode):
        if self.min_max_per_in_channel != []:
            im = torch.stack([x[0] for x in self.min_max_per_in_channel]).min(dim=0)[0]
            ix = torch.stack([x[1] for x in self.min_max_per_in_channel]).max(dim=0)[0]
            om = torch.stack([x[0] for x in self.min_max_per_out_channel]).min(dim=0)[0]
            ox = torch.stack([x[1] for x in self.min_max_per_out_channel]).max(dim=0)[0]
            r = orig_done(self, group_num, mode)
            ranges.setdefault(cur["t"], {})[names[id(self)]] = dict(
                in_min=im.clone(), in_max=ix.clone(), out_min=om.clone(), out_max=ox.clone(),
                delta=self.delta.data.clone(), zero_point=torch.as_tensor(self.zero_point).data.clone())
            return r
        return orig_done(self, group_num, mode)
    ref.ql.UniformAffineQuantizer.done_group_num = spy
    orig_qdone = ref.qm.QuantModel.done_group_num

    def qdone(self, group_num, mode):
        r = orig_qdone(self, group_num, mode)
        cur["t"] += 1
        return r
    ref.qm.QuantModel.done_group_num = qdone
    np.random.seed(c["np_seed"])
    with contextlib.redirect_stdout(io.StringIO()):
        model_dict = cgq.cali_model_aq("sd", qnn, (xs, ts, ctx), {}, c["G"], c["interval"], c["mode"])
    ref.ql.UniformAffineQuantizer.done_group_num = orig_done
    ref.qm.QuantModel.done_group_num = orig_qdone
    act = {k: {kk: vv.detach().clone() for kk, vv in v.items()} for k, v in model_dict.items()}
    print("intervals:", list(act), "keys per interval:", len(act["act_0"]), "grouped quantizers:", len(ranges[0]))
    import sklearn
    save("f8_calibration_mini.pt", dict(meta=dict(c, sklearn=sklearn.__version__), act=act, ranges=ranges))


def make_qstats(ref):
    """F8b: the statistics half of DGQ's calibration, quantizer by quantizer: the REFERENCE's
    UniformAffineQuantizer in the state cali_model_aq puts it in (scalar self-initialisation by the first batch, then
    group_num = G) over three batches of 3-D (Linear / unfolded conv), 4-D (attention q / k) and 2-D inputs:
    the per-batch (min, max) vectors it records (quant_layer.py:301-313), the folded ranges, the (δ, z) of done_group_num
    (:315-429), and — second quantizer per case — the scalar EMA range of act_momentum_update (:431-446, group_num = 1)."""
    G = 8
    out = {}
    for name, shape in recipes.QSTAT_CASES:
        xs = recipes.qstat_batches(name, shape)
        q = ref.ql.UniformAffineQuantizer(bits=8, channel_wise=False, scaler=ref.ql.Scaler.MINMAX, leaf_param=True)
        q(xs[0])                                              # scalar self-init (calibration_group_quantization.py:83-85)
        d0, z0 = q.delta.data.clone(), torch.as_tensor(q.zero_point).clone()
        q.group_num = G
        per_batch = []
        for x in xs:
            q(x)
            if q.min_max_per_in_channel:
                per_batch.append(tuple(t.clone() for t in q.min_max_per_in_channel[-1] + q.min_max_per_out_channel[-1]))
        rec = dict(shape=shape, init_delta=d0, init_zp=z0, per_batch=per_batch)
        if q.min_max_per_in_channel:
            im = torch.stack([m[0] for m in q.min_max_per_in_channel]).min(dim=0)[0]
            ix = torch.stack([m[1] for m in q.min_max_per_in_channel]).max(dim=0)[0]
            om = torch.stack([m[0] for m in q.min_max_per_out_channel]).min(dim=0)[0]
            ox = torch.stack([m[1] for m in q.min_max_per_out_channel]).max(dim=0)[0]
            rec["ranges"] = (im, ix, om, ox)
        q.done_group_num(G, "minmax")
        rec["delta"], rec["zero_point"] = q.delta.data.clone(), torch.as_tensor(q.zero_point).clone()
        q2 = ref.ql.UniformAffineQuantizer(bits=8, channel_wise=False, scaler=ref.ql.Scaler.MINMAX, leaf_param=True)
        q2(xs[0])
        q2.group_num = 1                                      # "elif self.group_num != -1": the scalar EMA branch
        for x in xs[1:]:
            q2(x)
        rec["ema"] = (q2.x_min.clone(), q2.x_max.clone(), q2.delta.data.clone(), torch.as_tensor(q2.zero_point).clone())
        out[name] = rec
        print(name, shape, "delta", tuple(rec["delta"].shape), "distinct", int(torch.unique(rec["delta"]).numel()))
    import sklearn
    save("f8b_quantizer_statistics.pt", dict(meta=dict(G=G, sklearn=sklearn.__version__), cases=out))


def make_scalers(ref):
    """F2b: the reference's scale initialisers (quant_layer.py:22-185) — MSE per output channel through
    UniformAffineQuantizer(channel_wise=True) exactly as a weight quantizer initialises (:253-264), and the scalar forms of
    MSE / KL / HIST / OMSE / LOGMINMAX."""
    out = {}
    for name, sc, shape, level, cw in recipes.SCALER_CASES:
        x = recipes.scaler_input(name, shape)
        fn = getattr(ref.ql.Scaler, sc)
        if cw:
            bits = {16: 4, 256: 8}[level]
            q = ref.ql.UniformAffineQuantizer(bits=bits, channel_wise=True, scaler=fn, leaf_param=False)
            d, z = q._init_quantization_param(x, True)
        elif sc == "LOGMINMAX":
            d, z = fn(x, False, level, True), torch.tensor(0.0)
        else:
            sym, az = recipes.scaler_flags(name)
            d, z = fn(x, sym, level, az)
        out[name] = dict(delta=torch.as_tensor(d).clone().float(), zero_point=torch.as_tensor(z).clone().float())
        print(name, sc, shape, "delta", out[name]["delta"].flatten()[:4].tolist(), "zp", out[name]["zero_point"].flatten()[:4].tolist())
    save("f2b_scale_initialisers.pt", out)


# --------------------------------------------------------------------------------------- weight PTQ (f4)
RECON = dict(wbits=4, n=8, res=16, ts=(901, 301), iters=8, batch_size=4, w=0.01, warmup=0.2, seed=1234,
             full_alpha=("model.conv_in", "model.time_embedding.linear_1", "model.down_blocks.0.attentions.0.proj_in",
                         "model.down_blocks.0.resnets.0.conv1",
                         "model.down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q"))


def recon_data():
    c = RECON
    xs = synth.named_randn("recon_x", (c["n"], 4, c["res"], c["res"]), 5)
    ts = torch.tensor([c["ts"][i * len(c["ts"]) // c["n"]] for i in range(c["n"])], dtype=torch.int64)
    ctx = synth.named_randn("recon_ctx", (c["n"], 77, 768), 6)
    return xs, ts, ctx


def make_recon(ref):
    """F9: the reference's weight PTQ (quant/calibration.py:100-206 cali_model -> quant/reconstruction.py layer_ /
    block_reconstruction, AdaRound soft targets, LossFunc) on the mini model: target order, per-target loss trajectories,
    per-layer rounding decisions (packed signs of α) and α moments, full α of a few layers, the saved ckpt's schema; and
    known-answer vectors of the soft quantiser / regulariser with their autograd gradients."""
    import contextlib, io
    import numpy as np
    import quant.reconstruction as rec
    import quant.reconstruction_util as ru
    c = RECON
    torch.set_grad_enabled(True)
    # --- known-answer vectors: AdaRoundQuantizer soft forward + regulariser, values and gradients
    kat = {}
    for name, (N, K, bits) in dict(a=(6, 40, 4), b=(5, 33, 8)).items():
        w = torch.randn(N, K, generator=g(91)) * 0.3
        uq = ref.ql.UniformAffineQuantizer(bits=bits, channel_wise=True, scaler=ref.ql.Scaler.MINMAX)
        uq(w)
        q = ref.ar.AdaRoundQuantizer(uaqtizer=uq, rmode=ref.ar.RMODE.LEARNED_HARD_SIGMOID, w=w)
        q.soft_tgt = True
        with torch.no_grad():
            q.alpha.add_(torch.randn(N, K, generator=g(92)) * 3.0)        # spread α over both clamp regions of h
            q.alpha[0, :4] = torch.tensor([0.0, 30.0, -30.0, 1e-3])
        gout = torch.randn(N, K, generator=g(93))
        out = q(w)
        (out * gout).sum().backward()
        ga = q.alpha.grad.clone()
        regs = {}
        for b in (20.0, 7.3, 2.0):
            q.alpha.grad = None
            r = (1 - ((q.get_soft_tgt() - .5).abs() * 2).pow(b)).sum()
            (0.01 * r).backward()
            regs[b] = dict(value=r.detach().clone(), galpha=q.alpha.grad.clone())
        kat[name] = dict(w=w, delta=q.delta.detach().clone(), zero_point=torch.as_tensor(q.zero_point).detach().clone(), bits=bits,
                         alpha=q.alpha.detach().clone(), gout=gout, out=out.detach().clone(), galpha=ga,
                         soft_tgt=q.get_soft_tgt().detach().clone(), reg=regs)
    sched = {}
    for iters, warm in ((8, 0.2), (20000, 0.2), (100, 0.0)):
        lf = ru.LinearTempDecay(t_max=iters, rel_start_decay=warm, start_b=20, end_b=2)
        pts = sorted(set([1, 2, max(1, int(iters * warm)), int(iters * warm) + 1, iters // 2, iters - 1, iters]))
        sched[(iters, warm)] = [(t, float(lf(t))) for t in pts]
    # --- the driver on the mini model
    torch.set_grad_enabled(False)
    unet = build_ref_mini(ref)
    wq = {"bits": c["wbits"], "channel_wise": True, "scaler": ref.ql.Scaler.MINMAX, "leaf_param": False}
    aq = {"bits": 8, "channel_wise": False, "scaler": ref.ql.Scaler.MINMAX, "leaf_param": False}
    sm = {"softmax_a_bit": 8, "t2i_log_quant": True, "t2i_real_time": True, "t2i_start_peak": True, "log_max_1": False}
    with contextlib.redirect_stdout(io.StringIO()):
        qnn = ref.qm.QuantModel(model=unet, wq_params=wq, aq_params=aq, softmax_aq_params=sm,
                                aq_mode=[ref.ql.QMODE.NORMAL.value, ref.ql.QMODE.QDIFF.value], tib_recon=False).eval()
    names = {id(m): n for n, m in qnn.named_modules()}
    order, traj = [], {}
    orig_layer, orig_block, orig_call = rec.layer_reconstruction, rec.block_reconstruction, ru.LossFunc.__call__

    def spy_call(self, pred, tgt, grad=None):
        total = orig_call(self, pred, tgt, grad)
        rec_l = float(ref.ql.lp_loss(pred, tgt, p=self.p))
        traj.setdefault(names[id(self.o)], []).append((self.count, float(total), rec_l, float(total) - rec_l,
                                                        float(self.temp_decay(self.count))))
        return total
    ru.LossFunc.__call__ = spy_call

    def wrap(kind, fn):
        def f(model, target, cali_data, **kw):
            order.append((kind, names[id(target)], bool(kw.get("keep_gpu"))))
            with torch.enable_grad():
                return fn(model, target, cali_data=cali_data, **kw)
        return f
    ref.cal.layer_reconstruction = wrap("layer", orig_layer)
    ref.cal.block_reconstruction = wrap("block", orig_block)
    torch.manual_seed(c["seed"])
    path = "/tmp/golden_recon/cali_ckpt.pth"
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        ref.cal.cali_model(qnn, w_cali_data=recon_data(), a_cali_data=None, use_aq=False, path=path, running_stat=False,
                           interval=c["n"], tib_recon=False, iters=c["iters"], batch_size=c["batch_size"], w=c["w"], asym=True,
                           warmup=c["warmup"], opt_mode=ru.RLOSS.MSE, multi_gpu=False, no_recon=False, resume_w=None)
    ru.LossFunc.__call__ = orig_call
    ref.cal.layer_reconstruction, ref.cal.block_reconstruction = orig_layer, orig_block
    ck = torch.load(path + "_weight_only")["weight"]
    layers = {}
    for k, v in ck.items():
        if k.endswith(".wqtizer.alpha"):
            name = k[:-len(".wqtizer.alpha")]
            a = v.float()
            layers[name] = dict(shape=tuple(a.shape), up=torch.from_numpy(np.packbits((a >= 0).numpy().reshape(-1))),
                                n_up=int((a >= 0).sum()), sum=float(a.double().sum()), abs_sum=float(a.double().abs().sum()),
                                delta=ck[name + ".wqtizer.delta"].clone(), zero_point=ck[name + ".wqtizer.zero_point"].clone())
            if name in c["full_alpha"]:
                layers[name]["alpha"] = a.clone()
    schema = {k: (tuple(v.shape), str(v.dtype)) for k, v in ck.items()}
    print("targets: %d (%d blocks), alpha layers: %d, ckpt keys: %d, %.0f s" % (
        len(order), sum(1 for o in order if o[0] == "block"), len(layers), len(schema), time.time() - t0))
    save("f9_weight_ptq_mini.pt", dict(meta=c, kat=kat, sched=sched, order=order, traj=traj, layers=layers, schema=schema))


# --------------------------------------------------------------------------------------- scheduler (f3)
def load_vendored_pndm(which="pndm"):
    """The reference's vendored diffusers 0.26.0 does not import as a package here; its
    schedulers/scheduling_pndm.py (or scheduling_euler_ancestral_discrete.py) is loaded on its own with minimal stand-ins
    for the three modules it imports from."""
    import dataclasses, enum, importlib.util, types
    import numpy  # noqa: F401
    root = "/root/reference/diffusers/src/diffusers"
    pk = types.ModuleType("vdiff"); pk.__path__ = [root]
    cu = types.ModuleType("vdiff.configuration_utils")

    class ConfigMixin:
        pass

    def register_to_config(init):
        import functools, inspect

        @functools.wraps(init)
        def inner(self, *a, **k):
            sig = inspect.signature(init)
            ba = sig.bind(self, *a, **k); ba.apply_defaults()
            kw = {n: v for n, v in ba.arguments.items() if n != "self"}
            self.config = types.SimpleNamespace(**kw)
            init(self, *a, **k)
        return inner
    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    ut = types.ModuleType("vdiff.utils"); ut.__path__ = []
    import logging as _logging

    class BaseOutput(dict):
        pass
    ut.BaseOutput = BaseOutput
    ut.logging = types.SimpleNamespace(get_logger=_logging.getLogger)
    tu = types.ModuleType("vdiff.utils.torch_utils")

    def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
        # what diffusers' randn_tensor does for a CPU generator: draw on the CPU, then move
        return torch.randn(shape, generator=generator, dtype=dtype).to(device)
    tu.randn_tensor = randn_tensor
    sc = types.ModuleType("vdiff.schedulers"); sc.__path__ = [root + "/schedulers"]
    su = types.ModuleType("vdiff.schedulers.scheduling_utils")

    class KarrasDiffusionSchedulers(enum.Enum):
        PNDMScheduler = 1

    class SchedulerMixin:
        pass

    @dataclasses.dataclass
    class SchedulerOutput:
        prev_sample: torch.Tensor
    su.KarrasDiffusionSchedulers, su.SchedulerMixin, su.SchedulerOutput = KarrasDiffusionSchedulers, SchedulerMixin, SchedulerOutput
    sys.modules.update({"vdiff": pk, "vdiff.configuration_utils": cu, "vdiff.utils": ut, "vdiff.utils.torch_utils": tu,
                        "vdiff.schedulers": sc, "vdiff.schedulers.scheduling_utils": su})
    fname = "scheduling_pndm" if which == "pndm" else "scheduling_euler_ancestral_discrete"
    spec = importlib.util.spec_from_file_location("vdiff.schedulers." + fname, root + "/schedulers/%s.py" % fname)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod.PNDMScheduler if which == "pndm" else mod.EulerAncestralDiscreteScheduler


def fake_eps(x, t):
    """Deterministic stand-in for the UNet in the scheduler fixture."""
    return 0.3 * x * math.cos(t / 100.0) + 0.1 * torch.sin(3.0 * x + t)


def make_pndm():
    """F7: the vendored PNDM scheduler (SD-v1-4 config: scaled_linear 0.00085-0.012, skip_prk_steps, steps_offset 1)
    driven exactly like pipeline_stable_diffusion.py:1013-1044 with a closed-form ε model: timesteps + every latent."""
    P = load_vendored_pndm()
    out = {}
    for n in (25, 50, 8):
        sch = P(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                skip_prk_steps=True, set_alpha_to_one=False, steps_offset=1)
        sch.set_timesteps(n)
        x = torch.randn(2, 4, 8, 8, generator=g(77))
        xs = [x.clone()]
        for t in sch.timesteps:
            x = sch.step(fake_eps(x, int(t)), int(t), x, return_dict=False)[0]
            xs.append(x.clone())
        out[n] = dict(timesteps=[int(t) for t in sch.timesteps], samples=torch.stack(xs))
        print("pndm n=%d: %d unet calls, first timesteps %s" % (n, len(sch.timesteps), out[n]["timesteps"][:4]))
    save("f7_pndm_schedule.pt", out)


def make_euler():
    """F7b: the vendored EulerAncestralDiscreteScheduler in SDXL-turbo's configuration (scaled_linear 0.00085-0.012,
    "trailing" spacing, ε-prediction) driven like pipeline_stable_diffusion_xl.py:1170-1200 (init_noise_sigma, scale_model_input,
    step with a seeded CPU generator) with a closed-form ε model: timesteps, sigmas, every scaled model input and latent."""
    E = load_vendored_pndm("euler")
    out = {}
    for n in (4, 1, 8):
        sch = E(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                timestep_spacing="trailing", steps_offset=1)
        sch.set_timesteps(n)
        gen = g(1234 + n)
        x = torch.randn(2, 4, 8, 8, generator=g(78)) * sch.init_noise_sigma
        xs, ins = [x.clone()], []
        for t in sch.timesteps:
            xi = sch.scale_model_input(x, t)
            ins.append(xi.clone())
            x = sch.step(fake_eps(xi, float(t)), t, x, generator=gen, return_dict=False)[0]
            xs.append(x.clone())
        out[n] = dict(timesteps=[float(t) for t in sch.timesteps], sigmas=sch.sigmas.clone(), init_noise_sigma=float(sch.init_noise_sigma),
                      samples=torch.stack(xs), inputs=torch.stack(ins), noise_seed=1234 + n)
        print("euler-ancestral n=%d: timesteps %s sigmas %s" % (n, out[n]["timesteps"], [round(float(v), 4) for v in sch.sigmas]))
    save("f7_euler_ancestral_schedule.pt", out)


def make_pndm_unet(steps=8, res=64, name="c2"):
    """F10: the REFERENCE QuantModel under the pipeline's PNDM loop — the vendored PNDMScheduler,
    CFG 7.5, the time-aware slot formula with num_inference_steps = steps (N + 1 UNet calls, the 2nd and 3rd at the same
    timestep = the same slot) exactly as pipeline_stable_diffusion.py:1013-1044 drives ``pipe.unet``.  Stored: the timesteps,
    for the first three calls the UNet input and output (and the output of the same call with one BLAS thread: the
    reference's own sensitivity to summation order), the final latent of the 8-thread and of a 1-thread trajectory."""
    ref = rh.import_reference("sd")
    P = load_vendored_pndm("pndm")
    c = dict(UNET_CFG[name])
    c["steps"] = steps
    qnn, path = build_ref_unet_qnn(ref, "sd", c, res, 2, steps, path="/tmp/golden_sd_pndm%d_%s_r%d.pth" % (steps, name, res))
    lat0 = synth.named_randn("latent", (1, 4, res, res), 1)
    ctx = synth.named_randn("ctx", (2, 77, 768), 2)

    def trajectory(record):
        sch = P(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                skip_prk_steps=True, set_alpha_to_one=False, steps_offset=1)
        sch.set_timesteps(steps)
        x = lat0.clone() * sch.init_noise_sigma
        for i, t in enumerate(sch.timesteps):
            t0 = time.time()
            inp = sch.scale_model_input(torch.cat([x] * 2), t)
            y = qnn(inp, t, encoder_hidden_states=ctx, timestep_cond=None, cross_attention_kwargs=None,
                    added_cond_kwargs=None, return_dict=False)[0]
            if record is not None and i < 3:
                record.append((int(t), inp.clone(), y.clone()))
            u, v = y.chunk(2)
            x = sch.step(u + 7.5 * (v - u), t, x, return_dict=False)[0]
            print("  call %d t=%d %.1fs" % (i, int(t), time.time() - t0), flush=True)
        return x, [int(t) for t in sch.timesteps]
    rec = []
    final, ts = trajectory(rec)
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    outs1 = []
    for (t, inp, y) in rec:                                   # the same three calls, one BLAS thread
        t0 = time.time()
        y1 = qnn(inp, torch.tensor(t), encoder_hidden_states=ctx, return_dict=False)[0]
        print("  1-thread call t=%d %.1fs rel-L2 vs %d threads %.4g" % (t, time.time() - t0, nt, ((y1 - y).norm() / y.norm()).item()), flush=True)
        outs1.append(y1.clone())
    final1, _ = trajectory(None)
    torch.set_num_threads(nt)
    print("final latent: 1-thread vs %d-thread rel-L2 %.4g" % (nt, ((final1 - final).norm() / final.norm()).item()))
    meta = dict(c)
    meta.update(arch="sd", res=res, guidance=7.5, scheduler="pndm", threads=nt)
    save("f10_pndm%d_sd_%s_r%d.pt" % (steps, name, res),
         dict(meta=meta, timesteps=ts, calls=[dict(t=t, inp=inp, out=y, out_1thread=y1) for (t, inp, y), y1 in zip(rec, outs1)],
              final_latent=final, final_latent_1thread=final1))


if __name__ == "__main__":  # noqa: C901
    what = sys.argv[1]
    if what == "pndm":
        make_pndm()
        sys.exit(0)
    if what == "euler":
        make_euler()
        sys.exit(0)
    if what == "pndm_unet":
        make_pndm_unet(int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 64)
        sys.exit(0)
    arch = os.environ.get("DIFFUSERS_REWRITE", "sd")
    ref = rh.import_reference(arch)
    if what == "small":
        make_quantizers(ref)
        make_layers(ref)
        make_blocks(ref)
    elif what == "wide":
        make_layers_wide(ref)
    elif what == "schema":
        make_schema(ref, arch)
    elif what == "unet":
        res = int(sys.argv[3]) if len(sys.argv) > 3 else None
        make_unet(ref, arch, sys.argv[2], res=res)
    elif what == "unet_half":
        make_unet_half(ref, arch, sys.argv[2] if len(sys.argv) > 2 else "c2", int(sys.argv[3]) if len(sys.argv) > 3 else 16)
    elif what == "calib":
        make_calib(ref)
    elif what == "qstats":
        make_qstats(ref)
    elif what == "scalers":
        make_scalers(ref)
    elif what == "recon":
        make_recon(ref)
    elif what == "ddim_traj":
        make_ddim_traj(ref, int(sys.argv[2]) if len(sys.argv) > 2 else 50, int(sys.argv[3]) if len(sys.argv) > 3 else 64)
    elif what == "ddim":
        make_ddim(ref, int(sys.argv[2]) if len(sys.argv) > 2 else 50,
                  int(sys.argv[3]) if len(sys.argv) > 3 else 64)
