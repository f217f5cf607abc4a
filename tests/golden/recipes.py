"""Seeded INPUT recipes shared by tests/golden/make_golden.py (which feeds them to the real
reference) and by the parity tests (which feed them to the oracle and to the HIP path).
Pure functions of names/seeds through dgq_amd.synth — no reference import, travels to the GPU box."""
from collections import OrderedDict

import torch

from dgq_amd import synth

SEED = 7
F4_HIDDEN = 64


def nt(key, shape, scale=1.0):
    return synth.named_randn("golden|" + key, shape, SEED) * scale


# ------------------------------------------------------------------------------------------ F3
def f3_cases():
    cases = []
    for wb in (4, 8):
        for ab in (6, 8):
            for G in (1, 8, 16):
                tag = "w%da%dg%d" % (wb, ab, G)
                layouts = ("scalar",) if G == 1 else ("perK", "perM")
                for layout in layouts:
                    cases.append(dict(name="linear_%s_%s" % (tag, layout), kind="linear", wbits=wb, abits=ab, G=G,
                                      layout=layout, state="wa"))
                    for ctag, k, s, p in (("c3s1", 3, 1, 1), ("c3s2", 3, 2, 1), ("c1", 1, 1, 0)):
                        cases.append(dict(name="conv_%s_%s_%s" % (ctag, tag, layout), kind="conv", wbits=wb, abits=ab,
                                          G=G, layout=layout, state="wa", k=k, stride=s, padding=p))
    cases.append(dict(name="linear_w8_only", kind="linear", wbits=8, state="w", layout="none"))
    cases.append(dict(name="linear_fp", kind="linear", wbits=8, state="fp", layout="none"))
    cases.append(dict(name="conv_c3s1_w8_only", kind="conv", wbits=8, state="w", layout="none", k=3, stride=1, padding=1))
    # 2-D input (time-embedding style): scalar activation scale
    cases.append(dict(name="linear2d_w4a8", kind="linear", wbits=4, abits=8, G=1, layout="scalar", state="wa", two_d=True))
    return cases


def f3_wide_cases():
    """F3c: one wide Linear layer per activation-scale layout (K = 9216 > the 2176 codes (8704 until round 5) a W4A8 running total may span, so the per-K
    plan carries clears of the running totals; M = 300 rows, N = 256): the shapes the 256-row GEMM and the K-split launches meet with
    REAL plan_act tables (non-representable flush coefficients), not the power-of-two scales of the exact-integer tests."""
    return [dict(name="wide_linear_w4a8g16_%s" % lay, kind="linear", wbits=4, abits=8, G=(1 if lay == "scalar" else 16), layout=lay,
                 state="wa", dims=dict(B=2, T=150, K=9216, N=256)) for lay in ("perK", "perM", "scalar")]


F3_LIN = dict(B=2, T=24, K=96, N=40)
F3_CONV = dict(B=2, C=32, H=9, W=9, N=24)


def f3_inputs(case):
    out = {}
    ab = case.get("abits", 8)
    G = case.get("G", 1)
    if case["kind"] == "linear":
        d = case.get("dims", F3_LIN)
        out["x"] = nt("f3|lin|x", (d["B"], d["T"], d["K"]), 1.5)
        if case.get("two_d"):
            out["x"] = nt("f3|lin|x2d", (d["B"], d["K"]), 1.5)
        out["w"] = nt("f3|lin|w", (d["N"], d["K"]), 0.1 if "dims" not in case else d["K"] ** -0.5)
        out["b"] = nt("f3|lin|b", (d["N"],), 0.1)
        nK, nM = d["K"], d["T"]
        shpK, shpM = (1, 1, -1), (1, -1, 1)
    else:
        d = F3_CONV
        k, s, p = case["k"], case["stride"], case["padding"]
        out["x"] = nt("f3|conv|x", (d["B"], d["C"], d["H"], d["W"]), 1.5)
        out["w"] = nt("f3|conv|w%d" % k, (d["N"], d["C"], k, k), 0.08)
        out["b"] = nt("f3|conv|b", (d["N"],), 0.1)
        ho = (d["H"] + 2 * p - k) // s + 1
        wo = (d["W"] + 2 * p - k) // s + 1
        nK, nM = d["C"] * k * k, ho * wo
        shpK, shpM = (1, -1, 1), (1, 1, -1)       # the quantizer sees the unfolded [B, C*kh*kw, L]
    if case["state"] == "wa":
        key = "f3|act|%s" % case["name"]
        if case["layout"] == "perK":
            dl, zp = synth._group_params(nK, G, ab, key, SEED)
            out["adelta"], out["azp"] = dl.view(shpK), zp.view(shpK)
        elif case["layout"] == "perM":
            dl, zp = synth._group_params(nM, G, ab, key, SEED)
            out["adelta"], out["azp"] = dl.view(shpM), zp.view(shpM)
        else:
            out["adelta"] = torch.tensor(0.03)
            out["azp"] = torch.tensor(float(2 ** (ab - 1) + (5 if case["kind"] == "conv" else -3)))
    return out


# ------------------------------------------------------------------------------------------ F4
def f4_tblock_cases():
    return [dict(name="a8g8_log_rt_sp", abits=8, G=8, log=True, rt=True, sp=True),
            dict(name="a6g4_log_rt", abits=6, G=4, log=True, rt=True, sp=False),
            dict(name="a8g1_uniform", abits=8, G=1, log=False, rt=False, sp=False)]


F4_T, F4_B = 40, 2


def _tblock_param_shapes(hidden=F4_HIDDEN, ctx=768):
    shp = OrderedDict()
    for n in ("norm1", "norm2", "norm3"):
        shp[n + ".weight"] = (hidden,)
        shp[n + ".bias"] = (hidden,)
    for a, cd in (("attn1", hidden), ("attn2", ctx)):
        shp[a + ".to_q.weight"] = (hidden, hidden)
        shp[a + ".to_k.weight"] = (hidden, cd)
        shp[a + ".to_v.weight"] = (hidden, cd)
        shp[a + ".to_out.0.weight"] = (hidden, hidden)
        shp[a + ".to_out.0.bias"] = (hidden,)
    shp["ff.net.0.proj.weight"] = (hidden * 8, hidden)
    shp["ff.net.0.proj.bias"] = (hidden * 8,)
    shp["ff.net.2.weight"] = (hidden, hidden * 4)
    shp["ff.net.2.bias"] = (hidden,)
    return shp


def _named_params(prefix, shapes):
    sd = OrderedDict()
    for k, shp in shapes.items():
        if len(shp) >= 2:
            fan = 1
            for d in shp[1:]:
                fan *= d
            sd[k] = nt(prefix + k, shp, fan ** -0.5)
        elif k.endswith("weight"):
            sd[k] = 1.0 + nt(prefix + k, shp, 0.1)
        else:
            sd[k] = nt(prefix + k, shp, 0.05)
    return sd


def f4_tblock_inputs(case):
    hidden, T, B, G, ab = F4_HIDDEN, F4_T, F4_B, case["G"], case["abits"]
    out = dict(x=nt("f4|x", (B, T, hidden), 1.2), ctx=nt("f4|ctx", (B, 77, 768)),
               fp_sd=_named_params("f4|tb|", _tblock_param_shapes()))
    ov = {}
    if G > 1:
        kT = 76 if case["sp"] else 77

        def gp(name, n, shape):
            d, z = synth._group_params(n, G, ab, "f4|%s|%s" % (case["name"], name), SEED)
            ov[name] = (d.view(shape), z.view(shape))
        gp("attn1.aqtizer_q", T, (1, -1, 1))
        gp("attn1.aqtizer_k", hidden // 8, (1, 1, -1))
        gp("attn1.aqtizer_v", T, (1, -1, 1))
        gp("attn2.aqtizer_q", hidden // 8, (1, 1, -1))
        gp("attn2.aqtizer_k", kT, (1, -1, 1))
        gp("attn2.aqtizer_v", 77, (1, -1, 1))
        gp("attn1.to_q.aqtizer", hidden, (1, 1, -1))
        gp("attn1.to_k.aqtizer", T, (1, -1, 1))
        gp("attn1.to_out.0.aqtizer", T, (1, -1, 1))
        gp("attn2.to_k.aqtizer", 768, (1, 1, -1))
        gp("attn2.to_v.aqtizer", 77, (1, -1, 1))
        gp("ff.net.0.proj.aqtizer", hidden, (1, 1, -1))
        gp("ff.net.2.aqtizer", hidden * 4, (1, 1, -1))
    out["act_override"] = ov
    return out


def f4_resnet_inputs():
    shp = OrderedDict([("norm1.weight", (64,)), ("norm1.bias", (64,)), ("conv1.weight", (96, 64, 3, 3)),
                       ("conv1.bias", (96,)), ("time_emb_proj.weight", (96, 1280)), ("time_emb_proj.bias", (96,)),
                       ("norm2.weight", (96,)), ("norm2.bias", (96,)), ("conv2.weight", (96, 96, 3, 3)),
                       ("conv2.bias", (96,)), ("conv_shortcut.weight", (96, 64, 1, 1)), ("conv_shortcut.bias", (96,))])
    out = dict(x=nt("f4r|x", (2, 64, 10, 10)), temb=nt("f4r|temb", (2, 1280)), fp_sd=_named_params("f4r|", shp))
    ov = {}
    for name, n, shape in (("conv1", 64 * 9, (1, -1, 1)), ("conv2", 100, (1, 1, -1)), ("conv_shortcut", 64, (1, -1, 1))):
        d, z = synth._group_params(n, 8, 8, "f4r|act|" + name, SEED)
        ov[name] = (d.view(shape), z.view(shape))
    out["act_override"] = ov
    return out


# --------------------------------------------------------------------------------------- F8b (quantizer statistics)
QSTAT_CASES = (("linear3d", (4, 37, 96)), ("unfolded_conv", (4, 576, 49)), ("attn_q", (4, 8, 33, 40)), ("attn_k", (4, 8, 76, 40)),
               ("wide_tokens", (2, 256, 64)), ("linear2d", (4, 128)))


def qstat_batches(name, shape, n=3):
    """name-keyed inputs of the statistics fixture: per-channel and per-token structure, a different scale per batch"""
    out = []
    for i in range(n):
        x = _synth().named_randn("qstat|%s|%d" % (name, i), shape, 11)
        if len(shape) > 2:
            x = x * (0.5 + _synth().named_rand("qstat|%s|c" % name, (shape[-1],), 11)) + _synth().named_randn("qstat|%s|t" % name, shape[-2:-1] + (1,), 11)
        out.append(x * (1.0 + 0.3 * i))
    return out




def _synth():
    from dgq_amd import synth
    return synth


# ------------------------------------------------------------------------------------------- scale initialisers (F2b)
SCALER_CASES = [
    # name, scaler, shape, level, channel_wise
    ("mse_linear_w4", "MSE", (12, 96), 16, True),
    ("mse_conv_w4", "MSE", (8, 6, 3, 3), 16, True),
    ("mse_linear_w8", "MSE", (6, 64), 256, True),
    ("mse_scalar_a8", "MSE", (4, 50, 24), 256, False),
    ("kl_scalar_a8", "KL", (4, 50, 24), 256, False),
    ("hist_scalar_a8", "HIST", (4, 50, 24), 256, False),
    ("omse_scalar_l16", "OMSE", (3, 40), 16, False),
    ("logminmax_probs", "LOGMINMAX", (2, 4, 16, 16), 256, False),
    # the flags of the scalar MSE search (quant_layer.py:62-86): always_zero is what the uniform softmax quantizer aqtizer_w is
    # built with (quant_block.py:145-156) — `quantize_weight --cali --use_aq` initialises it through Scaler.MSE
    ("mse_always_zero_probs", "MSE", (2, 4, 16, 16), 256, False),
    ("mse_always_zero_a6", "MSE", (4, 50, 24), 64, False),
    ("mse_symmetric_a8", "MSE", (4, 50, 24), 256, False),
]


def scaler_flags(name):
    """(symmetric, always_zero) of a SCALER_CASES entry (by its name)"""
    return ("symmetric" in name, "always_zero" in name)


def scaler_input(name, shape):
    """Heavy-tailed data (a few outliers per channel) so that the range searches actually shrink the range."""
    x = _synth().named_randn("scaler|" + name, shape, 11)
    x = x * (1.0 + 4.0 * (_synth().named_randn("scaler_tail|" + name, shape, 12).abs() > 2.2).float())
    if name.startswith("logminmax") or name.endswith("_probs"):
        x = torch.softmax(x.reshape(shape[0], shape[1], shape[2], shape[3]) * 2.0, dim=-1)
    return x
