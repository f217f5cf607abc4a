"""Synthetic SD1.4 / SDXL weights and ``cali_ckpt`` files in the reference's on-disk format.

No pretrained weights, COCO captions or real checkpoints exist in this environment
(SURVEY.md §8(c)), so both the parity oracle and the benchmark use *name-keyed
deterministic tensors*: every tensor is a pure function of ``(name, shape, seed)``
through its own CPU ``torch.Generator`` — independent of construction order and of
the machine — so the development container (where the reference is imported to
make golden vectors) and the GPU box build bit-identical models without shipping
3.4 GB.

Checkpoint layout written here = what ``results/merge.py:13-18`` produces and
``quant/calibration.py:208-327`` (load_cali_model) reads:

  {'weight': {'model.<path>.w' | '.b' | '.wqtizer.delta' | '.wqtizer.zero_point' [| '.wqtizer.alpha'],
              'model.<norm>.weight' | '.bias'},
   'act_0': {'model.<path>.aqtizer.delta' | '.zero_point',
             'model.<blk>.attn{1,2}.aqtizer_{q,k,v}.delta' | '.zero_point', ...},
   'act_1': ..., }
"""
import hashlib
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from .diffusers_rewrite import UNet2DConditionModel, Attention, ARCH


# ----------------------------------------------------------------------------- name-keyed RNG
def _gen(name: str, seed: int) -> torch.Generator:
    h = hashlib.sha256(("%d|%s" % (seed, name)).encode()).digest()
    g = torch.Generator(device="cpu")
    g.manual_seed(int.from_bytes(h[:8], "little") & 0x7FFFFFFFFFFFFFFF)
    return g


def named_randn(name, shape, seed=0):
    return torch.randn(tuple(shape), generator=_gen(name, seed), dtype=torch.float32)


def named_rand(name, shape, seed=0):
    return torch.rand(tuple(shape), generator=_gen(name, seed), dtype=torch.float32)


def named_randint(name, low, high, shape, seed=0):
    return torch.randint(low, high, tuple(shape), generator=_gen(name, seed))


# ----------------------------------------------------------------------------- model weights
#: Test suites that build several models of one architecture set this to keep the generated state dicts (3.4 GB for SD,
#: 10 GB for SDXL: the CPU normal generator is the slowest part of a build); the tensors are shared, never mutated.
CACHE_STATE_DICTS = False
_SD_CACHE = {}


def synth_state_dict(arch="sd", seed=0):
    """FP32 state-dict for ``UNet2DConditionModel(arch)`` (HF key names). Weights N(0, 1/fan_in),
    so activations stay O(1) through the 860 M / 2.6 B-parameter graph."""
    if CACHE_STATE_DICTS and (arch, seed) in _SD_CACHE:
        return _SD_CACHE[(arch, seed)]
    out = _synth_state_dict(arch, seed)
    if CACHE_STATE_DICTS:
        _SD_CACHE[(arch, seed)] = out
    return out


def _synth_state_dict(arch, seed):
    with torch.device("meta"):
        skel = UNet2DConditionModel(arch)
    out = OrderedDict()
    for k, p in skel.state_dict().items():
        shp = tuple(p.shape)
        if len(shp) >= 2:                                  # Linear [N,K] / Conv [N,C,kh,kw]
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = named_randn(k, shp, seed) * (1.0 / math.sqrt(fan_in))
        elif k.endswith("weight"):                         # norm gains
            t = 1.0 + 0.1 * named_randn(k, shp, seed)
        else:                                              # biases
            t = 0.02 * named_randn(k, shp, seed)
        out[k] = t
    return out


def load_synth_weights(unet: nn.Module, arch="sd", seed=0):
    sd = synth_state_dict(arch, seed)
    unet.load_state_dict(sd)
    return unet


def state_dict_from_ckpt(path):
    """The FP state-dict (HF key names) read back from a cali_ckpt's 'weight' dict (``model.<path>.w`` / ``.b`` of wrapped
    layers), memory-mapped: the ranks of a multi-GPU launch load the file rank 0 wrote instead of
    regenerating 3.4 GB (SD) / 10 GB (SDXL) of normal variates each."""
    full = torch.load(path, map_location="cpu", mmap=True)
    out = OrderedDict()
    for k, v in full["weight"].items():
        if "wqtizer" in k:
            continue
        k = k[len("model."):] if k.startswith("model.") else k
        if k.endswith(".w"):
            k = k[:-2] + ".weight"
        elif k.endswith(".b"):
            k = k[:-2] + ".bias"
        out[k] = v
    return out


def synth_inputs(arch="sd", batch=2, seed=1, res=None):
    """(sample, t-independent) inputs of SURVEY.md §8(d): latents/ctx ~ N(0,1)."""
    a = ARCH[arch]
    res = res or a["sample_size"]
    d = dict(sample=named_randn("sample", (batch, 4, res, res), seed),
             encoder_hidden_states=named_randn("ctx", (batch, 77, a["ctx_dim"]), seed + 1))
    if arch == "sdxl":
        d["text_embeds"] = named_randn("text_embeds", (batch, 1280), seed + 2)
        d["time_ids"] = torch.tensor([[float(res * 8), float(res * 8), 0.0, 0.0, float(res * 8), float(res * 8)]]
                                     ).repeat(batch, 1)
    return d


# ----------------------------------------------------------------------------- quantizer inventory
def enumerate_act_quantizers(arch="sd", batch=2, res=None):
    """Walks the FP graph on the meta device and records, for every activation quantizer the
    reference attaches, the tensor shape it sees.

    Returns a list of dicts: name (module path without 'model.'), kind in
    {'linear3d','linear2d','conv','attn'}, and dims (K, T | L | D ...)."""
    a = ARCH[arch]
    res = res or a["sample_size"]
    with torch.device("meta"):
        net = UNet2DConditionModel(arch)
    recs, hooks = [], []

    def lin_hook(name):
        def f(mod, inp):
            x = inp[0]
            if x.dim() == 3:
                recs.append(dict(name=name + ".aqtizer", kind="linear3d", K=x.shape[-1], T=x.shape[1]))
            else:
                recs.append(dict(name=name + ".aqtizer", kind="linear2d", K=x.shape[-1]))
        return f

    def conv_hook(name):
        def f(mod, inp):
            x = inp[0]
            kh, kw = mod.kernel_size
            s, p = mod.stride[0], mod.padding[0]
            ho = (x.shape[2] + 2 * p - kh) // s + 1
            wo = (x.shape[3] + 2 * p - kw) // s + 1
            recs.append(dict(name=name + ".aqtizer", kind="conv", K=x.shape[1] * kh * kw, L=ho * wo,
                             C=x.shape[1], kh=kh, kw=kw))
        return f

    def attn_hook(name):
        def f(mod, args, kwargs):
            x = args[0]
            ctx = kwargs.get("encoder_hidden_states", args[1] if len(args) > 1 else None)
            tq = x.shape[1]
            tk = ctx.shape[1] if ctx is not None else tq
            for which, t in (("q", tq), ("k", tk), ("v", tk)):
                recs.append(dict(name="%s.aqtizer_%s" % (name, which), kind="attn", T=t, D=mod.head_dim,
                                 H=mod.num_heads, cross=ctx is not None, which=which))
        return f

    for name, mod in net.named_modules():
        if name in ("conv_in", "conv_out"):
            continue                                        # never quantised (quant_model.py:118-124)
        if isinstance(mod, nn.Linear):
            hooks.append(mod.register_forward_pre_hook(lin_hook(name)))
        elif isinstance(mod, nn.Conv2d):
            hooks.append(mod.register_forward_pre_hook(conv_hook(name)))
        elif isinstance(mod, Attention):
            hooks.append(mod.register_forward_pre_hook(attn_hook(name), with_kwargs=True))
    x = torch.empty(batch, 4, res, res, device="meta")
    t = torch.empty((), dtype=torch.int64, device="meta")
    ctx = torch.empty(batch, 77, a["ctx_dim"], device="meta")
    kw = {}
    if arch == "sdxl":
        kw["added_cond_kwargs"] = dict(text_embeds=torch.empty(batch, 1280, device="meta"),
                                       time_ids=torch.empty(batch, 6, device="meta"))
    net(x, t, encoder_hidden_states=ctx, **kw)
    for h in hooks:
        h.remove()
    return recs


# ----------------------------------------------------------------------------- activation tables
def _group_params(n, G, bits, key, seed, lo_scale=3.0, hi_scale=3.0):
    """Per-channel (δ, z) with ≤G distinct pairs and an arbitrary, non-contiguous channel→group map —
    what ``done_group_num`` (quant_layer.py:315-429, 'minmax' mode) leaves behind: per-cluster
    δ = (max−min)/(2^b−1) (≥1e-8), z = rne(−min/δ)."""
    lo = -(0.3 + lo_scale * named_rand(key + "|lo", (n,), seed))
    hi = 0.3 + hi_scale * named_rand(key + "|hi", (n,), seed)
    if G <= 1:
        labels = torch.zeros(n, dtype=torch.int64)
        G = 1
    else:
        # cluster-like labels: rank channels by (hi-lo) with noise, cut at random quantiles → uneven groups
        score = (hi - lo) + 0.5 * named_randn(key + "|noise", (n,), seed)
        order = torch.argsort(score)
        cuts = torch.sort(named_rand(key + "|cuts", (G - 1,), seed))[0]
        bounds = (cuts * n).long()
        labels = torch.empty(n, dtype=torch.int64)
        labels[order] = torch.bucketize(torch.arange(n), bounds, right=True)
    delta = torch.empty(n)
    zp = torch.empty(n)
    levels = float(2 ** bits - 1)
    for g in range(G):
        m = labels == g
        if not bool(m.any()):
            continue
        gmin, gmax = lo[m].min(), hi[m].max()
        d = torch.clamp((gmax - gmin) / levels, min=1e-8)
        delta[m] = d
        zp[m] = torch.round(-gmin / d)
    return delta, zp


def synth_act_entry(rec, bits, G, slot, seed):
    """(δ, z) tensors in the reference's shapes for one quantizer at one timestep slot."""
    key = "%s|slot%d" % (rec["name"], slot)
    kind = rec["kind"]
    pick = float(named_rand(key + "|axis", (1,), seed))
    if kind == "linear2d" or G <= 1:
        d, z = _group_params(1, 1, bits, key, seed)
        return d.reshape(()), z.reshape(())
    if kind == "linear3d":
        if pick < 0.6:                                     # per-K (in-channel) : view(1,1,-1)
            d, z = _group_params(rec["K"], G, bits, key, seed)
            return d.view(1, 1, -1), z.view(1, 1, -1)
        d, z = _group_params(rec["T"], G, bits, key, seed)  # per-token : view(1,-1,1)
        return d.view(1, -1, 1), z.view(1, -1, 1)
    if kind == "conv":
        if pick < 0.6:                                     # per (channel,tap) on the unfolded dim 1
            d, z = _group_params(rec["K"], G, bits, key, seed)
            return d.view(1, -1, 1), z.view(1, -1, 1)
        d, z = _group_params(rec["L"], G, bits, key, seed)  # per spatial position : view(1,1,-1)
        return d.view(1, 1, -1), z.view(1, 1, -1)
    if kind == "attn":
        if pick < 0.5:                                     # per head-dim : view(1,1,-1) on [B,H,T,D]
            d, z = _group_params(rec["D"], G, bits, key, seed)
            return d.view(1, 1, -1), z.view(1, 1, -1)
        d, z = _group_params(rec["T"], G, bits, key, seed)  # per token : view(1,-1,1)
        return d.view(1, -1, 1), z.view(1, -1, 1)
    raise ValueError(kind)


def synth_act_slot(arch, bits, G, slot, seed=0, batch=2, res=None, start_peak=False, uniform_softmax=False,
                   recs=None):
    """One ``act_<slot>`` dict. With ``start_peak`` the cross-attention ``aqtizer_k`` sees T−1 tokens
    (token 0 bypasses the quantizer, sd.py:176-180)."""
    recs = recs if recs is not None else enumerate_act_quantizers(arch, batch, res)
    out = OrderedDict()
    for rec in recs:
        r = dict(rec)
        if start_peak and r["kind"] == "attn" and r.get("cross") and r["which"] == "k":
            r["T"] = r["T"] - 1
        d, z = synth_act_entry(r, bits, G, slot, seed)
        out["model.%s.delta" % r["name"]] = d
        out["model.%s.zero_point" % r["name"]] = z
    if uniform_softmax:
        # UniformAffineQuantizer aqtizer_w with always_zero (quant_block.py:145-156): scalar δ, z=0
        for rec in recs:
            if rec["kind"] == "attn" and rec["which"] == "q":
                base = rec["name"].rsplit(".", 1)[0]
                out["model.%s.aqtizer_w.delta" % base] = torch.tensor(1.0 / (2 ** bits - 1))
                out["model.%s.aqtizer_w.zero_point" % base] = torch.tensor(0.0)
    return out


# ----------------------------------------------------------------------------- weight side
def channel_minmax(w: torch.Tensor, bits: int):
    """Vectorised per-output-channel ``minmax`` (quant_layer.py:22-38 applied per channel by
    :253-264): δ = (max(w,0)−min(w,0))/(2^b−1) clamped ≥1e-8 ; z = rne(−min/δ)."""
    flat = w.reshape(w.shape[0], -1).double()
    mn = torch.clamp(flat.min(dim=1)[0], max=0.0)
    mx = torch.clamp(flat.max(dim=1)[0], min=0.0)
    # the reference computes float(x_max - x_min) / (level-1) in python doubles, then casts to fp32
    delta = ((mx - mn) / float(2 ** bits - 1)).float()
    delta = torch.where(delta < 1e-8, torch.full_like(delta, 1e-8), delta)
    zp = torch.round(-mn.float() / delta)
    shape = (-1,) + (1,) * (w.dim() - 1)
    return delta.view(shape), zp.view(shape)


def synth_weight_ckpt(arch="sd", wbits=4, seed=0, adaround=False, delta_jitter=True):
    """The ``'weight'`` state-dict (1250 keys for SD)."""
    sd = synth_state_dict(arch, seed)
    with torch.device("meta"):
        skel = UNet2DConditionModel(arch)
    quant_paths = [n for n, m in skel.named_modules() if isinstance(m, (nn.Linear, nn.Conv2d))]
    out = OrderedDict()
    for k, v in sd.items():
        path, leaf = k.rsplit(".", 1)
        if path in quant_paths:
            out["model.%s.%s" % (path, "w" if leaf == "weight" else "b")] = v
            if leaf == "weight":
                d, z = channel_minmax(v, wbits)
                if delta_jitter:   # learned / searched scales differ from the self-initialised ones
                    d = d * (0.9 + 0.1 * named_rand(path + "|wjit", (v.shape[0],), seed)).view(d.shape)
                out["model.%s.wqtizer.delta" % path] = d
                out["model.%s.wqtizer.zero_point" % path] = z
                if adaround and path not in ("conv_in", "conv_out"):
                    out["model.%s.wqtizer.alpha" % path] = named_randn(path + "|alpha", v.shape, seed)
        else:
            out["model." + k] = v
    return out


def slot_list(num_slots):
    """``num_slots`` as an int (act_0 .. act_{n-1}) or an explicit collection of slot ids (act_0 is always written:
    load_act_ckpt_with_difference_shape reads it, calibration.py:268-291)."""
    if isinstance(num_slots, int):
        return list(range(num_slots))
    return sorted(set(int(s) for s in num_slots) | {0})


class _one_thread:
    """The quantizer-table generators issue ~100 k torch ops on tensors of a few hundred elements: with the intra-op pool of
    a 64-thread host each costs milliseconds (50 slots of SD tables: 118 s, 42 of them in torch.bucketize alone); on one
    thread the same values (bit for bit) take a third of the time."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(1)

    def __exit__(self, *a):
        torch.set_num_threads(self.n)


def build_cali_ckpt(arch="sd", wbits=4, abits=8, G=16, num_slots=1, seed=0, batch=2, res=None,
                    start_peak=False, uniform_softmax=False, adaround=False, with_act=True):
    """The merged checkpoint as a dict (what write_cali_ckpt saves)."""
    ck = OrderedDict()
    if with_act:
        recs = enumerate_act_quantizers(arch, batch, res)
        with _one_thread():
            for s in slot_list(num_slots):
                ck["act_%d" % s] = synth_act_slot(arch, abits, G, s, seed, batch, res, start_peak, uniform_softmax, recs)
    ck["weight"] = synth_weight_ckpt(arch, wbits, seed, adaround)
    return ck


def write_cali_ckpt(path, arch="sd", wbits=4, abits=8, G=16, num_slots=1, seed=0, batch=2, res=None,
                    start_peak=False, uniform_softmax=False, adaround=False, with_act=True):
    torch.save(build_cali_ckpt(arch, wbits, abits, G, num_slots, seed, batch, res, start_peak, uniform_softmax,
                               adaround, with_act), path)
    return path
