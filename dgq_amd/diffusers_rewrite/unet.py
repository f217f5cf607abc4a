"""Table-driven SD-v1.4 / SDXL UNet definition (floating-point graph).

Replaces the reference's two hand-flattened files ``diffusers_rewrite/sd.py`` and
``diffusers_rewrite/sdxl.py`` with ONE architecture table (``ARCH``) and generic
builders.  What is kept identical, because the ``cali_ckpt`` format depends on it
(SURVEY.md §5.4): the module tree / state-dict key names (the public Hugging Face
names), parameter shapes, and the floating-point semantics of every node:

  * sinusoidal ``Timesteps`` with cos‖sin order            (sd.py:20-39)
  * ``ResnetBlock2D``  GN32(eps 1e-5)→SiLU→conv→+temb→GN→SiLU→conv (+1×1 shortcut iff Cin≠Cout)  (sd.py:57-99)
  * ``Transformer2DModel`` GN32(eps **1e-6**), proj_in/out = 1×1 Conv2d (SD, sd.py:273-305) or
    Linear applied on the token layout (SDXL, sdxl.py:296-326)
  * ``BasicTransformerBlock`` LN→self-attn, LN→cross-attn, LN→GEGLU FF  (sd.py:239-270)
  * heads: SD = 8 heads (head_dim = C/8, sd.py:243-245); SDXL head_dim = 64 (sdxl.py:266-268)
  * SDXL ``add_embedding(text_embeds ‖ Timesteps(256)(time_ids))``  (sdxl.py:567-577)
  * forward returns a one-element list ``[sample]``                  (sd.py:620)

4-D activations are kept in ``torch.channels_last`` memory format (logical NCHW,
physical NHWC) so that the HIP kernels of ``dgq_amd.csrc`` see row-major
``[B·H·W, C]`` matrices without copies; ``permute(0,2,3,1).reshape`` is then a view.
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F

def _cat_channels(a, b):
    """skip concatenation; tensors that carry GroupNorm partials from the GEMM that produced them (dgq_amd.ops) keep them"""
    if getattr(a, "_dgq_gn", None) is not None and getattr(b, "_dgq_gn", None) is not None:
        from .. import ops
        return ops.cat_channels(a, b)
    return torch.cat([a, b], dim=1)


class _SkipEntry:
    __slots__ = ("tensor", "ver", "buf", "c1", "h_inplace")


class _CatInPlace:
    """The skip concatenations of the up path (sd.py:558-613: ``torch.cat([hidden_states, res_hidden_states], dim=1)`` in front of every
    up resnet) without a concatenation launch: for each skip tensor a channels-last buffer [B][H][W][C1 + C2] exists from the moment the
    skip is produced; the skip's producing layer stores its rows into [..., C1:] AS WELL (dgq_gemm_extra_t.y2 / dgq_conv2d_f32w's y2),
    the layer that produces the up path's hidden state stores INTO [..., :C1] (its ``y`` with the buffer's row pitch) — both through
    ``ops.OutputRedirect``, taken by the modules' final layer calls.  Whatever is not placed that way (a module without such a call, a
    layer tap, an eager FP block) is concatenated as before: ``join`` checks where the tensors actually are."""

    def __init__(self, unet, dtype, device):
        from .. import ops
        self.ops = ops
        self.dtype, self.device = dtype, device
        tot = [r.norm1.num_channels for blk in unet.up_blocks for r in blk.resnets]     # up resnets in execution order
        self.ctot = tot[::-1]                                                           # by skip index (skips are popped last-first)
        self.entries = []

    def _views(self, buf, c1):
        b2 = buf.view(-1, buf.shape[-1])
        return b2[:, :c1], b2[:, c1:]

    def produce_skip(self, fn, shape):
        """run fn() (the module call that produces the next skip tensor, of logical NCHW ``shape``) with its second copy redirected"""
        i = len(self.entries)
        e = _SkipEntry()
        e.buf, e.c1, e.h_inplace = None, 0, False
        B, C2, H, W = shape
        ok = i < len(self.ctot) and self.ctot[i] > C2
        if ok:
            e.c1 = self.ctot[i] - C2
            buf = torch.empty((B, H, W, self.ctot[i]), dtype=self.dtype, device=self.device)
            rd = self.ops.OutputRedirect(out2=self._views(buf, e.c1)[1])
            self.ops.set_redirect(rd)
            try:
                y = fn()
            finally:
                self.ops.set_redirect(None)
            if rd.taken and tuple(y.shape) == tuple(shape):
                e.buf = buf
        else:
            y = fn()
        e.tensor, e.ver = y, y._version
        self.entries.append(e)
        return y

    def produce_h(self, fn):
        """run fn() (the module call whose result is the hidden state of the NEXT concatenation) with its output redirected INTO the
        buffer of the skip that concatenation pops"""
        e = self.entries[-1] if self.entries else None
        if e is None or e.buf is None:
            return fn()
        rd = self.ops.OutputRedirect(out=self._views(e.buf, e.c1)[0])
        self.ops.set_redirect(rd)
        try:
            y = fn()
        finally:
            self.ops.set_redirect(None)
        e.h_inplace = bool(rd.taken and y.data_ptr() == e.buf.data_ptr() and y.shape[1] == e.c1 and y.shape[2:] == e.tensor.shape[2:])
        return y

    def join(self, h):
        """torch.cat([h, skip], dim=1) for the skip popped now — the buffer itself where both halves already sit in it"""
        e = self.entries.pop()
        if e.buf is not None and e.h_inplace and e.tensor._version == e.ver and h.data_ptr() == e.buf.data_ptr():
            y = e.buf.permute(0, 3, 1, 2)
            ga, gb = self.ops._gn_of(h), self.ops._gn_of(e.tensor)
            if ga is not None and gb is not None and len(ga["parts"]) == 1 and len(gb["parts"]) == 1 and ga["B"] == gb["B"] and ga["HW"] == gb["HW"]:
                y._dgq_gn = dict(parts=ga["parts"] + gb["parts"], B=ga["B"], HW=ga["HW"], C=ga["C"] + gb["C"], ver=y._version)
            return y
        return _cat_channels(h, e.tensor)


def _out_channels(conv):
    w = getattr(conv, "w", None)
    return int(w.shape[0]) if w is not None else int(conv.out_channels)


ARCH = {
    # name: dict(block_out, down=(kind, n_tf_layers, has_down), up=(kind, n_tf_layers, has_up), ...)
    "sd": dict(
        sample_size=64, ctx_dim=768, heads=8, head_dim=None, proj="conv", mid_layers=1,
        block_out=(320, 640, 1280, 1280),
        down=(("xattn", 1, True), ("xattn", 1, True), ("xattn", 1, True), ("plain", 0, False)),
        up=(("plain", 0, True), ("xattn", 1, True), ("xattn", 1, True), ("xattn", 1, False)),
        addition_time_embed_dim=None, add_in=None,
    ),
    # a miniature of the SD layout (same block kinds, conv projections, 8 heads) for fast tests
    "tiny": dict(
        sample_size=16, ctx_dim=64, heads=8, head_dim=None, proj="conv", mid_layers=1,
        block_out=(64, 128),
        down=(("xattn", 1, True), ("plain", 0, False)),
        up=(("plain", 0, True), ("xattn", 1, False)),
        addition_time_embed_dim=None, add_in=None, temb_dim=128,
    ),
    # two-level model with the REFERENCE's hard-wired widths (temb 1280, ctx 768, 8 heads): the one shape that can be built
    # from the reference's own block classes too — used to golden-check the calibration producer (tests/golden/make_golden.py calib)
    "mini": dict(
        sample_size=16, ctx_dim=768, heads=8, head_dim=None, proj="conv", mid_layers=1,
        block_out=(64, 64),
        down=(("xattn", 1, True), ("plain", 0, False)),
        up=(("plain", 0, True), ("xattn", 1, False)),
        addition_time_embed_dim=None, add_in=None, temb_dim=1280,
    ),
    "sdxl": dict(
        sample_size=128, ctx_dim=2048, heads=None, head_dim=64, proj="linear", mid_layers=10,
        block_out=(320, 640, 1280),
        down=(("plain", 0, True), ("xattn", 2, True), ("xattn", 10, False)),
        up=(("xattn", 10, True), ("xattn", 2, True), ("plain", 0, False)),
        addition_time_embed_dim=256, add_in=2816,
    ),
}


def _cl(x):
    return x.contiguous(memory_format=torch.channels_last) if x.dim() == 4 else x


class _Config(dict):
    __getattr__ = dict.get


class Timesteps(nn.Module):
    def __init__(self, num_channels: int = 320):
        super().__init__()
        self.num_channels = num_channels

    def forward(self, timesteps):
        if timesteps.is_cuda and timesteps.dim() == 1 and timesteps.dtype in (torch.int64, torch.float32) and _glue_on():
            from .. import ops
            return ops.timestep_embedding(timesteps, self.num_channels)       # the eight-kernel chain below as one launch
        half = self.num_channels // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
                          / (half - 0.0))
        ang = timesteps[:, None].float() * freqs[None, :]
        return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.linear_1 = nn.Linear(in_features, out_features)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(out_features, out_features)

    def forward(self, x):
        h = self.linear_1(x)
        fused = getattr(self.linear_2, "forward_fused", None)
        if fused is not None and _glue_on() and isinstance(self.act, nn.SiLU) and h.is_cuda:
            return fused(h, pre_act=1)                                         # SiLU in linear_2's load (as the resnets' time_emb_proj)
        return self.linear_2(self.act(h))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim=1280):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=1e-5)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(32, cout, eps=1e-5)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = nn.Conv2d(cin, cout, 1, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(self.nonlinearity(self.norm1(x)))
        h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    def __init__(self, inner_dim, cross_attention_dim=None, num_heads=None, head_dim=None):
        super().__init__()
        if num_heads is None:
            self.head_dim = head_dim or 64
            self.num_heads = inner_dim // self.head_dim
        else:
            self.num_heads = num_heads
            self.head_dim = inner_dim // num_heads
        self.scale = self.head_dim ** -0.5
        cdim = cross_attention_dim or inner_dim
        self.to_q = nn.Linear(inner_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(cdim, inner_dim, bias=False)
        self.to_v = nn.Linear(cdim, inner_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner_dim, inner_dim), nn.Dropout(0.0)])

    def split_heads(self, t):
        return t.view(t.size(0), t.size(1), self.num_heads, self.head_dim).transpose(1, 2)

    def forward(self, hidden_states, encoder_hidden_states=None):
        src = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q, k, v = (self.split_heads(p) for p in (self.to_q(hidden_states), self.to_k(src), self.to_v(src)))
        p = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) * self.scale, dim=-1)
        o = torch.matmul(p, v).transpose(1, 2).reshape(hidden_states.size(0), hidden_states.size(1), -1)
        return self.to_out[1](self.to_out[0](o))


class GEGLU(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.proj = nn.Linear(cin, cout * 2)

    def forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, ctx_dim, heads=None, head_dim=None):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, heads, head_dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, ctx_dim, heads, head_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, encoder_hidden_states=None):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), encoder_hidden_states)
        return x + self.ff(self.norm3(x))


def _residual_fusion_on():
    from ..quant import quant_block
    return quant_block.FUSION and quant_block._F_RES


def _fusion_on():
    from ..quant import quant_block
    return quant_block.FUSION and quant_block.FUSE_NORM


def _glue_on():
    """DGQ_GLUE=0 (A/B runs): timestep embedding and its SiLU as the torch kernels"""
    import os
    return _fusion_on()


class Transformer2DModel(nn.Module):
    def __init__(self, channels, n_layers, ctx_dim, heads, head_dim, proj):
        super().__init__()
        self.proj_kind = proj
        self.norm = nn.GroupNorm(32, channels, eps=1e-6)
        mk = (lambda: nn.Conv2d(channels, channels, 1, 1)) if proj == "conv" else (lambda: nn.Linear(channels, channels))
        self.proj_in = mk()
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(channels, ctx_dim, heads, head_dim) for _ in range(n_layers)])
        self.proj_out = mk()

    def forward(self, x, encoder_hidden_states=None):
        b, c, hh, ww = x.shape
        res = x
        fuse = getattr(self.proj_in, "can_fuse_prenorm", None)
        fuse_t = getattr(self.proj_in, "can_fuse_tokens", None)
        tokens_done = False
        if self.proj_kind == "conv" and fuse is not None and _fusion_on() and fuse(x):
            h = self.proj_in.forward_prenorm(x, self.norm, silu=False)     # GroupNorm folded into the quantise-on-load pass
        elif self.proj_kind != "conv" and fuse_t is not None and _fusion_on() and fuse_t(x):
            h = self.proj_in.forward_prenorm_tokens(x, self.norm)          # likewise for the Linear projection (SDXL): tokens
            tokens_done = True
        else:
            h = self.norm(x)
            if self.proj_kind == "conv":
                h = self.proj_in(h)
        if not tokens_done:
            h = _cl(h).permute(0, 2, 3, 1).reshape(b, hh * ww, c)
            if self.proj_kind != "conv":
                h = self.proj_in(h)
        for blk in self.transformer_blocks:
            h = blk(h, encoder_hidden_states=encoder_hidden_states)
        if self.proj_kind != "conv":
            fo = getattr(self.proj_out, "can_fuse_tokens", None)
            if fo is not None and _residual_fusion_on() and fo(h):
                return self.proj_out.forward_residual_tokens(h, res, final=True)   # h + res in proj_out's GEMM epilogue (+ GroupNorm partials)
            h = self.proj_out(h)
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)      # NHWC storage viewed as NCHW
        if self.proj_kind == "conv":
            fr = getattr(self.proj_out, "forward_residual", None)
            if fr is not None and _residual_fusion_on():
                return fr(h, res, final=True)                      # h + res in proj_out's GEMM epilogue (the module's own result)
            h = self.proj_out(h)
        return h + res


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 2, 1)

    def forward(self, x):
        ff = getattr(self.conv, "forward_final", None)
        return ff(x) if (ff is not None and _fusion_on()) else self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 1, 1)

    def forward(self, x):
        folded = getattr(self.conv, "forward_upsampled", None)
        if folded is not None and _fusion_on():
            return folded(x, final=True)                       # QuantLayer: the interpolate folded into the conv's quantise-on-load pass
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Stage(nn.Module):
    """One down / mid / up stage: resnets [+ attentions] [+ down/up-sampler]."""

    def __init__(self, res_io, tf_layers, a, sampler=None):
        super().__init__()
        # registration order as the reference's block classes (attentions, resnets, samplers — sd.py:362-376,400-416): the
        # weight-PTQ driver walks named_children() and reconstructs in that order (calibration.py:113-141)
        if tf_layers:
            n_att = len(res_io) if sampler != "mid" else len(res_io) - 1
            self.attentions = nn.ModuleList(
                [Transformer2DModel(res_io[0][1], tf_layers, a["ctx_dim"], a["heads"], a["head_dim"], a["proj"])
                 for _ in range(n_att)])
        else:
            self.attentions = None
        self.resnets = nn.ModuleList([ResnetBlock2D(i, o, a.get("temb_dim", 1280)) for i, o in res_io])
        c = res_io[-1][1]
        self.downsamplers = nn.ModuleList([Downsample2D(c)]) if sampler == "down" else None
        self.upsamplers = nn.ModuleList([Upsample2D(c)]) if sampler == "up" else None
        if self.attentions is None:
            del self.attentions
        if self.downsamplers is None:
            del self.downsamplers
        if self.upsamplers is None:
            del self.upsamplers

    def _att(self, j, h, ctx):
        att = getattr(self, "attentions", None)
        return att[j](h, encoder_hidden_states=ctx) if att is not None else h

    def run_down(self, h, temb, ctx, skips, cat=None):
        """cat (a _CatInPlace, or None): every skip's producer also stores into the skip's slot of its concatenation buffer"""
        has_att = hasattr(self, "attentions")
        for j, r in enumerate(self.resnets):
            if cat is None:
                h = self._att(j, r(h, temb), ctx)
            else:
                shape = (h.shape[0], _out_channels(r.conv2), h.shape[2], h.shape[3])
                if has_att:
                    hr = r(h, temb)
                    h = cat.produce_skip(lambda: self._att(j, hr, ctx), shape)
                else:
                    hp = h
                    h = cat.produce_skip(lambda: r(hp, temb), shape)
            skips.append(h)
        if hasattr(self, "downsamplers"):
            if cat is None:
                h = self.downsamplers[0](h)
            else:
                hp = h
                shape = (h.shape[0], h.shape[1], (h.shape[2] - 1) // 2 + 1, (h.shape[3] - 1) // 2 + 1)
                h = cat.produce_skip(lambda: self.downsamplers[0](hp), shape)
            skips.append(h)
        return h

    def run_mid(self, h, temb, ctx, cat=None):
        h = self.resnets[0](h, temb)
        rest = list(self.resnets[1:])
        for j, r in enumerate(rest):
            ha = self._att(j, h, ctx)
            if cat is not None and j == len(rest) - 1:
                h = cat.produce_h(lambda: r(ha, temb))             # the hidden state of the first up concatenation
            else:
                h = r(ha, temb)
        return h

    def run_up(self, h, temb, ctx, skips, cat=None):
        has_att, has_up = hasattr(self, "attentions"), hasattr(self, "upsamplers")
        n = len(self.resnets)
        for j, r in enumerate(self.resnets):
            if cat is None:
                h = self._att(j, r(_cat_channels(h, skips.pop()), temb), ctx)
                continue
            skips.pop()
            x = cat.join(h)
            feeds_cat = not (j == n - 1 and has_up)                # the last hidden state of a stage goes through its upsampler first
            if has_att:
                hr = r(x, temb)
                h = cat.produce_h(lambda: self._att(j, hr, ctx)) if feeds_cat else self._att(j, hr, ctx)
            else:
                h = cat.produce_h(lambda: r(x, temb)) if feeds_cat else r(x, temb)
        if has_up:
            hp = h
            h = cat.produce_h(lambda: self.upsamplers[0](hp)) if cat is not None else self.upsamplers[0](h)
        return h


class UNet2DConditionModel(nn.Module):
    """``arch`` in {"sd","sdxl"}; default from env ``DIFFUSERS_REWRITE`` like the reference
    (diffusers_rewrite/__init__.py:1-6)."""

    def __init__(self, arch=None):
        super().__init__()
        import os
        arch = arch or os.environ.get("DIFFUSERS_REWRITE", "sd")
        a = ARCH[arch]
        self.arch = arch
        self.config = _Config(in_channels=4, sample_size=a["sample_size"], time_cond_proj_dim=None)
        if a["addition_time_embed_dim"]:
            self.config["addition_time_embed_dim"] = a["addition_time_embed_dim"]
        bo = a["block_out"]
        self.conv_in = nn.Conv2d(4, bo[0], 3, 1, 1)
        self.time_proj = Timesteps(bo[0])
        td = a.get("temb_dim", 1280)
        self.time_embedding = TimestepEmbedding(bo[0], td)
        if a["add_in"]:
            self.add_time_proj = Timesteps(a["addition_time_embed_dim"])
            self.add_embedding = TimestepEmbedding(a["add_in"], td)
        # skip-channel bookkeeping: every down resnet output and every downsampler output is a skip
        skip_ch = [bo[0]]
        downs, cin = [], bo[0]
        for (kind, nl, has_down), cout in zip(a["down"], bo):
            downs.append(_Stage([(cin, cout), (cout, cout)], nl if kind == "xattn" else 0, a,
                                "down" if has_down else None))
            skip_ch += [cout, cout] + ([cout] if has_down else [])
            cin = cout
        self.down_blocks = nn.ModuleList(downs)
        ups, prev = [], cin
        for (kind, nl, has_up), cout in zip(a["up"], reversed(bo)):
            io = []
            for _ in range(3):
                io.append((prev + skip_ch.pop(), cout))
                prev = cout
            ups.append(_Stage(io, nl if kind == "xattn" else 0, a, "up" if has_up else None))
        self.up_blocks = nn.ModuleList(ups)                         # registered before mid_block, as sd.py:509-541
        self.mid_block = _Stage([(cin, cin), (cin, cin)], a["mid_layers"], a, "mid")
        self.conv_norm_out = nn.GroupNorm(32, bo[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(bo[0], 4, 3, 1, 1)

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def forward(self, sample, timesteps, encoder_hidden_states=None, added_cond_kwargs=None, **kwargs):
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.int64, device=sample.device)
        timesteps = timesteps.to(sample.device).expand(sample.shape[0])
        emb = self.time_embedding(self.time_proj(timesteps).to(dtype=sample.dtype))
        if hasattr(self, "add_embedding"):
            text_embeds = added_cond_kwargs.get("text_embeds")
            time_ids = added_cond_kwargs.get("time_ids")
            te = self.add_time_proj(time_ids.flatten()).reshape((text_embeds.shape[0], -1))
            emb = emb + self.add_embedding(torch.cat([text_embeds, te], dim=-1).to(emb.dtype))
        ctx = encoder_hidden_states
        cat = None
        if _fusion_on() and sample.is_cuda and not torch.is_grad_enabled():
            from .. import ops
            if ops.CAT_INPLACE:
                cat = _CatInPlace(self, sample.dtype, sample.device)
        x0 = _cl(sample)
        ff = getattr(self.conv_in, "forward_final", None)
        if cat is not None and ff is not None:
            h = cat.produce_skip(lambda: ff(x0), (x0.shape[0], _out_channels(self.conv_in), x0.shape[2], x0.shape[3]))
        else:
            cat = None
            h = self.conv_in(x0)
        skips = [h]
        for blk in self.down_blocks:
            h = blk.run_down(h, emb, ctx, skips, cat)
        h = self.mid_block.run_mid(h, emb, ctx, cat)
        for blk in self.up_blocks:
            h = blk.run_up(h, emb, ctx, skips, cat)
        fuse = getattr(self.conv_out, "can_fuse_prenorm", None)
        if fuse is not None and _fusion_on() and isinstance(self.conv_act, nn.SiLU) and fuse(h):
            return [self.conv_out.forward_prenorm(h, self.conv_norm_out, silu=True)]
        fuse_fp = getattr(self.conv_out, "can_fuse_prenorm_fp", None)
        if fuse_fp is not None and _fusion_on() and isinstance(self.conv_act, nn.SiLU) and fuse_fp(h):
            return [self.conv_out.forward_prenorm_fp(h, self.conv_norm_out, silu=True)]   # FP conv_out: norm + SiLU in its load
        return [self.conv_out(self.conv_act(self.conv_norm_out(h)))]
