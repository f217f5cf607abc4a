"""TEST INFRASTRUCTURE — CPU oracle for the DGQ quantized-UNet hot path.

A functional PyTorch-CPU (fp32) restatement of what the reference executes for one
``QuantModel.forward`` — no nn.Modules, no module surgery: it interprets the reference's
``cali_ckpt`` dictionaries directly.  Every function cites the reference
file:line it restates (paths relative to /root/reference).

Pinned (not "parity unpinned"): ``tests/golden/make_golden.py`` runs the REAL reference
(imported on CPU through ``oracle/ref_harness.py``) on the same seeded inputs and commits its
outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against
those vectors on every CPU test run (the reference has no tests/golden vectors of its own,
SURVEY.md §4).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the reported CPU baseline.  The product
(``dgq_amd/``) never imports it and has no CPU fallback.

The op sequence is deliberately *reference-faithful* (it is also the timed CPU baseline):
weights are re-quantised on every call (quant_layer.py:642-643), grouped convs go through
``F.unfold`` + dequantised fp32 matmul (:630-638, :652-657), attention probabilities are
fully materialised and log-quantised elementwise (sd.py:183-201).
"""
import math
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- quantizers


def uaq(x, delta, zero_point, bits):
    """UniformAffineQuantizer.forward, inference branch, asymmetric (quant_layer.py:295-299):
    δ·(clamp(rne(x/δ)+z, 0, 2^b−1) − z). ``torch.round`` is round-half-to-even; the reference
    divides (does not multiply by a reciprocal)."""
    q = torch.clamp(torch.round(x / delta) + zero_point, 0, 2 ** bits - 1)
    return delta * (q - zero_point)


def uaq_codes(x, delta, zero_point, bits):
    """The integer codes inside ``uaq`` (integer-valued fp32)."""
    return torch.clamp(torch.round(x / delta) + zero_point, 0, 2 ** bits - 1)


def minmax_scalar(x, bits, always_zero=False):
    """minmax (quant_layer.py:22-38), asymmetric: python-double range arithmetic, δ≥1e-8,
    z = rne(−min/δ) (0 for always_zero)."""
    level = 2 ** bits
    x_min, x_max = min(x.min().item(), 0), max(x.max().item(), 0)
    delta = torch.tensor(float(x_max - x_min) / (level - 1))
    if always_zero:
        delta = torch.tensor(float(x_max) / (level - 1))
    if delta < 1e-8:
        delta = torch.tensor(1e-8)
    zp = torch.round(-torch.tensor(float(x_min)) / delta) if not always_zero else torch.tensor(0.0)
    return delta.float(), zp.float()


def minmax_channel(w, bits):
    """_init_quantization_param(channel_wise=True) (quant_layer.py:253-264): ``minmax`` per
    output channel, reshaped to [N,1(,1,1)]."""
    n = w.shape[0]
    d = torch.empty(n)
    z = torch.empty(n)
    for c in range(n):
        d[c], z[c] = minmax_scalar(w[c], bits)
    shp = (-1,) + (1,) * (w.dim() - 1)
    return d.view(shp), z.view(shp)


def adaround_hard(w, delta, zero_point, alpha, bits):
    """AdaRoundQuantizer.forward, LEARNED_HARD_SIGMOID with soft_tgt=False
    (adaptive_rounding.py:51,58-70): δ·(clamp(floor(w/δ)+(α≥0)+z, 0, 2^b−1) − z)."""
    q = torch.clamp(torch.floor(w / delta) + (alpha >= 0).float() + zero_point, 0, 2 ** bits - 1)
    return delta * (q - zero_point)


def adaround_soft_target(alpha, zeta=1.1, gamma=-0.1):
    """h(α) = clamp(sigmoid(α)·(ζ − γ) + γ, 0, 1) — quant/adaptive_rounding.py:39-40 (get_soft_tgt)."""
    return torch.clamp(torch.sigmoid(alpha) * (zeta - gamma) + gamma, 0, 1)


def adaround_soft(w, delta, zero_point, alpha, bits):
    """AdaRoundQuantizer.forward with soft_tgt = True — quant/adaptive_rounding.py:50-57,66-70; differentiable in alpha
    through torch autograd (the checker of dgq_adaround_soft_fwd / _bwd)."""
    x_int = torch.floor(w / delta) + adaround_soft_target(alpha)
    x_q = torch.clamp(x_int + zero_point, 0, 2 ** bits - 1)
    return delta * (x_q - zero_point)


def adaround_round_loss(alpha, b):
    """Σ (1 − |2h(α) − 1|^b) — quant/reconstruction_util.py:68-70 (the caller multiplies by the loss weight)."""
    return (1 - ((adaround_soft_target(alpha) - .5).abs() * 2).pow(b)).sum()


def log_quant(x, delta, bits):
    """T2ILogQuantizer.forward body (quant_layer_text.py:101-105):
    δ·2^(−clamp(rne(−log2(x/δ)), 0, 2^b−1)); log2(0) = −inf → code clamps to 2^b−1."""
    q = torch.clamp(torch.round(-1 * torch.log2(x / delta)), 0, 2 ** bits - 1)
    return 2 ** (-1 * q) * delta


def log_quant_init_delta(x, bits):
    """T2ILogQuantizer._init_quantization_param (quant_layer_text.py:49-76): best of the
    0.999 / 0.9999 / 0.99999 quantiles under an L2 score."""
    best, delta = 1e10, x.max()
    for pct in (0.999, 0.9999, 0.99999):
        try:
            nd = torch.quantile(x.reshape(-1), pct)
        except Exception:
            import numpy as np
            nd = torch.tensor(np.percentile(x.reshape(-1).numpy(), pct * 100), dtype=torch.float32)
        xq = log_quant(x, nd, bits)
        score = (x - xq).abs().pow(2).mean()
        if score < best:
            best, delta = score, nd
    return delta


# ----------------------------------------------------------------------------- checkpoint view


def slot_for_timestep(t, num_inference_steps):
    """Time-aware slot (calibration.py:301-304): int((1000 − t)//(1000//N))."""
    return int((1000 - int(t)) // (1000 // num_inference_steps))


# (n transformer layers, has down/up-sampler) per block; the same tables as the reference's two model files
# (sd.py:493-544, sdxl.py:505-556) plus a miniature of the SD layout for fast tests.
ORACLE_ARCH = {
    "sd": dict(bo=(320, 640, 1280, 1280), down=((1, True), (1, True), (1, True), (0, False)),
               up=((0, True), (1, True), (1, True), (1, False)), proj="conv", heads=lambda c: 8, mid=1),
    "sdxl": dict(bo=(320, 640, 1280), down=((0, True), (2, True), (10, False)),
                 up=((10, True), (2, True), (0, False)), proj="linear", heads=lambda c: c // 64, mid=10),
    "mini": dict(bo=(64, 64), down=((1, True), (0, False)), up=((0, True), (1, False)), proj="conv",
                 heads=lambda c: 8, mid=1),
    "tiny": dict(bo=(64, 128), down=((1, True), (0, False)), up=((0, True), (1, False)), proj="conv",
                 heads=lambda c: 8, mid=1),
}


class OracleConfig:
    def __init__(self, arch="sd", wbits=4, abits=8, use_wq=True, use_aq=True, softmax_bits=None,
                 t2i_log_quant=False, t2i_real_time=False, t2i_start_peak=False,
                 time_aware=False, num_inference_steps=50, use_group=True, exact_gemm=False):
        # exact_gemm: every contraction (F.linear, unfold-matmul, F.conv2d, Q·Kᵀ, P·V) is evaluated in float64 and
        # rounded to fp32 once — the "exact" target of the parity tests: independent of BLAS blocking / thread count,
        # which is all that separates two fp32 runs of the reference.  Quantizers, softmax, norms and
        # activations stay the reference's fp32 elementwise arithmetic.
        self.exact_gemm = exact_gemm
        self.arch, self.wbits, self.abits = arch, wbits, abits
        self.use_wq, self.use_aq = use_wq, use_aq
        self.softmax_bits = softmax_bits if softmax_bits is not None else abits
        self.t2i_log_quant, self.t2i_real_time, self.t2i_start_peak = t2i_log_quant, t2i_real_time, t2i_start_peak
        self.time_aware, self.num_inference_steps, self.use_group = time_aware, num_inference_steps, use_group


class OracleModel:
    """Interprets a merged ``cali_ckpt`` dict. ``fp_sd`` (HF keys) supplies the constructor-time
    FP weights used by conv_in / conv_out (``original_w`` is not in the ckpt)."""

    def __init__(self, ckpt, cfg: OracleConfig, fp_sd=None):
        self.cfg = cfg
        self.ck = ckpt
        self.w = ckpt["weight"] if "weight" in ckpt else ckpt
        self.fp_sd = fp_sd
        self.act = None            # current act_<slot> dict
        self.lazy = {}             # self-initialised quantizers (first-forward init, quant_layer.py:274-278)
        self.softmax_delta = {}    # non-real-time T2ILogQuantizer δ, keyed by attention path
        self._wq_cache = {}
        if cfg.use_aq and not cfg.time_aware and "act_0" in ckpt:
            self.act = ckpt["act_0"]                                  # calibration.py:313-325

    # -- parameter access ------------------------------------------------------------------
    def P(self, key):
        return self.w["model." + key]

    def has(self, key):
        return ("model." + key) in self.w

    def wq_params(self, path, w):
        kd = "model.%s.wqtizer.delta" % path
        if kd in self.w:
            return self.w[kd], self.w["model.%s.wqtizer.zero_point" % path]
        if path not in self._wq_cache:          # self-init on first forward (quant_layer.py:274-275)
            self._wq_cache[path] = minmax_channel(w, self.cfg.wbits)
        return self._wq_cache[path]

    def weight(self, path):
        """wqtizer(self.w) — executed on every call like the reference (quant_layer.py:642-643)."""
        w = self.P(path + ".w")
        if not self.cfg.use_wq:
            return w
        d, z = self.wq_params(path, w)
        ka = "model.%s.wqtizer.alpha" % path
        if ka in self.w:                                               # calibration.py:227-230
            return adaround_hard(w, d, z, self.w[ka], self.cfg.wbits)
        return uaq(w, d, z, self.cfg.wbits)

    def bias(self, path):
        k = "model.%s.b" % path
        return self.w[k] if k in self.w else None

    def aq(self, qname, x):
        """Activation quantizer ``qname`` ('<path>.aqtizer' | '<blk>.attnN.aqtizer_q' ...)."""
        if not self.cfg.use_aq:
            return x
        kd = "model.%s.delta" % qname
        if self.act is not None and kd in self.act:
            d, z = self.act[kd], self.act["model.%s.zero_point" % qname]
        else:
            if qname not in self.lazy:
                self.lazy[qname] = minmax_scalar(x, self.cfg.abits)
            d, z = self.lazy[qname]
        return uaq(x, d, z, self.cfg.abits)

    def act_codes(self, path, x, stride=1, padding=0):
        """Integer codes the activation quantizer of layer ``path`` assigns to the layer input ``x`` (on the unfolded
        operand for grouped convs, like ``conv``) — used by the parity statistics to count code flips between two runs."""
        kd = "model.%s.aqtizer.delta" % path
        if self.act is not None and kd in self.act:
            d, z = self.act[kd], self.act["model.%s.aqtizer.zero_point" % path]
        else:
            d, z = self.lazy[path + ".aqtizer"]
        if x.dim() == 4 and self.grouped(path):
            w = self.P(path + ".w")
            x = F.unfold(x, kernel_size=(w.shape[2], w.shape[3]), dilation=1, padding=padding, stride=stride)
        return uaq_codes(x, d, z, self.cfg.abits)

    def grouped(self, path):
        """use_group_num is switched on by load_act_ckpt_with_difference_shape when the ckpt δ is
        not the live scalar shape (calibration.py:268-291)."""
        if not (self.cfg.use_aq and self.cfg.use_group and self.act is not None):
            return False
        kd = "model.%s.aqtizer.delta" % path
        return kd in self.act and self.act[kd].dim() > 0

    # -- layers ----------------------------------------------------------------------------
    def linear(self, path, x):
        """QuantLayer.forward for nn.Linear (quant_layer.py:640-661)."""
        x = self.aq(path + ".aqtizer", x)
        if self.cfg.exact_gemm:
            b = self.bias(path)
            return F.linear(x.double(), self.weight(path).double(), b.double() if b is not None else None).float()
        return F.linear(x, self.weight(path), self.bias(path))

    def conv(self, path, x, stride=1, padding=0):
        """QuantLayer.forward for nn.Conv2d: grouped → unfold + fq on [B,C·kh·kw,L] + matmul
        (quant_layer.py:630-638, 652-657, 526-574); else native conv2d (:659)."""
        w = self.weight(path)
        b = self.bias(path)
        if self.grouped(path):
            kh, kw = w.shape[2], w.shape[3]
            cols = F.unfold(x, kernel_size=(kh, kw), dilation=1, padding=padding, stride=stride)
            cols = self.aq(path + ".aqtizer", cols)
            if self.cfg.exact_gemm:
                out = (w.view(w.shape[0], -1).double() @ cols.double()).float()
            else:
                out = w.view(w.shape[0], -1) @ cols
            ho = (x.shape[2] + 2 * padding - (kh - 1) - 1) // stride + 1
            wo = (x.shape[3] + 2 * padding - (kw - 1) - 1) // stride + 1
            out = out.view(x.shape[0], w.shape[0], ho, wo)
            if b is not None:
                out = out + b.view(1, -1, 1, 1)
            return out
        x = self.aq(path + ".aqtizer", x)
        if self.cfg.exact_gemm:
            return F.conv2d(x.double(), w.double(), b.double() if b is not None else None, stride=stride, padding=padding).float()
        return F.conv2d(x, w, b, stride=stride, padding=padding)

    def fp_conv(self, path, x, padding=1):
        """conv_in / conv_out: FP weights, no activation quantisation (quant_model.py:118-124)."""
        if self.fp_sd is not None:
            w, b = self.fp_sd[path + ".weight"], self.fp_sd[path + ".bias"]
        else:
            w, b = self.P(path + ".w"), self.P(path + ".b")
        return F.conv2d(x, w, b, stride=1, padding=padding)

    def group_norm(self, path, x, eps):
        return F.group_norm(x, 32, self.P(path + ".weight"), self.P(path + ".bias"), eps)

    def layer_norm(self, path, x):
        return F.layer_norm(x, (x.shape[-1],), self.P(path + ".weight"), self.P(path + ".bias"), 1e-5)

    # -- blocks ----------------------------------------------------------------------------
    def softmax_quant(self, apath, p):
        cfg = self.cfg
        if cfg.t2i_log_quant:
            if cfg.t2i_real_time:
                delta = p.max()                                       # quant_layer_text.py:96-97
            else:
                if apath not in self.softmax_delta:
                    self.softmax_delta[apath] = log_quant_init_delta(p, cfg.softmax_bits)
                delta = self.softmax_delta[apath]
            return log_quant(p, delta, cfg.softmax_bits)
        # UniformAffineQuantizer with always_zero (quant_block.py:145-156, quant_layer.py:32-37)
        kd = "model.%s.aqtizer_w.delta" % apath
        if self.act is not None and kd in self.act:
            d, z = self.act[kd], self.act["model.%s.aqtizer_w.zero_point" % apath]
        else:
            if apath + ".aqtizer_w" not in self.lazy:
                self.lazy[apath + ".aqtizer_w"] = minmax_scalar(p, cfg.softmax_bits, always_zero=True)
            d, z = self.lazy[apath + ".aqtizer_w"]
        return uaq(p, d, z, cfg.softmax_bits)

    def attention(self, apath, x, ctx, heads, start_peak):
        """Attention.Attention_forward (diffusers_rewrite/sd.py:151-207)."""
        src = x if ctx is None else ctx
        q = self.linear(apath + ".to_q", x)
        k = self.linear(apath + ".to_k", src)
        v = self.linear(apath + ".to_v", src)
        b, t, c = q.shape
        hd = c // heads
        q = q.view(b, q.shape[1], heads, hd).transpose(1, 2)
        k = k.view(b, k.shape[1], heads, hd).transpose(1, 2)
        v = v.view(b, v.shape[1], heads, hd).transpose(1, 2)
        use_aq = self.cfg.use_aq
        if use_aq:
            q = self.aq(apath + ".aqtizer_q", q)
            if start_peak:
                k = torch.cat([k[..., 0:1, :], self.aq(apath + ".aqtizer_k", k[..., 1:, :])], dim=-2)
            else:
                k = self.aq(apath + ".aqtizer_k", k)
        if self.cfg.exact_gemm:
            sc = torch.matmul(q.double(), k.double().transpose(-2, -1)).float()
        else:
            sc = torch.matmul(q, k.transpose(-2, -1))
        p = torch.softmax(sc * (hd ** -0.5), dim=-1)
        del sc
        if use_aq:
            p = p.to(torch.float32)
            if start_peak:
                p = torch.cat([p[..., 0:1], self.softmax_quant(apath, p[..., 1:])], dim=-1)
            else:
                p = self.softmax_quant(apath, p)
            v = self.aq(apath + ".aqtizer_v", v)
        if self.cfg.exact_gemm:
            o = torch.matmul(p.double(), v.double()).float().transpose(1, 2).contiguous().view(b, t, c)
        else:
            o = torch.matmul(p, v).transpose(1, 2).contiguous().view(b, t, c)
        return self.linear(apath + ".to_out.0", o)

    def transformer_block(self, path, x, ctx, heads):
        """QuantBasicTransformerBlock.forward (quant_block.py:165-186); start_peak only on attn2
        (:157-158); GEGLU FF (sd.py:210-236)."""
        x = x + self.attention(path + ".attn1", self.layer_norm(path + ".norm1", x), None, heads, False)
        x = x + self.attention(path + ".attn2", self.layer_norm(path + ".norm2", x), ctx, heads,
                               self.cfg.t2i_start_peak)
        h = self.linear(path + ".ff.net.0.proj", self.layer_norm(path + ".norm3", x))
        a, g = h.chunk(2, dim=-1)
        return x + self.linear(path + ".ff.net.2", a * F.gelu(g))

    def resnet(self, path, x, temb):
        """QuantResnetBlock2D.forward (quant_block.py:98-119)."""
        h = self.conv(path + ".conv1", F.silu(self.group_norm(path + ".norm1", x, 1e-5)), 1, 1)
        h = h + self.linear(path + ".time_emb_proj", F.silu(temb))[:, :, None, None]
        h = self.conv(path + ".conv2", F.silu(self.group_norm(path + ".norm2", h, 1e-5)), 1, 1)
        if self.has(path + ".conv_shortcut.w"):
            x = self.conv(path + ".conv_shortcut", x, 1, 0)
        return x + h

    def transformer2d(self, path, x, ctx, n_layers, heads, proj):
        """Transformer2DModel.forward: SD 1×1-conv projections (sd.py:283-305), SDXL Linear
        projections on the token layout (sdxl.py:306-326); GroupNorm eps 1e-6."""
        b, c, hh, ww = x.shape
        res = x
        h = self.group_norm(path + ".norm", x, 1e-6)
        if proj == "conv":
            h = self.conv(path + ".proj_in", h, 1, 0)
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        if proj != "conv":
            h = self.linear(path + ".proj_in", h)
        for i in range(n_layers):
            h = self.transformer_block("%s.transformer_blocks.%d" % (path, i), h, ctx, heads)
        if proj != "conv":
            h = self.linear(path + ".proj_out", h)
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2).contiguous()
        if proj == "conv":
            h = self.conv(path + ".proj_out", h, 1, 0)
        return h + res

    # -- whole UNet ------------------------------------------------------------------------
    @staticmethod
    def timesteps_embedding(t, dim):
        """Timesteps.forward (sd.py:20-39): cos ‖ sin."""
        half = dim // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / (half - 0.0))
        ang = t[:, None].float() * freqs[None, :]
        return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)

    def time_embed(self, path, x):
        return self.linear(path + ".linear_2", F.silu(self.linear(path + ".linear_1", x)))

    @torch.no_grad()
    def forward(self, sample, t, ctx, text_embeds=None, time_ids=None):
        """UNet2DConditionModel.forward (sd.py:546-620 / sdxl.py:558-631) under QuantModel
        (quant_model.py:113-116) with the time-aware act reload (calibration.py:297-312)."""
        cfg = self.cfg
        if cfg.use_aq and cfg.time_aware:
            self.act = self.ck["act_%d" % slot_for_timestep(t, cfg.num_inference_steps)]
        A = ORACLE_ARCH[cfg.arch]
        xl = cfg.arch == "sdxl"
        tt = torch.as_tensor(t).reshape(-1).expand(sample.shape[0])
        emb = self.time_embed("time_embedding", self.timesteps_embedding(tt, A["bo"][0]))
        if xl:
            te = self.timesteps_embedding(time_ids.flatten(), 256).reshape(text_embeds.shape[0], -1)
            emb = emb + self.time_embed("add_embedding", torch.cat([text_embeds, te], dim=-1))
        bo, down, up, proj, heads = A["bo"], A["down"], A["up"], A["proj"], A["heads"]
        h = self.fp_conv("conv_in", sample)
        skips = [h]
        for i, ((nl, has_down), c) in enumerate(zip(down, bo)):
            for j in range(2):
                h = self.resnet("down_blocks.%d.resnets.%d" % (i, j), h, emb)
                if nl:
                    h = self.transformer2d("down_blocks.%d.attentions.%d" % (i, j), h, ctx, nl, heads(c), proj)
                skips.append(h)
            if has_down:
                h = self.conv("down_blocks.%d.downsamplers.0.conv" % i, h, 2, 1)
                skips.append(h)
        h = self.resnet("mid_block.resnets.0", h, emb)
        h = self.transformer2d("mid_block.attentions.0", h, ctx, A["mid"], heads(bo[-1]), proj)
        h = self.resnet("mid_block.resnets.1", h, emb)
        for i, ((nl, has_up), c) in enumerate(zip(up, reversed(bo))):
            for j in range(3):
                h = torch.cat([h, skips.pop()], dim=1)
                h = self.resnet("up_blocks.%d.resnets.%d" % (i, j), h, emb)
                if nl:
                    h = self.transformer2d("up_blocks.%d.attentions.%d" % (i, j), h, ctx, nl, heads(c), proj)
            if has_up:
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
                h = self.conv("up_blocks.%d.upsamplers.0.conv" % i, h, 1, 1)
        h = F.silu(self.group_norm("conv_norm_out", h, 1e-5))
        return self.fp_conv("conv_out", h)


# ----------------------------------------------------------------------------- DDIM


class DDIM:
    """Deterministic DDIM (η=0) restating diffusers ``scheduling_ddim.py``: leading timestep
    spacing with steps_offset=1 (:325-330) and the ε-prediction update (:404-450);
    scaled_linear β 0.00085→0.012, set_alpha_to_one=False, clip_sample=False — the public SD-v1-4
    scheduler config."""

    def __init__(self, num_inference_steps, num_train=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.n, self.num_train = num_inference_steps, num_train
        ratio = num_train // num_inference_steps
        self.timesteps = [int(i * ratio) + steps_offset for i in range(num_inference_steps)][::-1]

    def step(self, eps, t, sample):
        prev_t = t - self.num_train // self.n
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        x0 = (sample - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps


@torch.no_grad()
def denoise_loop(model_fn, latents, ctx_pair, num_inference_steps, guidance=7.5, extra=None):
    """CFG denoise loop: one UNet call per step on the (uncond ‖ cond) pair
    (pipeline_stable_diffusion.py:1027-1040)."""
    sch = DDIM(num_inference_steps)
    x = latents.clone()
    for t in sch.timesteps:
        inp = torch.cat([x, x], dim=0) if guidance > 0 else x
        eps = model_fn(inp, t, ctx_pair, **(extra or {}))
        if guidance > 0:
            e_u, e_c = eps.chunk(2)
            eps = e_u + guidance * (e_c - e_u)
        x = sch.step(eps, t, x)
    return x
