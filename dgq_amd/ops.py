"""Thin Python wrappers, one per C-ABI entry point of libdgq_hip.so, plus the per-layer objects
(`PackedWeight`, `ActBinding`) that own the device-resident tables.  No arithmetic happens in Python on the
hot path: a quantized layer call is `dgq_quant_act` + `dgq_gemm_wxa8`."""
import ctypes
import os
import threading
from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib
from .plan import ActLayout, natural_kperm, round_up, KTILE, act_offset, mark_clears, flush_coefficients

_c = ctypes


def _lib_call(name, *args):
    _lib.check(getattr(_lib.load(), name)(*args), name)


# ------------------------------------------------------------------------------------------ weights
def quantize_weight(w: torch.Tensor, delta: torch.Tensor, zp: torch.Tensor, alpha: Optional[torch.Tensor], bits: int):
    """Integer codes (uint8 [N][K_ref]) of wqtizer(w); K_ref = w.view(N,-1) order."""
    _lib.require_gpu()
    N = w.shape[0]
    w2 = w.detach().reshape(N, -1).contiguous().float()
    K = w2.shape[1]
    d = delta.detach().reshape(-1).contiguous().float()
    z = zp.detach().reshape(-1).contiguous().float()
    assert d.numel() == N and z.numel() == N, "weight quantizer must be per output channel"
    a = alpha.detach().reshape(N, -1).contiguous().float() if alpha is not None else None
    codes = torch.empty((N, K), dtype=torch.uint8, device=w.device)
    _lib_call("dgq_quantize_weight", _lib.ptr(w2), _lib.ptr(d), _lib.ptr(z), _lib.ptr(a), N, K, bits,
              _lib.ptr(codes), _lib.stream())
    return codes


class _AdaRoundSoft(torch.autograd.Function):
    """ŵ(α) of the AdaRound soft quantiser (adaptive_rounding.py:39-70, soft_tgt) — dgq_adaround_soft_fwd / _bwd.
    Only α receives a gradient: it is the only tensor the reconstruction optimiser owns (reconstruction.py:37-41)."""

    @staticmethod
    def forward(ctx, w2, d, z, alpha2, bits):
        N, K = w2.shape
        out = torch.empty_like(w2)
        _lib_call("dgq_adaround_soft_fwd", _lib.ptr(w2), _lib.ptr(d), _lib.ptr(z), _lib.ptr(alpha2), N, K, bits,
                  _lib.ptr(out), _lib.stream())
        ctx.save_for_backward(w2, d, z, alpha2)
        ctx.bits = bits
        return out

    @staticmethod
    def backward(ctx, gout):
        w2, d, z, alpha2 = ctx.saved_tensors
        N, K = w2.shape
        g = gout.contiguous().float()
        galpha = torch.empty_like(alpha2)
        _lib_call("dgq_adaround_soft_bwd", _lib.ptr(g), _lib.ptr(w2), _lib.ptr(d), _lib.ptr(z), _lib.ptr(alpha2), N, K,
                  ctx.bits, _lib.ptr(galpha), _lib.stream())
        return None, None, None, galpha, None


def adaround_soft(w: torch.Tensor, delta: torch.Tensor, zp: torch.Tensor, alpha: torch.Tensor, bits: int):
    """δ·(clamp(floor(w/δ) + h(α) + z, 0, 2^bits−1) − z) with h the rectified sigmoid; differentiable in α (fp32)."""
    _lib.require_gpu()
    N = w.shape[0]
    w2 = w.detach().reshape(N, -1).contiguous().float()
    d = delta.detach().reshape(-1).contiguous().float()
    z = torch.as_tensor(zp).detach().reshape(-1).float().to(w.device)
    z = (z.expand(N) if z.numel() == 1 else z).contiguous()
    assert d.numel() == N and z.numel() == N, "weight quantizer must be per output channel"
    a2 = alpha.reshape(N, -1)
    if a2.dtype != torch.float32 or not a2.is_contiguous():
        a2 = a2.float().contiguous()
    return _AdaRoundSoft.apply(w2, d, z, a2, bits).view(w.shape)


class _AdaRoundReg(torch.autograd.Function):
    """Σ (1 − |2h(α) − 1|^b) (reconstruction_util.py:68-70) — dgq_adaround_reg_fwd / _bwd."""

    @staticmethod
    def forward(ctx, alpha, b):
        a = alpha.contiguous()
        n = a.numel()
        part = torch.empty((_lib.load().dgq_adaround_reg_blocks(n),), dtype=torch.float32, device=a.device)
        _lib_call("dgq_adaround_reg_fwd", _lib.ptr(a), n, _c.c_float(b), _lib.ptr(part), _lib.stream())
        ctx.save_for_backward(a)
        ctx.b = b
        return part.sum()

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        gs = g.reshape(1).float().contiguous()
        galpha = torch.empty_like(a)
        _lib_call("dgq_adaround_reg_bwd", _lib.ptr(a), a.numel(), _c.c_float(ctx.b), _lib.ptr(gs), _lib.ptr(galpha),
                  _lib.stream())
        return galpha.view_as(a), None


def adaround_reg(alpha: torch.Tensor, b: float):
    """The rounding regulariser Σ (1 − |2h(α) − 1|^b) as a 0-d tensor, differentiable in α."""
    _lib.require_gpu()
    assert alpha.dtype == torch.float32
    return _AdaRoundReg.apply(alpha, float(b))


#: dgq_pack_w4 layout every kernel of this package reads (rows with bit 4 set: 8-byte halves of a 32-chunk exchanged)
W4_LAYOUT = 1


#: dgq_pack_w4 layout of the short-K kernel's weight image (gemm_panel.hip): fragment-major, N padded to 32-column tiles
W4_FRAG_LAYOUT = 2
#: keep a fragment-major copy of every W4 weight beside the row-major one, so that dgq_gemm_wxa8 may take its short-K kernel
#: (DGQ_GEMM_PANEL=0: never — the tile family alone, as in round 4)
GEMM_PANEL = True


def pack_weight(codes: torch.Tensor, kperm: Optional[torch.Tensor], Kp: int, bits: int, layout: int = W4_LAYOUT):
    N, K = codes.shape
    kp = kperm.to(codes.device, torch.int32).contiguous() if kperm is not None else None
    if bits == 4:
        rows = (N + 31) // 32 * 32 if layout == W4_FRAG_LAYOUT else N
        out = torch.empty((rows, Kp // 2), dtype=torch.uint8, device=codes.device)
        _lib_call("dgq_pack_w4", _lib.ptr(codes), N, K, _lib.ptr(kp), Kp, layout, _lib.ptr(out), _lib.stream())
    elif bits == 8:
        out = torch.empty((N, Kp), dtype=torch.int8, device=codes.device)
        _lib_call("dgq_pack_w8", _lib.ptr(codes), N, K, _lib.ptr(kp), Kp, _lib.ptr(out), _lib.stream())
    else:
        raise NotImplementedError("weight bits %d: the HIP path implements W4 and W8" % bits)
    return out


def unpack_w4(packed: torch.Tensor, Kp: int, layout: int = W4_LAYOUT):
    N = packed.shape[0]
    out = torch.empty((N, Kp), dtype=torch.uint8, device=packed.device)
    _lib_call("dgq_unpack_w4", _lib.ptr(packed), N, Kp, layout, _lib.ptr(out), _lib.stream())
    return out


class PackedWeight:
    """One quantized layer's weight, frozen to integer codes once (the reference re-derives them from fp32
    ``w`` on every call, quant_layer.py:642-643)."""

    def __init__(self, w, delta, zp, alpha, bias, bits, C, taps):
        self.bits, self.C, self.taps = bits, C, taps
        self.N = w.shape[0]
        self.K = C * taps
        dev = w.device
        self.codes = quantize_weight(w, delta, zp, alpha, bits)           # [N][K_ref] u8
        woff = 0.0 if bits == 4 else 128.0
        self.alpha = delta.detach().reshape(-1).float().contiguous().to(dev)
        self.zp_true = zp.detach().reshape(-1).float().contiguous().to(dev)
        self.zw = (self.zp_true - woff).contiguous()                       # zero point in the stored code domain
        self.bias = (bias.detach().float().contiguous().to(dev) if bias is not None
                     else torch.zeros(self.N, device=dev))
        self._natural = None
        self._natural_frag = None
        self._vn = None

    def natural_frag(self):
        """the natural-order weights fragment-major (dgq_pack_w4 layout 2), W4 only; None otherwise"""
        if self._natural_frag is None and self.bits == 4 and GEMM_PANEL:
            perm = natural_kperm(self.C, self.taps)
            self._natural_frag = pack_weight(self.codes, perm, perm.numel(), 4, W4_FRAG_LAYOUT)
        return self._natural_frag

    def natural(self):
        if self._natural is None:
            perm = natural_kperm(self.C, self.taps)
            self._natural = (pack_weight(self.codes, perm, perm.numel(), self.bits), perm.numel())
        return self._natural

    def centered(self):
        """(q − z) as float64 [N][K_ref] — load-time helper for the epilogue constants."""
        return self.codes.double() - self.zp_true.double()[:, None]

    def vn(self):
        if self._vn is None:
            self._vn = self.centered().sum(dim=1).float().contiguous()
        return self._vn


class ActBinding:
    """Device-resident tables of one (layer, timestep-slot) activation quantizer."""

    def koff(self, W, ldc):
        """ksrc resolved to element offsets (dh*W + dw)*ldc + c for one input geometry (cached)."""
        if self.ksrc is None:
            return None
        key = (W, ldc)
        if key not in self._koff:
            e = self.ksrc
            off = (((e >> 24) & 0x7F) * W + ((e >> 16) & 0xFF)) * ldc + (e & 0xFFFF)
            self._koff[key] = torch.where(e >= 0, off, torch.full_like(off, -1)).to(torch.int32).contiguous()
        return self._koff[key]

    def klds(self, kw, C):
        """ksrc resolved to indices (dh*kw + dw)*C + c into the [tap][C] strip a wave stages in LDS (cached)."""
        if self.ksrc is None:
            return None
        key = ("lds", kw, C)
        if key not in self._koff:
            e = self.ksrc
            idx = (((e >> 24) & 0x7F) * kw + ((e >> 16) & 0xFF)) * C + (e & 0xFFFF)
            self._koff[key] = torch.where(e >= 0, idx, torch.full_like(idx, -1)).to(torch.int32).contiguous()
        return self._koff[key]

    def kdst(self, kw, C, taps):
        """inverse of ksrc: packed position kp of element (tap, c), [taps*C] int32 (cached) — dgq_quant_act's scatter path"""
        if self.ksrc is None:
            return None
        key = ("dst", kw, C, taps)
        if key not in self._koff:
            e = self.ksrc
            valid = e >= 0
            idx = ((((e >> 24) & 0x7F) * kw + ((e >> 16) & 0xFF)) * C + (e & 0xFFFF))[valid].long()
            out = torch.full((taps * C,), -1, dtype=torch.int32, device=e.device)
            out[idx] = torch.nonzero(valid).flatten().to(torch.int32)
            assert int((out < 0).sum()) == 0, "per-K table does not cover every (tap, channel)"
            self._koff[key] = out.contiguous()
        return self._koff[key]

    def kpat(self, kh, kw, C, stride):
        """(dh·PW + dw)·C + c of every packed position inside the input patch of a conv tile (dgq_quant_act's block-staged
        path; PW from dgq_quant_act_conv_tile), -1 for padding; None when the geometry has no such path.  Cached."""
        key = ("pat", kh, kw, C, stride)
        if key not in self._koff:
            pw_ = _c.c_int(0)
            tile = _lib.load().dgq_quant_act_conv_tile(C, kh, kw, stride, self.Kp, _c.byref(pw_))
            if tile == 0:
                self._koff[key] = None
            elif self.ksrc is not None:
                e = self.ksrc
                idx = (((e >> 24) & 0x7F) * pw_.value + ((e >> 16) & 0xFF)) * C + (e & 0xFFFF)
                self._koff[key] = torch.where(e >= 0, idx, torch.full_like(idx, -1)).to(torch.int32).contiguous()
            else:                                                   # natural order kp = tap·C + c, padded to Kp
                kp = torch.arange(self.Kp, device=self.pw.codes.device)
                tap, c = kp // C, kp % C
                idx = ((tap // kw) * pw_.value + (tap % kw)) * C + c
                self._koff[key] = torch.where(kp < kh * kw * C, idx, torch.full_like(idx, -1)).to(torch.int32).contiguous()
        return self._koff[key]

    def kpat_for(self, kh, kw, C, pw):
        """``kpat`` for a patch PW = pw pixels wide (dgq_gemm_act_t.kpat of the conv form of quantise-on-load, csrc/gemm_convq.hip:
        4 x 8 output positions per workgroup, pw = 8 + kw − 1).  Cached."""
        key = ("patw", kh, kw, C, pw)
        if key not in self._koff:
            if self.ksrc is not None:
                e = self.ksrc
                idx = (((e >> 24) & 0x7F) * pw + ((e >> 16) & 0xFF)) * C + (e & 0xFFFF)
                self._koff[key] = torch.where(e >= 0, idx, torch.full_like(idx, -1)).to(torch.int32).contiguous()
            else:                                                   # natural order kp = tap·C + c, padded to Kp
                kp = torch.arange(self.Kp, device=self.pw.codes.device)
                tap, c = kp // C, kp % C
                idx = ((tap // kw) * pw + (tap % kw)) * C + c
                self._koff[key] = torch.where(kp < kh * kw * C, idx, torch.full_like(idx, -1)).to(torch.int32).contiguous()
        return self._koff[key]

    def input_binding(self, C):
        """This (scalar) quantizer applied to the conv's INPUT tensor as a 1x1 layer in natural order: what dgq_quant_act needs to
        write the int8 NHWC code tensor of the implicit-im2col path (cached)."""
        hit = self._koff.get(("input", C))
        if hit is None:
            import copy
            hit = copy.copy(self)
            hit.Kp = round_up(C, KTILE)
            hit._koff = {}
            self._koff[("input", C)] = hit
        return hit

    def conv_zero_code(self):
        """centred int8 code of the value 0.0 under this scalar quantizer, z − offset (set at construction); None when z is not a
        code of the b-bit range (no implicit operand for such a layer)"""
        return self._zc

    def conv_fill(self):
        """32 bytes: 16 x the code of 0.0 (a tap outside the image), 16 x 0 (the K padding) — dgq_gemm_conv_t.fill"""
        return self._fill

    def __init__(self, layout: ActLayout, pw: PackedWeight, abits: int):
        dev = pw.codes.device
        self.mode, self.abits, self.offset = layout.mode, abits, act_offset(abits)
        self.pw = pw
        if layout.mode == "perK":
            self.Kp = layout.Kp
            self.ksrc = layout.ksrc.to(dev)
            self._koff = {}
            self.cdelta = layout.cdelta.to(dev)
            self.czp = layout.czp.to(dev)
            cfl = mark_clears(layout.cflush, abits, pw.bits)                     # + where the GEMM clears its running total
            self.cflush = cfl.to(dev)
            self.ccoef = flush_coefficients(layout.cdelta, cfl).to(dev)          # the same information as scalar-loadable coefficients
            self.wpacked = pack_weight(pw.codes, layout.kperm, layout.Kp, pw.bits)
            self.wfrag = pack_weight(pw.codes, layout.kperm, layout.Kp, 4, W4_FRAG_LAYOUT) if (pw.bits == 4 and GEMM_PANEL) else None
            U = (pw.centered() @ layout.kcoef.to(dev)).float()             # Σ_k δ_k(o − z_k)(qw − zw)
            self.gamma = (pw.bias + pw.alpha * U).contiguous()
            self.n_groups = layout.n_groups
        else:
            self.wpacked, self.Kp = pw.natural()
            self.wfrag = pw.natural_frag()
            self.ksrc = None
            self._koff = {}
            self.mdelta = layout.mdelta.to(dev).contiguous()
            self.mzp = layout.mzp.to(dev).contiguous()
            self.L = layout.L
            self.gamma = pw.bias
            self.vn = pw.vn()
            if layout.mode == "scalar":                    # implicit-im2col convolutions: the code of 0.0 and the DMA fill line
                # The reference's scalar-δ convolution is the native F.conv2d(aqtizer(x), ŵ, padding) (quant_layer.py:659): it pads with
                # exact 0.0 BEHIND the quantizer, i.e. with the code z.  That is a code of the b-bit range only while 0 <= z <= 2^b − 1
                # (always, for the min/max-derived scales of every DGQ recipe; a Scaler.MSE range of a one-signed input could leave it).
                # Outside it neither this path nor the materialised one can represent the padding value: no implicit operand then
                # (``conv_zero_code() is None``), the layer takes the materialising pass and its documented pad-then-quantise form.
                z = float(layout.mzp.reshape(-1)[0])
                if 0.0 <= round(z) <= 2.0 ** abits - 1.0:
                    self._zc = float(round(z) - self.offset)
                    self._fill = torch.tensor([int(self._zc)] * 16 + [0] * 16, dtype=torch.int8, device=dev)
                else:
                    self._zc, self._fill = None, None


# ------------------------------------------------------------------------------------------ hot path
def act_ksplits(M, Kp):
    """K splits of the activation pre-pass: aim for >= 8192 waves (one per row x split) on low-M layers."""
    ks = max(1, min(16, 8192 // max(M, 1), Kp // 512))
    return _lib.load().dgq_quant_act_parts(Kp, ks)


FLOAT_DTYPES = (torch.float32, torch.float16, torch.bfloat16)
def as_f32(t: torch.Tensor) -> torch.Tensor:
    """fp32 copy of a small parameter vector (norm γ/β ...) of a model that was cast with .half(): the kernels take
    their per-channel vectors in fp32.  The copy is cached ON the tensor object (pass the Parameter, not ``.data``) and
    refreshed when the tensor is modified in place or moved; nothing is keyed by address, so a later model that
    happens to reuse the same device memory can never see a stale vector."""
    if t.dtype == torch.float32:
        return t.detach()
    key = (t._version, t.data_ptr(), t.dtype)
    hit = t.__dict__.get("_dgq_f32")
    if hit is None or hit[0] != key:
        hit = (key, t.detach().float().contiguous())
        t.__dict__["_dgq_f32"] = hit
    return hit[1]


def groupnorm_scale_shift(x_cl: torch.Tensor, B, HW, C, groups, eps, gamma, beta):
    """GN(x) = x*scale + shift with scale/shift [B][C] (see dgq_groupnorm_scale_shift)."""
    dev = x_cl.device
    # enough (batch, group, slice) blocks to fill the chip; one block per (batch, group) — a single launch — only at 8x8
    # (measured: one block per group at 64x64 costs +1.1 ms per step)
    slices = max(1, min(32, (2048 + B * groups - 1) // (B * groups), HW // 64))
    scale = torch.empty((B, C), dtype=torch.float32, device=dev)
    shift = torch.empty((B, C), dtype=torch.float32, device=dev)
    part = torch.empty((B * groups * slices * 3,), dtype=torch.float32, device=dev)
    _lib_call("dgq_groupnorm_scale_shift", _lib.ptr(x_cl), _lib.DTYPE_CODE[x_cl.dtype], B, HW, C, groups,
              _c.c_float(eps), _lib.ptr(as_f32(gamma)), _lib.ptr(as_f32(beta)), _lib.ptr(scale), _lib.ptr(shift),
              _lib.ptr(part), slices, _lib.stream())
    return scale, shift


#: GroupNorm statistics out of the producing GEMM's epilogue (dgq_gemm_extra_t.gn_partial + dgq_groupnorm_from_partials) instead
#: of a pass over the tensor.  A conv output carries its partials as a tensor attribute (``_dgq_gn``); a GroupNorm folded
#: into the next layer's load uses them when the very same tensor object (or a channel concat of two such tensors,
#: ``cat_channels``) reaches it.  DGQ_GN_FROM_GEMM=0 restores the standalone statistics kernels.
GN_FROM_GEMM = True
#: ... also where the producing GEMM is K-split: its combine kernel writes the partials (=0: statistics pass for those tensors)
GN_FROM_SPLITK = True


#: convolutions whose activation quantizer is one (δ, z) pair (config C5 / C2U; the reference's native conv path) take the
#: implicit-im2col GEMM (dgq_gemm_conv_t): the input is quantised once per pixel and the unfolded operand never exists.
#: DGQ_CONV_IMPLICIT=0: the materialising pass of every other conv layer.
CONV_IMPLICIT = True


class OutputRedirect:
    """Where the NEXT "final" layer call of a module may put its output: ``out`` — an [M][N] row view it stores INTO (its result tensor is
    then a view of that memory), ``out2`` — one it stores into AS WELL.  The UNet sets it (``ops.set_redirect``) around the call of a module whose
    output is one half of an up-path concatenation (diffusers_rewrite/sd.py:558-613), both halves being row slices [:, :C1] / [:, C1:] of
    one [M][C1 + C2] buffer; a layer call made with ``final=True`` (the call whose result IS the module's result, residual included)
    takes it.  Nobody taking it is fine: the caller then finds its tensors elsewhere and concatenates as before."""

    def __init__(self, out=None, out2=None):
        self.out, self.out2, self.taken = out, out2, False


_REDIRECT_TLS = threading.local()                      # per thread: two threads running a model each never see the other's redirect


def set_redirect(rd):
    _REDIRECT_TLS.rd = rd


def pending_redirect():
    return getattr(_REDIRECT_TLS, "rd", None)


#: DGQ_CAT_INPLACE=0 (A/B runs): torch.cat for every skip concatenation
CAT_INPLACE = os.environ.get("DGQ_CAT_INPLACE", "1") != "0"


def take_redirect(M, N, dtype):
    """the pending OutputRedirect's (out, out2) if its views have this layer's output shape and dtype, else (None, None)"""
    rd = pending_redirect()
    if rd is None or rd.taken:
        return None, None
    for v in (rd.out, rd.out2):
        if v is not None and (tuple(v.shape) != (M, N) or v.dtype != dtype or v.stride(1) != 1):
            return None, None
    rd.taken = True
    return rd.out, rd.out2


def cat_channels(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat([a, b], dim=1) of two NCHW tensors that keeps the GroupNorm partials of both sources (the skip concatenation in
    front of an up block's norm1): dgq_groupnorm_from_partials takes the two partial buffers as one channel range."""
    y = torch.cat([a, b], dim=1)
    ga, gb = _gn_of(a), _gn_of(b)
    if ga is not None and gb is not None and len(ga["parts"]) == 1 and len(gb["parts"]) == 1 and ga["B"] == gb["B"] and ga["HW"] == gb["HW"]:
        y._dgq_gn = dict(parts=ga["parts"] + gb["parts"], B=ga["B"], HW=ga["HW"], C=ga["C"] + gb["C"], ver=y._version)
    return y


def _gn_of(x: torch.Tensor):
    """The GroupNorm partials a producing GEMM attached to x — only while x is still the tensor it wrote: an in-place op in
    between (``x += ...``, a hook that mutates and returns its output) bumps ``_version`` and the statistics are stale."""
    gn = getattr(x, "_dgq_gn", None) if GN_FROM_GEMM else None
    return gn if (gn is not None and gn.get("ver") == x._version) else None


def groupnorm_from_partials(gn, groups, eps, gamma, beta):
    """scale / shift [B][C] of GN(x) from the partial statistics the producer(s) of x wrote (see dgq_groupnorm_from_partials)"""
    parts = gn["parts"]
    B, HW, C = gn["B"], gn["HW"], gn["C"]
    dev = parts[0][0].device
    scale = torch.empty((B, C), dtype=torch.float32, device=dev)
    shift = torch.empty((B, C), dtype=torch.float32, device=dev)
    p2, c2 = (parts[1][0], parts[1][1]) if len(parts) > 1 else (None, 0)
    _lib_call("dgq_groupnorm_from_partials", _lib.ptr(parts[0][0]), parts[0][1], _lib.ptr(p2), c2, B, HW, groups, _c.c_float(eps),
              _lib.ptr(as_f32(gamma)), _lib.ptr(as_f32(beta)), _lib.ptr(scale), _lib.ptr(shift), _lib.stream())
    return scale, shift


def quant_act(x_cl: torch.Tensor, B, H, W, C, kh, kw, stride, pad, ab: ActBinding, pre=None, ln=None, ups=False):
    """x_cl: contiguous channels-last storage [B][H][W][C] (any fp dtype). Returns (codes, rowsum[parts][M], M).
    pre = (scale [B][C], shift [B][C], act) folds a GroupNorm (+SiLU when act == 1) into the load;
    ln = (gamma [C], beta [C], eps) folds a LayerNorm over each row's C elements (Linear inputs).
    ups: x_cl holds [B][H/2][W/2][C] and is read through a 2x nearest upsample (Upsample2D's interpolate folded into the load);
    returns None when the layer's quantise variant has no such form."""
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    M = B * Ho * Wo
    per_m = 0 if ab.mode == "perK" else 1
    ldc = 2 * C if (pre and pre[2] == 2) else C
    a = _lib.QuantActArgs()
    a.x, a.x_dtype, a.B, a.H, a.W, a.C, a.kh, a.kw, a.stride, a.pad = x_cl.data_ptr(), _lib.DTYPE_CODE[x_cl.dtype], B, H, W, C, kh, kw, stride, pad
    a.ksrc, a.koff, a.klds = _dp(ab.ksrc), _dp(ab.koff(W, ldc)), _dp(ab.klds(kw, C))
    a.kdst = _dp(ab.kdst(kw, C, kh * kw)) if ab.ksrc is not None else None
    a.kpat = _dp(ab.kpat(kh, kw, C, stride)) if (kh * kw > 1 and C % 4 == 0) else None
    a.Kp, a.per_m = ab.Kp, per_m
    a.delta, a.zp = (ab.cdelta.data_ptr(), ab.czp.data_ptr()) if not per_m else (ab.mdelta.data_ptr(), ab.mzp.data_ptr())
    a.L, a.bits = (1 if not per_m else ab.L), ab.abits
    a.pre_scale = _dp(pre[0]) if pre and pre[0] is not None else None
    a.pre_shift = _dp(pre[1]) if pre and pre[1] is not None else None
    a.pre_act = pre[2] if pre else 0
    lnp = (as_f32(ln[0]), as_f32(ln[1]), float(ln[2])) if ln else None
    a.ln_gamma, a.ln_beta, a.ln_eps = (lnp[0].data_ptr(), lnp[1].data_ptr(), lnp[2]) if lnp else (None, None, 0.0)
    a.ups = 1 if ups else 0
    # the LDS-scatter path (per-K convs, per-K Linear inputs with short groups) takes the whole row in one wave / block:
    # ask for it with one K split first
    parts = 1
    a.ksplits = 1
    a.codes = a.rowsum = 1                                   # placeholders: dgq_quant_act_variant only validates non-NULL
    if (a.kdst is None and a.kpat is None) or _lib.load().dgq_quant_act_variant(_c.byref(a)) not in (3, 4, 5):
        if ups:
            return None                                      # (the folded upsample exists on the scatter / block-staged paths)
        parts = act_ksplits(M, ab.Kp)
        a.ksplits = parts
    codes = torch.empty((M, ab.Kp), dtype=torch.int8, device=x_cl.device)
    rowsum = torch.empty((parts, M), dtype=torch.float32, device=x_cl.device)
    a.codes, a.rowsum = codes.data_ptr(), rowsum.data_ptr()

    def issue():
        _lib_call("dgq_quant_act_batch", 1, _c.byref(a), _lib.stream())
    issue()
    if QUANT_LAUNCH_HOOK is not None:
        QUANT_LAUNCH_HOOK(issue, B * H * W * ldc * x_cl.element_size() >> (2 if ups else 0), M * ab.Kp + 4 * parts * M)
    return codes, rowsum, M


def _dp(t):
    return t.data_ptr() if t is not None else None


_WORKSPACE = {}
WORKSPACE_BYTES = 128 << 20
#: slabs kept for user streams (other than the default / capture stream): least recently
#: used first out, so a program that keeps creating streams does not leak 128 MB per handle
_STREAM_SLABS_MAX = 4


def workspace(device):
    """Persistent split-K scratch (caller-owned; one per device AND stream, so that layers running concurrently on
    different streams never share it): the library never allocates."""
    cur = torch.cuda.current_stream(device).cuda_stream
    # the default stream and torch's graph-capture stream of a device are one serial chain (a capture runs while the
    # default stream is idle) and share the "main" slab; any OTHER user stream gets a slab of its own, keyed by handle,
    # so two user streams can never race on one split-K workspace
    is_main = cur == torch.cuda.default_stream(device).cuda_stream or torch.cuda.is_current_stream_capturing()
    branch = "main" if is_main else ("stream", cur)
    key = (str(device), branch)
    if key not in _WORKSPACE:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("dgq_amd: the split-K workspace must exist before graph capture (run the model once eagerly)")
        if isinstance(branch, tuple):                       # a user stream: bounded, least recently used first out
            old = [k for k in _WORKSPACE if k[0] == str(device) and isinstance(k[1], tuple)]
            while len(old) >= _STREAM_SLABS_MAX:
                _WORKSPACE.pop(old.pop(0))
        _WORKSPACE[key] = torch.empty(WORKSPACE_BYTES, dtype=torch.uint8, device=device)
    elif isinstance(branch, tuple):
        _WORKSPACE[key] = _WORKSPACE.pop(key)               # most recently used last
    return _WORKSPACE[key]


def make_extra(residual=None, fq=None, res_div=1, geglu=False, gn_partial=None, conv=None, y2=None):
    """dgq_gemm_extra_t: residual [M / res_div][N] fp32 (row stride = its stride(0); res_div > 1 broadcasts each row
    over res_div consecutive output rows); fq = (mode, delta, zp, T, D, skip, bits) with mode 1 scalar / 2 per token /
    3 per head-dim; geglu = the pair epilogue of a row-interleaved ff.net.0.  Keeps the tensors alive on the returned object."""
    if residual is None and fq is None and not geglu and gn_partial is None and conv is None and y2 is None:
        return None
    ex = _lib.GemmExtra()
    ex.res_div = 1
    if y2 is not None:                                      # a second copy of the output rows ([M][N] view, its own row pitch)
        assert y2.dim() == 2 and y2.stride(1) == 1
        ex.y2, ex.ldy2 = y2.data_ptr(), y2.stride(0)
        ex._y2_keep = y2
    ex.conv = None
    ex.flush_coef = None
    if conv is not None:                                    # (dgq_gemm_conv_t, tensors it points to)
        ex.conv = _c.cast(_c.pointer(conv[0]), _c.c_void_p)
        ex._conv_keep = conv
    ex.geglu = 1 if geglu else 0
    ex.gn_partial = gn_partial.data_ptr() if gn_partial is not None else None
    keep = []
    if residual is not None:
        assert residual.dtype in _lib.DTYPE_CODE and residual.stride(-1) == 1
        ex.residual, ex.ldr, ex.res_div = residual.data_ptr(), residual.stride(0), res_div
        ex.res_dtype = _lib.DTYPE_CODE[residual.dtype]
        keep.append(residual)
    if gn_partial is not None:
        keep.append(gn_partial)
    if fq is not None:
        mode, delta, zp, T, D, skip, bits = fq
        ex.fq_mode, ex.fq_delta, ex.fq_zp = mode, delta.data_ptr(), zp.data_ptr()
        ex.fq_T, ex.fq_D, ex.fq_skip, ex.fq_qmax = T, D, skip, float(2 ** bits - 1)
        keep += [delta, zp]
    else:
        ex.fq_T = ex.fq_D = 1
    ex._keep = keep
    return ex


#: measurement hooks (bench.py's roofline leg), None in production.  QUANT_LAUNCH_HOOK(issue, algorithmic_bytes, overhead_bytes): every
#: dgq_quant_act_batch launch — algorithmic = the layer input read once, un-unfolded; overhead = the int8 code matrix
#: and the row sums it writes, which the algorithm does not need; ATTN_LAUNCH_HOOK(issue, flops, bytes): every dgq_attention call (its two or three kernels).
QUANT_LAUNCH_HOOK = None
ATTN_LAUNCH_HOOK = None
#: when set, every GEMM launch of the dgq_gemm_wxa8 family is issued through
#: ``GEMM_LAUNCH_HOOK(issue, problems)`` — ``issue()`` launches it (again) on the current stream, ``problems`` lists the
#: (M, ActBinding, out_element_size, input_bytes) of the layers the launch computes — input_bytes > 0 where the launch quantises its own
#: operand (the fp input it reads), 0 where it reads a code matrix a dgq_quant_act launch wrote.  None in production.
GEMM_LAUNCH_HOOK = None


def with_layer_tables(extra, ab: "ActBinding", M):
    """extra (or a blank one) carrying the layer's optional operand images: the flush-coefficient table for the shapes the 256-row
    kernel may take, the fragment-major weights for the short-K kernel"""
    need_coef = ab.mode == "perK" and M >= 2048 and ab.pw.N >= 128
    wfrag = getattr(ab, "wfrag", None)
    if not need_coef and wfrag is None:
        return extra
    if extra is None:
        extra = _lib.GemmExtra()
        extra.res_div, extra.fq_T, extra.fq_D, extra._keep = 1, 1, 1, []
    if need_coef:
        extra.flush_coef = ab.ccoef.data_ptr()
    if wfrag is not None:
        extra.wfrag = wfrag.data_ptr()
    return extra


#: Linear / 1x1 layers whose whole padded K fits the short-K kernel's LDS panel run as ONE launch: the GEMM quantises its own rows
#: (dgq_gemm_act_t) instead of reading the codes a dgq_quant_act launch wrote.  DGQ_GEMM_FUSE=0: the two-launch form everywhere.
GEMM_FUSE = True
#: ... also for the 3x3 convolutions whose input patch fits the LDS (csrc/gemm_convq.hip); False: quantise launch + GEMM launch
CONV_FUSE = os.environ.get("DGQ_CONV_FUSE", "1") != "0"         # (this round's A/B switch: DGQ_CONV_FUSE=0)


def _act_operand_ok(x2):
    """what dgq_gemm_wxa8 requires of a quantise-on-load operand (fill_gemm): 16-byte aligned base, row pitch a multiple of 8 bytes"""
    return x2 is None or (x2.data_ptr() % 16 == 0 and (x2.stride(0) * x2.element_size()) % 8 == 0 and x2.stride(-1) == 1)


def act_fuses(ab: "ActBinding", M, K, dtype, n_problems=1, x2=None, N=None, Kp=None):
    """True where quant_linear / quant_conv2d (1x1) may hand the layer to dgq_gemm_wxa8 with quantise-on-load.  x2: the [M][K] operand
    view the launch would read — a view the kernel cannot address (an unaligned slice) keeps the layer on the two-launch form instead
    of failing the call.  N / Kp: what the library plans a shared launch with (problem 0's N, the widest Kp of the launch)."""
    if not (GEMM_FUSE and GEMM_PANEL) or getattr(ab, "wfrag", None) is None or ab.pw.taps != 1 or K % 4 != 0 or dtype not in _lib.DTYPE_CODE:
        return False
    if not _act_operand_ok(x2):
        return False
    dt = _lib.DTYPE_CODE[dtype]
    return bool(_lib.load().dgq_gemm_act_fuses(M, ab.pw.N if N is None else N, K, ab.Kp if Kp is None else Kp, ab.pw.bits,
                                               0 if ab.mode == "perK" else 1, n_problems, dt, dt))


def _multi_fuses(bindings, M, Kin, dtype, x2):
    """quant_linear_multi: every shared launch of _quant_linear_multi_fused (one per scale mode and 8 layers) must take quantise-on-load
    with the shape the library plans it by — problem 0's N and the launch's widest Kp (dgq_gemm_wxa8_batch)"""
    groups = {}
    for i, ab in enumerate(bindings):
        if ab.pw.K != Kin:
            return False
        groups.setdefault(0 if ab.mode == "perK" else 1, []).append(ab)
    for abs_ in groups.values():
        for j0 in range(0, len(abs_), 8):
            chunk = abs_[j0:j0 + 8]
            kp = max(ab.Kp for ab in chunk)
            if not all(act_fuses(ab, M, Kin, dtype, len(chunk), x2, N=chunk[0].pw.N, Kp=kp) for ab in chunk):
                return False
    return True


def conv_act_fuses(ab: "ActBinding", B, H, W, C, kh, kw, stride, pad, dtype, x_store=None):
    """True where quant_conv2d may hand a k x k convolution to dgq_gemm_wxa8 with its activation quantiser inside the launch
    (csrc/gemm_convq.hip: no int8 code matrix in HBM)."""
    if not (GEMM_FUSE and CONV_FUSE) or getattr(ab, "wfrag", None) is None or ab.mode == "scalar" or dtype not in _lib.DTYPE_CODE:
        return False
    if x_store is not None and not (x_store.is_contiguous() and x_store.data_ptr() % 16 == 0 and (C * x_store.element_size()) % 8 == 0):
        return False
    dt = _lib.DTYPE_CODE[dtype]
    return bool(_lib.load().dgq_gemm_conv_act_fuses(B, H, W, C, kh, kw, stride, pad, ab.pw.N, ab.Kp, ab.pw.bits,
                                                    0 if ab.mode == "perK" else 1, dt, dt))


def make_act(x2: torch.Tensor, ab: "ActBinding", pre=None, ln=None, rows_per_image=1):
    """dgq_gemm_act_t for the rows x2 [M][K] (row stride = stride(0)) under the layer's activation quantizer; pre = (scale, shift,
    act) a folded GroupNorm (+ SiLU), ln = (gamma, beta, eps) a folded LayerNorm.  Keeps its tensors alive on the struct."""
    K = ab.pw.K
    a = _lib.GemmAct()
    a.x, a.x_dtype, a.ldx, a.K = x2.data_ptr(), _lib.DTYPE_CODE[x2.dtype], x2.stride(0), K
    keep = [x2]
    if ab.mode == "perK":
        kd = ab.kdst(1, K, 1)
        a.kdst, a.czp = kd.data_ptr(), ab.czp.data_ptr()
        keep.append(kd)
    a.bits = ab.abits
    a.rows_per_image, a.pre_act = rows_per_image, 0
    if pre is not None:
        if pre[0] is not None:
            a.pre_scale, a.pre_shift = pre[0].data_ptr(), pre[1].data_ptr()
            keep += [pre[0], pre[1]]
        a.pre_act = pre[2]
    if ln is not None:
        g, b = as_f32(ln[0]), as_f32(ln[1])
        a.ln_gamma, a.ln_beta, a.ln_eps = g.data_ptr(), b.data_ptr(), float(ln[2])
        keep += [g, b]
    a._keep = keep
    return a


def gemm_act(x2, M, ab: "ActBinding", out_dtype, extra=None, pre=None, ln=None, rows_per_image=1, out=None):
    """one launch: aqtizer(x2) @ Wᵀ with the layer's epilogue — dgq_gemm_wxa8 with quantise-on-load (see act_fuses)"""
    act = make_act(x2, ab, pre, ln, rows_per_image)
    extra = with_layer_tables(extra, ab, M)
    extra.act = _c.cast(_c.pointer(act), _c.c_void_p)
    extra._act_keep = act
    dummy = ab.wfrag                                          # codes / rowsum are ignored under quantise-on-load: any device pointer
    return gemm_wxa8(dummy, dummy, M, ab, out_dtype, out=out, extra=extra, _fused_bytes=x2.element_size() * M * ab.pw.K)


def gemm_wxa8(codes, rowsum, M, ab: ActBinding, out_dtype, out: Optional[torch.Tensor] = None, extra=None, _fused_bytes=0):
    pw = ab.pw
    ws = workspace(codes.device)
    extra = with_layer_tables(extra, ab, M)
    if out is None:
        out = torch.empty((M, pw.N // 2 if (extra is not None and extra.geglu) else pw.N), dtype=out_dtype, device=codes.device)
    per_m = 0 if ab.mode == "perK" else 1
    parts = rowsum.shape[0] if (rowsum.dim() == 2 and not _fused_bytes) else 1
    def issue():
        _lib_call("dgq_gemm_wxa8", _lib.ptr(codes), _lib.ptr(rowsum), parts, M, ab.Kp, _lib.ptr(ab.wpacked), pw.bits, pw.N,
                  per_m,
                  _lib.ptr(ab.cdelta) if not per_m else None, _lib.ptr(ab.cflush) if not per_m else None,
                  _lib.ptr(ab.mdelta) if per_m else None, _lib.ptr(ab.mzp) if per_m else None,
                  ab.L if per_m else 1, _c.c_float(ab.offset),
                  _lib.ptr(pw.alpha), _lib.ptr(pw.zw), _lib.ptr(ab.gamma), _lib.ptr(ab.vn) if per_m else None,
                  _lib.ptr(out), _lib.DTYPE_CODE[out.dtype], out.stride(0), _lib.ptr(ws), ws.numel(),
                  _c.byref(extra) if extra is not None else None, _lib.stream())
    issue()
    if GEMM_LAUNCH_HOOK is not None:
        GEMM_LAUNCH_HOOK(issue, [(M, ab, out.element_size(), _fused_bytes)])
    return out


def quant_linear(x: torch.Tensor, ab: ActBinding, pre_act=0, residual=None, fq=None, ln=None, geglu=False):
    """x [..., K] -> [..., N].  pre_act: 0 none, 1 SiLU(x), 2 GEGLU (x is [..., 2K]: x[:K]·gelu(x[K:])) folded into the
    quantise-on-load pass; residual [..., N] and fq (see make_extra) folded into the GEMM epilogue.  geglu: the weight rows
    are (value, gate) interleaved and the epilogue writes value·gelu(gate), [..., N/2]."""
    Kin = x.shape[-1]
    K = Kin // 2 if pre_act == 2 else Kin
    x2 = x.reshape(-1, Kin)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    rows = x2.shape[0]
    pre = (None, None, pre_act) if pre_act else None
    res2 = None
    if residual is not None:
        res2 = residual.reshape(-1, ab.pw.N)
        if not res2.is_contiguous():
            res2 = res2.contiguous()
    if pre_act != 2 and act_fuses(ab, rows, K, x.dtype, x2=x2):
        y = gemm_act(x2, rows, ab, x.dtype, extra=make_extra(res2, fq, geglu=geglu), pre=pre, ln=ln)
        return y.view(*x.shape[:-1], y.shape[-1])
    codes, rowsum, M = quant_act(x2, rows, 1, 1, K, 1, 1, 1, 0, ab, pre, ln)
    y = gemm_wxa8(codes, rowsum, M, ab, x.dtype, extra=make_extra(res2, fq, geglu=geglu))
    return y.view(*x.shape[:-1], y.shape[-1])


def _quant_linear_multi_fused(x, x2, M, bindings, ln):
    """quant_linear_multi as dgq_gemm_wxa8_batch launches with quantise-on-load: one launch per scale mode and 8 layers"""
    dev = x2.device
    outs = [torch.empty((M, ab.pw.N), dtype=x.dtype, device=dev) for ab in bindings]
    groups = {}
    for i, ab in enumerate(bindings):
        groups.setdefault(0 if ab.mode == "perK" else 1, []).append(i)
    for per_m, idxs in groups.items():
        for j0 in range(0, len(idxs), 8):
            chunk = idxs[j0:j0 + 8]
            arr = (_lib.GemmArgs * len(chunk))()
            extras = []
            for g, i in zip(arr, chunk):
                ab, pw = bindings[i], bindings[i].pw
                act = make_act(x2, ab, None, ln)
                ex = with_layer_tables(None, ab, 0)
                ex.act = _c.cast(_c.pointer(act), _c.c_void_p)
                extras.append((ex, act))
                g.codes, g.rowsum, g.rowsum_parts, g.M, g.Kp = ab.wfrag.data_ptr(), ab.wfrag.data_ptr(), 1, M, ab.Kp
                g.wpacked, g.w_bits, g.N, g.per_m = ab.wpacked.data_ptr(), pw.bits, pw.N, per_m
                g.cdelta, g.cflush = (ab.cdelta.data_ptr(), ab.cflush.data_ptr()) if not per_m else (None, None)
                g.mdelta, g.mzp = (ab.mdelta.data_ptr(), ab.mzp.data_ptr()) if per_m else (None, None)
                g.L, g.offset = (ab.L if per_m else 1), ab.offset
                g.alpha, g.zw, g.gamma, g.vn = pw.alpha.data_ptr(), pw.zw.data_ptr(), ab.gamma.data_ptr(), (ab.vn.data_ptr() if per_m else None)
                g.y, g.y_dtype, g.ldy = outs[i].data_ptr(), _lib.DTYPE_CODE[outs[i].dtype], outs[i].stride(0)
                g.extra = _c.cast(_c.pointer(ex), _c.c_void_p)
            n_chunk = len(chunk)

            def issue(arr=arr, n_chunk=n_chunk, _keep=extras):
                _lib_call("dgq_gemm_wxa8_batch", n_chunk, _c.cast(arr, _c.c_void_p), _lib.stream())
            issue()
            if GEMM_LAUNCH_HOOK is not None:
                GEMM_LAUNCH_HOOK(issue, [(M, bindings[i], outs[i].element_size(), (M * bindings[i].pw.K * x2.element_size()) if j == 0 else 0)
                                         for j, i in enumerate(chunk)])      # (one shared input: counted once)
    return [o.view(*x.shape[:-1], o.shape[-1]) for o in outs]


def quant_linear_multi(x: torch.Tensor, bindings, ln=None):
    """[quant_linear(x, ab, ln=ln) for ab in bindings] with the launches shared: layers that consume the SAME input — the
    q / k / v projections of a self-attention, the to_k / to_v of every cross-attention (one text context) — are quantised
    by one dgq_quant_act_batch per (kernel variant, scale mode) and multiplied by one dgq_gemm_wxa8_batch per scale mode,
    8 problems per launch.  Same kernels, same arithmetic, same results as the one-layer calls."""
    Kin = x.shape[-1]
    x2 = x.reshape(-1, Kin)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M = x2.shape[0]
    dev = x2.device
    lib = _lib.load()
    if _multi_fuses(bindings, M, Kin, x.dtype, x2):
        return _quant_linear_multi_fused(x, x2, M, bindings, ln)
    lnp = (as_f32(ln[0]), as_f32(ln[1]), float(ln[2])) if ln else None
    qa, keep = [], []
    for ab in bindings:
        assert ab.pw.K == Kin and ab.pw.taps == 1
        per_m = 0 if ab.mode == "perK" else 1
        a = _lib.QuantActArgs()
        a.x, a.x_dtype, a.B, a.H, a.W, a.C, a.kh, a.kw, a.stride, a.pad = x2.data_ptr(), _lib.DTYPE_CODE[x2.dtype], M, 1, 1, Kin, 1, 1, 1, 0
        a.ksrc, a.koff, a.klds = _dp(ab.ksrc), _dp(ab.koff(1, Kin)), _dp(ab.klds(1, Kin))
        a.kdst = _dp(ab.kdst(1, Kin, 1)) if ab.ksrc is not None else None
        a.Kp, a.per_m = ab.Kp, per_m
        a.delta, a.zp = (ab.cdelta.data_ptr(), ab.czp.data_ptr()) if not per_m else (ab.mdelta.data_ptr(), ab.mzp.data_ptr())
        a.L, a.bits = (1 if not per_m else ab.L), ab.abits
        a.pre_scale = a.pre_shift = None
        a.pre_act = 0
        a.ln_gamma, a.ln_beta, a.ln_eps = (lnp[0].data_ptr(), lnp[1].data_ptr(), lnp[2]) if lnp else (None, None, 0.0)
        a.codes = a.rowsum = 1                               # placeholders: dgq_quant_act_variant only validates non-NULL
        parts, a.ksplits = 1, 1                              # the scatter path takes whole rows: ask for it unsplit first
        if a.kdst is None or lib.dgq_quant_act_variant(_c.byref(a)) not in (3, 4):
            parts = act_ksplits(M, ab.Kp)
            a.ksplits = parts
        codes = torch.empty((M, ab.Kp), dtype=torch.int8, device=dev)
        rowsum = torch.empty((parts, M), dtype=torch.float32, device=dev)
        a.codes, a.rowsum = codes.data_ptr(), rowsum.data_ptr()
        variant = lib.dgq_quant_act_variant(_c.byref(a))
        qa.append((variant, per_m, a, codes, rowsum, parts))
    groups = {}
    for i, (variant, per_m, a, *_rest) in enumerate(qa):
        groups.setdefault((variant, per_m), []).append(i)
    for idxs in groups.values():
        for j0 in range(0, len(idxs), 8):
            chunk = idxs[j0:j0 + 8]
            arr = (_lib.QuantActArgs * len(chunk))(*[qa[i][2] for i in chunk])

            def issue_q(arr=arr, n_chunk=len(chunk)):
                _lib_call("dgq_quant_act_batch", n_chunk, _c.cast(arr, _c.c_void_p), _lib.stream())
            issue_q()
            if QUANT_LAUNCH_HOOK is not None:          # one shared input, one code matrix + row sums per problem
                QUANT_LAUNCH_HOOK(issue_q, M * Kin * x2.element_size(), sum(M * bindings[i].Kp + 4 * qa[i][5] * M for i in chunk))
    outs = [torch.empty((M, ab.pw.N), dtype=x.dtype, device=dev) for ab in bindings]
    ggroups = {}
    for i, ab in enumerate(bindings):
        ggroups.setdefault((qa[i][1], ab.pw.bits), []).append(i)
    for (per_m, _bits), idxs in ggroups.items():
        for j0 in range(0, len(idxs), 8):
            chunk = idxs[j0:j0 + 8]
            arr = (_lib.GemmArgs * len(chunk))()
            extras = []
            for g, i in zip(arr, chunk):
                ab, pw = bindings[i], bindings[i].pw
                _v, _pm, _a, codes, rowsum, parts = qa[i]
                g.codes, g.rowsum, g.rowsum_parts, g.M, g.Kp = codes.data_ptr(), rowsum.data_ptr(), parts, M, ab.Kp
                g.wpacked, g.w_bits, g.N, g.per_m = ab.wpacked.data_ptr(), pw.bits, pw.N, per_m
                g.cdelta, g.cflush = (ab.cdelta.data_ptr(), ab.cflush.data_ptr()) if not per_m else (None, None)
                g.mdelta, g.mzp = (ab.mdelta.data_ptr(), ab.mzp.data_ptr()) if per_m else (None, None)
                g.L, g.offset = (ab.L if per_m else 1), ab.offset
                g.alpha, g.zw, g.gamma, g.vn = pw.alpha.data_ptr(), pw.zw.data_ptr(), ab.gamma.data_ptr(), (ab.vn.data_ptr() if per_m else None)
                ex = with_layer_tables(None, ab, 0)
                extras.append(ex)
                g.y, g.y_dtype, g.ldy = outs[i].data_ptr(), _lib.DTYPE_CODE[outs[i].dtype], outs[i].stride(0)
                g.extra = _c.cast(_c.pointer(ex), _c.c_void_p) if ex is not None else None
            n_chunk = len(chunk)

            def issue(arr=arr, n_chunk=n_chunk, _keep=extras):
                _lib_call("dgq_gemm_wxa8_batch", n_chunk, _c.cast(arr, _c.c_void_p), _lib.stream())
            issue()
            if GEMM_LAUNCH_HOOK is not None:
                GEMM_LAUNCH_HOOK(issue, [(M, bindings[i], outs[i].element_size(), 0) for i in chunk])
    keep.append(qa)
    return [o.view(*x.shape[:-1], o.shape[-1]) for o in outs]


def quant_conv2d(x: torch.Tensor, ab: ActBinding, kh, kw, stride, pad, norm=None, residual=None, bias_rows=None, gn_out=True, upsample=False,
                 out=None, out2=None):
    """x logical NCHW (any strides; made channels-last) -> logical NCHW output in channels-last storage.
    norm = (groups, eps, gamma, beta, act): GroupNorm (+SiLU) of x folded into the quantise-on-load pass;
    residual (logical NCHW, same shape as the output) is added in the GEMM epilogue; bias_rows [B][N] likewise, one row
    per image (conv1(...) + time_emb_proj(...)[:, :, None, None]).
    upsample: the layer's input is F.interpolate(x, scale_factor=2, mode="nearest") (Upsample2D.forward); where the quantise
    variant of the layer can read x through that mapping the 4x tensor is never written, otherwise it is materialised here.
    out / out2: [M][N] row views (own row pitch) the output is stored into / stored into AS WELL (OutputRedirect: the halves of a
    channel-concatenation buffer); the returned tensor is a view of ``out`` then."""
    if upsample and (norm is not None or ab.mode == "scalar" or kh * kw == 1):
        x, upsample = F.interpolate(x, scale_factor=2.0, mode="nearest"), False
    B, C, H, W = x.shape
    if upsample:
        H, W = 2 * H, 2 * W                           # the layer's input geometry; x stays the (H/2) x (W/2) source
    xc = x.contiguous(memory_format=torch.channels_last)
    x_store = xc.permute(0, 2, 3, 1)                  # [B,H,W,C] view over the same storage, contiguous
    pre = None
    if norm is not None:
        groups, eps, gamma, beta, act = norm
        gn = _gn_of(x)
        if gn is not None and gn["B"] == B and gn["HW"] == H * W and gn["C"] == C and C % groups == 0:
            sc, sh = groupnorm_from_partials(gn, groups, eps, gamma, beta)      # statistics left by the producing GEMM(s)
        else:
            sc, sh = groupnorm_scale_shift(x_store, B, H * W, C, groups, eps, gamma, beta)
        pre = (sc, sh, act)
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    M = B * Ho * Wo
    res2, res_div = None, 1
    if residual is not None:
        res2 = residual.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(M, ab.pw.N)
    elif bias_rows is not None:                       # [B][N]: one row per image, broadcast over its Ho*Wo positions
        res2, res_div = bias_rows.contiguous(), M // B
    N = ab.pw.N
    if kh == 1 and kw == 1 and stride == 1 and pad == 0 and act_fuses(ab, M, C, x.dtype, x2=x_store.reshape(M, C)):
        # a 1x1 convolution is a Linear layer over the pixels: one launch, the GEMM quantises its own rows (dgq_gemm_act_t)
        part = torch.empty((M // 16, N, 2), dtype=torch.float32, device=x.device) if (GN_FROM_GEMM and gn_out and (Ho * Wo) % 16 == 0 and N % 4 == 0) else None
        y = gemm_act(x_store.reshape(M, C), M, ab, x.dtype, extra=make_extra(res2, res_div=res_div, gn_partial=part, y2=out2), pre=pre, rows_per_image=H * W,
                     out=out)
        out = y.view(B, Ho, Wo, N).permute(0, 3, 1, 2)
        if part is not None:
            out._dgq_gn = dict(parts=[(part, N)], B=B, HW=Ho * Wo, C=N, ver=out._version)
        return out
    if kh * kw > 1 and not upsample and conv_act_fuses(ab, B, H, W, C, kh, kw, stride, pad, x.dtype, x_store):
        # the unfolded operand is quantised INSIDE the GEMM launch from a staged input patch (dgq_gemm_act_t with kh > 1): one launch,
        # no code matrix (QuantLayer.forward, quant_layer.py:626-661, as one kernel)
        part = torch.empty((M // 16, N, 2), dtype=torch.float32, device=x.device) if (GN_FROM_GEMM and gn_out and (Ho * Wo) % 16 == 0 and N % 4 == 0) else None
        act = _lib.GemmAct()
        kp_ = ab.kpat_for(kh, kw, C, 8 + kw - 1)
        act.x, act.x_dtype, act.ldx, act.K = x_store.data_ptr(), _lib.DTYPE_CODE[x_store.dtype], C, C
        act.kpat, act.bits, act.rows_per_image, act.pre_act = kp_.data_ptr(), ab.abits, H * W, 0
        act.B, act.H, act.W, act.kh, act.kw, act.stride, act.pad = B, H, W, kh, kw, stride, pad
        act._keep = [x_store, kp_]
        if ab.mode == "perK":
            act.czp = ab.czp.data_ptr()
        if pre is not None:
            if pre[0] is not None:
                act.pre_scale, act.pre_shift = pre[0].data_ptr(), pre[1].data_ptr()
                act._keep += [pre[0], pre[1]]
            act.pre_act = pre[2]
        extra = with_layer_tables(make_extra(res2, res_div=res_div, gn_partial=part, y2=out2), ab, M)
        extra.act = _c.cast(_c.pointer(act), _c.c_void_p)
        extra._act_keep = act
        y = gemm_wxa8(ab.wfrag, ab.wfrag, M, ab, x.dtype, out=out, extra=extra, _fused_bytes=x.element_size() * B * H * W * C)
        out = y.view(B, Ho, Wo, N).permute(0, 3, 1, 2)
        if part is not None:
            out._dgq_gn = dict(parts=[(part, N)], B=B, HW=Ho * Wo, C=N, ver=out._version)
        return out
    implicit = CONV_IMPLICIT and ab.mode == "scalar" and kh * kw > 1 and C % 16 == 0 and ab.pw.bits == 4 and ab.conv_zero_code() is not None
    conv_desc = None
    if implicit:
        # ONE (δ, z) for the whole operand (the reference's native path F.conv2d(aqtizer(x), ŵ), quant_layer.py:659): every input
        # pixel is quantised once, as a 1x1 layer in natural order, and the GEMM gathers the taps from that int8 NHWC tensor
        # (dgq_gemm_conv_t) — the 9x unfolded operand is never written
        abi = ab.input_binding(C)
        codes, pixsum, _ = quant_act(x_store, B, H, W, C, 1, 1, 1, 0, abi, pre)
        rowsum = torch.empty((1, M), dtype=torch.float32, device=x.device)
        cv = _lib.GemmConv()
        cv.codes_in, cv.pixsum, cv.fill = codes.data_ptr(), pixsum.data_ptr(), ab.conv_fill().data_ptr()
        cv.B, cv.H, cv.W, cv.C, cv.ldc, cv.kh, cv.kw, cv.stride, cv.pad, cv.Ho, cv.Wo = B, H, W, C, abi.Kp, kh, kw, stride, pad, Ho, Wo
        cv.zero_code, cv.pixsum_parts = ab.conv_zero_code(), pixsum.shape[0]
        conv_desc = (cv, pixsum)
    else:
        qa = quant_act(x_store, B, H, W, C, kh, kw, stride, pad, ab, pre, ups=upsample)
        if qa is None:                                # no folded form for this layer's quantise variant: materialise the upsample
            return quant_conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), ab, kh, kw, stride, pad, norm=norm, residual=residual,
                                bias_rows=bias_rows, gn_out=gn_out, out=out, out2=out2)
        codes, rowsum, M = qa
    # GroupNorm partials of the output for whoever normalises it next: from the GEMM's own epilogue, or — a K-split launch —
    # from its combine kernel (DGQ_GN_FROM_SPLITK=0: only unsplit launches, the tensor gets a statistics pass otherwise)
    N = ab.pw.N
    part = None
    if (GN_FROM_GEMM and gn_out and (Ho * Wo) % 16 == 0 and N % 4 == 0 and
            (GN_FROM_SPLITK or _lib.load().dgq_gemm_plan_splits(M, N, ab.Kp, ab.pw.bits, 0 if ab.mode == "perK" else 1, WORKSPACE_BYTES) == 1)):
        part = torch.empty((M // 16, N, 2), dtype=torch.float32, device=x.device)
    y = gemm_wxa8(codes, rowsum, M, ab, x.dtype, out=out, extra=make_extra(res2, res_div=res_div, gn_partial=part, conv=conv_desc, y2=out2))
    out = y.view(B, Ho, Wo, N).permute(0, 3, 1, 2)
    if part is not None:
        out._dgq_gn = dict(parts=[(part, N)], B=B, HW=Ho * Wo, C=N, ver=out._version)
    return out


def conv2d_f32w(x: torch.Tensor, w_nat: torch.Tensor, bias, kh, kw, stride, pad, norm=None, out2=None, gn_out=False):
    """Weight-only state: conv2d / linear of UNQUANTISED activations with the dequantised weight, exact fp32 MFMA with the
    im2col folded into the load (dgq_conv2d_f32w).  x logical NCHW (made channels-last) or [..., K] for a Linear layer
    (kh = kw = 1); w_nat [N][kh·kw·C] fp32 with K in (tap, c) order; bias [N] fp32 or None.
    norm = (groups, eps, gamma, beta, act) folds GroupNorm (+SiLU) of a 4-D x into the load, as in quant_conv2d."""
    N = w_nat.shape[0]
    is_conv = x.dim() == 4 and w_nat.shape[1] == kh * kw * x.shape[1]
    if x.dim() == 4 and not is_conv:
        # a Linear layer fed a 4-D [..., K] input (F.linear takes any rank): the rows are everything but the last dimension
        assert kh == kw == 1 and w_nat.shape[1] == x.shape[-1], "dgq conv2d_f32w: weight [N][%d] does not match input %s" % (w_nat.shape[1], tuple(x.shape))
    if is_conv:
        B, C, H, W = x.shape
        xs = x.contiguous(memory_format=torch.channels_last)
        sc = sh = None
        act = 0
        if norm is not None:
            groups, eps, gamma, beta, act = norm
            gn = _gn_of(x)
            if gn is not None and gn["B"] == B and gn["HW"] == H * W and gn["C"] == C and C % groups == 0:
                sc, sh = groupnorm_from_partials(gn, groups, eps, gamma, beta)
            else:
                sc, sh = groupnorm_scale_shift(xs.permute(0, 2, 3, 1), B, H * W, C, groups, eps, gamma, beta)
        Ho = (H + 2 * pad - kh) // stride + 1
        Wo = (W + 2 * pad - kw) // stride + 1
        y = torch.empty((B * Ho * Wo, N), dtype=x.dtype, device=x.device)
        part = None
        if gn_out and GN_FROM_GEMM and (Ho * Wo) % 16 == 0 and N % 4 == 0 and N > 8:      # statistics for whoever normalises the output next
            part = torch.empty((B * Ho * Wo // 16, N, 2), dtype=torch.float32, device=x.device)
        _lib_call("dgq_conv2d_f32w", _lib.ptr(xs), _lib.DTYPE_CODE[x.dtype], B, H, W, C, kh, kw, stride, pad,
                  _lib.ptr(w_nat), _lib.ptr(bias), N, _lib.ptr(y), _lib.DTYPE_CODE[y.dtype], N,
                  _lib.ptr(sc), _lib.ptr(sh), int(act), _lib.ptr(out2), out2.stride(0) if out2 is not None else 0, _lib.ptr(part), _lib.stream())
        out = y.view(B, Ho, Wo, N).permute(0, 3, 1, 2)
        if part is not None:
            out._dgq_gn = dict(parts=[(part, N)], B=B, HW=Ho * Wo, C=N, ver=out._version)
        return out
    K = x.shape[-1]
    assert w_nat.shape[1] == K and kh == kw == 1, "dgq conv2d_f32w: weight [N][%d] does not match input %s" % (w_nat.shape[1], tuple(x.shape))
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    y = torch.empty((x2.shape[0], N), dtype=x.dtype, device=x.device)
    assert norm is None
    _lib_call("dgq_conv2d_f32w", _lib.ptr(x2), _lib.DTYPE_CODE[x.dtype], x2.shape[0], 1, 1, K, 1, 1, 1, 0,
              _lib.ptr(w_nat), _lib.ptr(bias), N, _lib.ptr(y), _lib.DTYPE_CODE[y.dtype], N, None, None, 0, None, 0, None, _lib.stream())
    return y.view(*x.shape[:-1], N)


# ------------------------------------------------------------------------------------------ attention side
def fakequant_rows(x2d: torch.Tensor, T, D, mode, delta, zp, skip, bits, out=None):
    """x2d [rows][C] contiguous; see dgq_fakequant_rows."""
    assert x2d.is_contiguous()
    rows, C = x2d.shape
    out = x2d if out is None else out
    _lib_call("dgq_fakequant_rows", _lib.ptr(x2d), _lib.ptr(out), _lib.DTYPE_CODE[x2d.dtype], rows, C, T, D, mode,
              _lib.ptr(delta), _lib.ptr(zp), skip, bits, _lib.stream())
    return out


def timestep_embedding(timesteps: torch.Tensor, dim: int, out_dtype=torch.float32):
    """Timesteps.forward (diffusers_rewrite/sd.py:19-39) as one launch: ``timesteps`` [rows] int64 or fp32 (any stride: the expanded
    single timestep has stride 0) -> [rows, dim] = cat(cos, sin)."""
    assert timesteps.dim() == 1 and timesteps.dtype in (torch.int64, torch.float32) and dim % 2 == 0
    rows = timesteps.shape[0]
    out = torch.empty((rows, dim), dtype=out_dtype, device=timesteps.device)
    _lib_call("dgq_timestep_embedding", _lib.ptr(timesteps), 1 if timesteps.dtype == torch.float32 else 0, int(timesteps.stride(0)), rows, dim,
              _lib.ptr(out), _lib.DTYPE_CODE[out_dtype], _lib.stream())
    return out


def _dense_layout(t):
    """0: [N,C,H,W] contiguous, 1: channels-last dense, None: neither (4-D tensors; 2-/3-D contiguous counts as 0)"""
    if t.is_contiguous():
        return 0
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return 1
    return None


def cfg_ddim_step(eps_uncond, eps_cond, sample, guidance, s1, inv_s2, s3, s4):
    """eps = e_u + guidance·(e_c − e_u) (``eps_cond`` None: eps = e_u), then the DDIM update s3·((x − s1·eps)·inv_s2) + s4·eps in the
    order of the eager chain (pipeline_stable_diffusion.py:1037-1044, scheduling_ddim.py); fp32 / fp16 / bf16 tensors (one dtype; a 16-bit
    chain rounds after every statement as the eager one does); ``sample`` contiguous, the eps halves in
    the same layout or channels-last (what the UNet returns).  Returns None when the layouts are not of that kind."""
    lay = _dense_layout(eps_uncond)
    if (lay is None or not sample.is_contiguous() or sample.dim() < 2 or eps_uncond.shape != sample.shape
            or (eps_cond is not None and (eps_cond.shape != sample.shape or _dense_layout(eps_cond) != lay))):
        return None
    for t in (eps_uncond, eps_cond):
        assert t is None or t.dtype == sample.dtype
    C = sample.shape[1]
    HW = sample.numel() // (sample.shape[0] * C)
    out = torch.empty_like(sample)
    _lib_call("dgq_cfg_ddim_step", _lib.ptr(eps_uncond), _lib.ptr(eps_cond), _lib.ptr(sample), _lib.ptr(out), _lib.DTYPE_CODE[sample.dtype], sample.numel(), C, HW, lay,
              _c.c_float(guidance), _c.c_float(s1), _c.c_float(inv_s2), _c.c_float(s3), _c.c_float(s4), _lib.stream())
    return out


def max_f32(p: torch.Tensor, skip_cols=0):
    assert p.is_contiguous() and p.dtype == torch.float32
    S = p.shape[-1]
    rows = p.numel() // S
    out = torch.zeros((1,), dtype=torch.float32, device=p.device)
    _lib_call("dgq_max_f32", _lib.ptr(p), rows, S, skip_cols, _lib.ptr(out), _lib.stream())
    return out


def logquant_f32(p: torch.Tensor, delta: torch.Tensor, bits, skip_cols=0, out=None):
    assert p.is_contiguous() and p.dtype == torch.float32
    S = p.shape[-1]
    rows = p.numel() // S
    out = p if out is None else out
    _lib_call("dgq_logquant_f32", _lib.ptr(p), _lib.ptr(out), rows, S, skip_cols, _lib.ptr(delta), bits, _lib.stream())
    return out


SMALLM_MAX_M, SMALLM_MAX_K, SMALLM_MAX_PROBLEMS = 16, 2048, 24


def linear_smallm_batch(x2d: torch.Tensor, bindings, pre_act=0):
    """[ab.pw(aqtizer(act(x2d))) for ab in bindings] for scalar-quantizer Linear layers sharing the input x2d [M <= 16][K]
    in ONE launch per 24 layers (dgq_linear_smallm_batch): the time-embedding projections of a UNet forward."""
    assert x2d.dim() == 2 and x2d.stride(1) == 1 and x2d.dtype in _lib.DTYPE_CODE
    M, K = x2d.shape
    outs = []
    for i0 in range(0, len(bindings), SMALLM_MAX_PROBLEMS):
        chunk = bindings[i0:i0 + SMALLM_MAX_PROBLEMS]
        probs = (_lib.SmallMProblem * len(chunk))()
        keep = []
        for j, ab in enumerate(chunk):
            pw = ab.pw
            assert ab.mode != "perK" and ab.L == 1 and pw.K == K and pw.taps == 1
            y = torch.empty((M, pw.N), dtype=x2d.dtype, device=x2d.device)
            q = probs[j]
            q.wpacked, q.alpha, q.zw = ab.wpacked.data_ptr(), pw.alpha.data_ptr(), pw.zw.data_ptr()
            q.gamma, q.vn = ab.gamma.data_ptr(), ab.vn.data_ptr()
            q.mdelta, q.mzp = ab.mdelta.data_ptr(), ab.mzp.data_ptr()
            q.y, q.ldy, q.N, q.Kp, q.w_bits, q.a_bits = y.data_ptr(), y.stride(0), pw.N, ab.Kp, pw.bits, ab.abits
            keep.append(y)
        _lib_call("dgq_linear_smallm_batch", _lib.ptr(x2d), _lib.DTYPE_CODE[x2d.dtype], M, K, x2d.stride(0), pre_act,
                  len(chunk), _c.cast(probs, _c.c_void_p), _lib.DTYPE_CODE[x2d.dtype], _lib.stream())
        outs += keep
    return outs


def minmax_rows_cols(x2d: torch.Tensor, rows=True, cols=True):
    """Row-wise and column-wise (min, max) of a contiguous [R][C] view — the statistics of DGQ's calibration producer
    (dgq_minmax_rows_cols).  Returns (rowmin, rowmax, colmin, colmax), fp32, None for a pair that was not requested."""
    assert x2d.dim() == 2 and x2d.stride(1) == 1 and x2d.dtype in _lib.DTYPE_CODE
    R, C = x2d.shape
    dev = x2d.device
    rmin = rmax = cmin = cmax = part = None
    slices = max(1, min(256, R // 64))
    if rows:
        rmin = torch.empty((R,), dtype=torch.float32, device=dev)
        rmax = torch.empty((R,), dtype=torch.float32, device=dev)
    if cols:
        cmin = torch.empty((C,), dtype=torch.float32, device=dev)
        cmax = torch.empty((C,), dtype=torch.float32, device=dev)
        part = torch.empty((2 * slices * C,), dtype=torch.float32, device=dev)
    _lib_call("dgq_minmax_rows_cols", _lib.ptr(x2d), _lib.DTYPE_CODE[x2d.dtype], R, C, x2d.stride(0),
              _lib.ptr(rmin), _lib.ptr(rmax), _lib.ptr(cmin), _lib.ptr(cmax), _lib.ptr(part), slices, _lib.stream())
    return rmin, rmax, cmin, cmax


ATTN_HEAD_DIMS = (8, 16, 40, 64, 80, 160)
_ATTN_WS = {}


def _attn_workspace(device, nbytes):
    """The attention kernels' scratch (δ scalar, row statistics, K / V tile images): one grow-only buffer per device and
    stream chain instead of an allocation per call — attentions of one forward run back to back on one stream, each fully
    overwrites what it reads.  Under graph capture a buffer allocated here belongs to the graph's pool like any other."""
    cap = torch.cuda.is_current_stream_capturing()
    key = (str(device), torch.cuda.current_stream(device).cuda_stream if not cap else "capture")
    buf = _ATTN_WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty((max(nbytes, 1 << 20),), dtype=torch.uint8, device=device)      # caching allocator: 512-byte aligned
        if not cap:
            _ATTN_WS[key] = buf
    return buf


def attention_fuses_fakequant(D, mode):
    """True where dgq_attention_f32 applies the q/k/v fake-quantizers itself (``fq=``)."""
    return bool(_lib.load().dgq_attention_fuses_fakequant(D, mode))


def attention(q, k, v, H, D, scale, mode, skip, delta, bits, fq=None):
    """q [B,T,H*D], k/v [B,S,H*D] contiguous (fp32; fp16 / bf16 in the quantised modes) -> o [B,T,H*D]; see dgq_attention.
    fq: optional 3-tuple for q, k, v of None | (mode, delta, zp, skip, bits) — the aqtizer_q/k/v quantizers applied on
    load (only where ``attention_fuses_fakequant(D, mode)``)."""
    assert q.dtype in _lib.DTYPE_CODE and k.dtype == q.dtype and v.dtype == q.dtype
    assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
    B, T, _ = q.shape
    S = k.shape[1]
    nbytes = _lib.load().dgq_attention_workspace_bytes(B, H, T, S, D)
    ws = _attn_workspace(q.device, nbytes)
    desc = None
    if fq is not None and any(f is not None for f in fq):
        desc = (_lib.AttnFq * 3)()
        for i, f in enumerate(fq):
            if f is None:
                desc[i].mode = -1
                continue
            fmode, fd, fz, fskip, fbits = f
            ntok = (T if i == 0 else S) - fskip
            need = 1 if fmode == 0 else (ntok if fmode == 1 else D)
            assert fd.numel() == need and fz.numel() == need and fd.dtype == torch.float32 and fd.is_contiguous(), \
                "q/k/v quantizer table has %d entries, kernel addresses %d" % (fd.numel(), need)
            desc[i].mode, desc[i].skip, desc[i].bits = fmode, fskip, fbits
            desc[i].delta, desc[i].zero_point = _lib.ptr(fd), _lib.ptr(fz)
    o = torch.empty_like(q)

    def issue():
        _lib_call("dgq_attention", _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), _lib.DTYPE_CODE[q.dtype], B, H, T, S, D,
                  _c.c_float(scale), mode, skip, _lib.ptr(delta), bits,
                  _c.cast(desc, _c.c_void_p) if desc is not None else None, _lib.ptr(ws), nbytes, _lib.stream())
    issue()
    if ATTN_LAUNCH_HOOK is not None:                   # Q·Kᵀ and P·V: 4·T·S·D flops per (batch, head); q, k, v read and o written once
        ATTN_LAUNCH_HOOK(issue, 4.0 * T * S * D * B * H, (2 * B * T + 2 * B * S) * H * D * q.element_size())
    return o


attention_f32 = attention


def attention_sync_timeouts():
    """Workgroups of single-launch attention calls (csrc/attn_one.hip) that ever gave up the wait for the real-time δ exchange
    in this process; synchronises.  0 unless a grid was not resident as a whole."""
    n = _lib.load().dgq_attention_sync_timeouts()
    if n < 0:
        raise RuntimeError("dgq_attention_sync_timeouts failed")
    return n
