"""Host-side mirror of the reference's ``quant/quant_layer.py`` operator interface, executing on MI355X.

Same public names and argument meaning (``Scaler``, ``UniformAffineQuantizer``, ``QuantLayer``, ``QMODE``,
``StraightThrough``), different execution model:

  reference (quant_layer.py:626-661)                      here
  ---------------------------------------------------    -----------------------------------------------------
  fake-quant weights from fp32 on EVERY call (:642-643)   integer codes frozen once -> int4/int8 packed in HBM
  F.unfold materialises im2col in fp32 (:630-638)         gather + quantise straight to int8 codes (dgq_quant_act)
  6 elementwise fp passes per activation (:297-299)       fused into that single pre-pass
  fp32 F.linear / matmul / F.conv2d (:562,:659)           V_MFMA_I32_16X16X64_I8 GEMM, dequant in the epilogue
  ~750 host->device δ/z copies per step when time-aware   all timestep slots resident on the device

There is no CPU execution path: a quantized forward on a CPU tensor raises.
"""
import logging
import os
from enum import Enum
from typing import List, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..plan import plan_act

logger = logging.getLogger(__name__)


class StraightThrough(nn.Module):
    def forward(self, x):
        return x


def minmax(x: torch.Tensor, symmetric: bool = False, level: int = 256, always_zero: bool = False):
    """Scalar min/max scale initialiser — same arithmetic as quant_layer.py:22-38 (python-double range,
    δ clamped to ≥1e-8, z = rne(−min/δ))."""
    x_min, x_max = min(x.min().item(), 0.0), max(x.max().item(), 0.0)
    delta = torch.tensor(float(x_max - x_min) / (level - 1))
    if symmetric:
        m = max(abs(x_min), x_max)
        x_min, x_max = -m, m
        delta = torch.tensor(float(x_max - x_min) / (level - 2))
    if always_zero:
        delta = torch.tensor(float(x_max) / (level - 1))
    if delta < 1e-8:
        delta = torch.tensor(1e-8)
    if symmetric or always_zero:
        zero_point = torch.tensor(0.0)
    else:
        zero_point = torch.round(-torch.tensor(float(x_min)) / delta)
    return delta.to(device=x.device, dtype=x.dtype), zero_point.to(device=x.device, dtype=x.dtype)


def lp_loss(pred: torch.Tensor, tgt: torch.Tensor, p: float = 2.0):
    """quant_layer.py:198-209 with reduction ALL: mean |pred − tgt|^p over every element."""
    return (pred - tgt).abs().pow(p).mean()


def channel_mse(w: torch.Tensor, level: int):
    """``mse`` for every output channel at once (asymmetric weights): what quant_layer.py:253-264 computes with a python loop of
    80 full passes per channel.  Same arithmetic per channel — ranges shrunk in double precision and rounded to fp32 once,
    z = rne(fp32(−min)/δ), error = mean |·|^2.4 — the candidates of all channels evaluated together; a channel keeps the
    FIRST candidate with the smallest error, as the reference's strict ``<`` does."""
    flat = w.detach().reshape(w.shape[0], -1).float()
    mn64, mx64 = flat.min(dim=1)[0].double(), flat.max(dim=1)[0].double()
    best = torch.full((flat.shape[0],), 1e+10, device=flat.device)
    delta = torch.zeros_like(best)
    zp = torch.zeros_like(best)
    for i in range(80):
        f = 1. - (i * 0.01)
        new_min, new_max = mn64 * f, mx64 * f
        nd = ((new_max - new_min) / (level - 1)).float()
        nz = torch.round((-new_min).float() / nd)
        x_q = torch.clamp(torch.round(flat / nd[:, None]) + nz[:, None], 0, level - 1)
        err = (nd[:, None] * (x_q - nz[:, None]) - flat).abs().pow(2.4).mean(dim=1)
        better = err < best
        best = torch.where(better, err, best)
        delta = torch.where(better, nd, delta)
        zp = torch.where(better, nz, zp)
    shape = (-1,) + (1,) * (w.dim() - 1)
    return delta.view(shape).to(w.dtype), zp.view(shape).to(w.dtype)


def mse(x: torch.Tensor, symmetric: bool = False, level: int = 256, always_zero: bool = False):
    """The scalar L2.4 range search (quant_layer.py:62-86) = ``channel_mse`` of the tensor taken as ONE channel: all 80 shrink
    candidates are ranked by the same batched evaluation; 0-d (δ, z) like the reference returns."""
    if symmetric or always_zero:
        return _mse_flagged(x, symmetric, level, always_zero)
    d, z = channel_mse(x.reshape(1, -1), level)
    return d.reshape(()), z.reshape(())


def _mse_flagged(x: torch.Tensor, symmetric: bool, level: int, always_zero: bool):
    """The same search under the quantizer flags (quant_layer.py:72-80): ``always_zero`` (the uniform softmax quantizer aqtizer_w,
    quant_block.py:145-156: δ = max/(level − 1), z = 0, codes in [0, level − 1]) and ``symmetric`` (δ = 2·max|range|/(level − 2),
    z = 0, codes in [−level/2, level/2 − 1]).  The 80 shrink candidates are evaluated eight at a time over the flattened tensor;
    the first candidate with the smallest error wins (the reference's strict ``<``)."""
    flat = x.detach().reshape(-1).float()
    x_min, x_max = float(flat.min()), float(flat.max())
    f = 1.0 - 0.01 * torch.arange(80, dtype=torch.float64)
    new_min, new_max = x_min * f, x_max * f
    if always_zero:
        deltas = (new_max / (level - 1)).float()                     # torch.tensor(python double): rounded to fp32 once
        nb, pb = 0, level - 1
    else:
        m = torch.maximum(new_min.abs(), new_max)
        deltas = (m + m) / (level - 2)                               # stays a python double in the reference: the division below
        nb, pb = -level // 2, level // 2 - 1                         # rounds it to fp32 when it meets the fp32 tensor
    deltas = deltas.to(flat.device)
    best, best_d = None, None
    for i0 in range(0, 80, 8):
        d = deltas[i0:i0 + 8].float()[:, None]
        x_q = torch.clamp(torch.round(flat[None, :] / d), nb, pb)
        err = (d * x_q - flat[None, :]).abs().pow(2.4).mean(dim=1)
        for j in range(err.numel()):
            if best is None or float(err[j]) < best:
                best, best_d = float(err[j]), deltas[i0 + j]
    delta = best_d.float() if always_zero else best_d                # (symmetric: the reference returns the python double)
    return torch.as_tensor(delta).to(dtype=x.dtype, device=x.device).reshape(()), torch.tensor(0.0, dtype=x.dtype, device=x.device)


def _outside_scope(name):
    def scaler(x, symmetric=False, level=256, always_zero=False):
        raise NotImplementedError("Scaler.%s is not selected by any DGQ recipe (SURVEY.md §2: out of scope); use MINMAX or MSE" % name)
    scaler.__name__ = name.lower()
    return scaler


class Scaler(Enum):
    """Same member names as the reference enum (quant_layer.py:187-193); the DGQ recipes select MINMAX (``--fast``) or MSE."""
    MINMAX = minmax
    MSE = mse
    KL = _outside_scope("KL")
    HIST = _outside_scope("HIST")
    OMSE = _outside_scope("OMSE")
    LOGMINMAX = _outside_scope("LOGMINMAX")


QMODE = Enum("QMODE", ("QDIFF", "NORMAL", "PTQD"))


def channel_minmax(w: torch.Tensor, level: int):
    """Vectorised per-output-channel ``minmax`` (what quant_layer.py:253-264 computes with a python loop
    over channels — the dominant cost of the reference's load)."""
    flat = w.detach().reshape(w.shape[0], -1)
    mn = torch.clamp(flat.min(dim=1)[0], max=0.0).double()
    mx = torch.clamp(flat.max(dim=1)[0], min=0.0).double()
    delta = ((mx - mn) / float(level - 1)).float()
    delta = torch.where(delta < 1e-8, torch.full_like(delta, 1e-8), delta)
    zp = torch.round(-mn.float() / delta)
    shape = (-1,) + (1,) * (w.dim() - 1)
    return delta.view(shape), zp.view(shape)


def group_params_from_ranges(in_min, in_max, out_min, out_max, group_num, mode, level, force_in_channel=None):
    """The host half of ``done_group_num`` (quant_layer.py:338-418) as a pure function of the folded range vectors —
    returns (δ, z, in_channel_wise) in the checkpoint's broadcast shapes.  Arithmetic follows the reference line by
    line: numpy float32 ranges -> float64 K-Means -> per-cluster range in float64 -> δ rounded to fp32 by
    ``torch.tensor`` -> z = rne(−min/δ) in fp32."""
    import os
    import numpy as np
    from sklearn.cluster import KMeans
    imin, imax = in_min.detach().cpu().numpy().flatten(), in_max.detach().cpu().numpy().flatten()
    omin, omax = out_min.detach().cpu().numpy().flatten(), out_max.detach().cpu().numpy().flatten()
    in_spread = imax.max() - imax.min() + imin.max() - imin.min()
    out_spread = omax.max() - omax.min() + omin.max() - omin.min()
    in_channel_wise = bool(in_spread > out_spread or os.environ.get("IN_CHANNEL_WISE", False))
    if force_in_channel is not None:
        in_channel_wise = force_in_channel
    data = np.column_stack((imin, imax)) if in_channel_wise else np.column_stack((omin, omax))
    km = KMeans(n_clusters=group_num, random_state=0).fit(data)
    labels = km.labels_
    if mode == "mean":
        center = km.cluster_centers_
    elif mode == "minmax":
        center = []
        for i in range(group_num):
            cl = data[labels == i]
            center.append([cl.min(), cl.max()] if cl.size else [0.0, 1.0])
        center = np.array(center)
    else:
        raise NotImplementedError(mode)
    base = (in_min if in_channel_wise else out_min).detach().clone().cpu().float().flatten()
    delta, zero_point = base.clone(), base.clone()
    lab = torch.from_numpy(labels.astype("int64"))
    for i in range(group_num):
        d = torch.tensor((center[i, 1] - center[i, 0]) / (level - 1))
        if d < 1e-8:
            d = torch.tensor(1e-8)
        d = d.float()
        delta[lab == i] = d
        zero_point[lab == i] = torch.round(torch.tensor(-center[i, 0]).float() / d)
    shape = (1, 1, -1) if in_channel_wise else (1, -1, 1)
    return delta.view(shape), zero_point.view(shape), in_channel_wise


class UniformAffineQuantizer(nn.Module):
    """δ·(clamp(rne(x/δ)+z, 0, 2^b−1) − z)  — quant_layer.py:216-299 (inference branch; the calibration branches of
    :284-293 are ``observe``)."""

    def __init__(self, bits: int = 8, symmetric: bool = False, channel_wise: bool = False,
                 scaler: Scaler = Scaler.MINMAX, leaf_param: bool = False, always_zero: bool = False,
                 quant_emb: bool = False) -> None:
        super().__init__()
        if symmetric:
            raise NotImplementedError("symmetric quantizers are not used by the DGQ inference path")
        self.level = 2 ** bits
        self.symmetric = symmetric
        self.channel_wise = channel_wise
        self.scaler = scaler
        self.leaf_param = leaf_param
        if leaf_param:
            self.x_min, self.x_max = None, None
        self.running_stat = False
        self.always_zero = always_zero
        self.delta = None
        self.zero_point = None
        self.init = False
        self.quant_emb = quant_emb
        self.group_num = -1
        # calibration producer state (quant_layer.py:247-248): one (min, max) pair of vectors per observed batch
        self.min_max_per_in_channel = []
        self.min_max_per_out_channel = []

    @property
    def bits(self):
        return int(self.level).bit_length() - 1

    # -- calibration producer (DGQ activation calibration, quant_layer.py:284-293, 301-446) ------------------------
    def calibrating(self) -> bool:
        return self.running_stat or self.group_num != -1

    def observe(self, x: torch.Tensor) -> None:
        """The statistics side of the reference's forward (quant_layer.py:284-293), called with the tensor the quantizer
        sees (the unfolded operand for grouped convs; [B,H,T,D] for the attention-side quantizers)."""
        if not self.init:
            self.init_from(x)
        if self.running_stat:
            self.act_momentum_update(x)
        if 1 < self.group_num:
            if x.dim() > 2:
                self.record_min_max_ema(x)
        elif self.group_num != -1:
            self.act_momentum_update(x)

    def record_min_max_ema(self, x: torch.Tensor, act_range_momentum: float = 0.95) -> None:
        """quant_layer.py:301-313: per-"in-channel" (last dim) and per-"out-channel" (second-to-last dim; the token /
        unfolded-row axis) minima and maxima of this batch, reduced over everything else — one pass of
        dgq_minmax_rows_cols over the tensor viewed as [rows][C] plus a fold of the batch (/head) index."""
        if x.dim() not in (3, 4):
            raise NotImplementedError("DGQ statistics are defined for 3-D (Linear / unfolded conv) and 4-D (attention) inputs")
        xc = x.detach().contiguous()
        C, T = xc.shape[-1], xc.shape[-2]
        rmin, rmax, cmin, cmax = ops.minmax_rows_cols(xc.view(-1, C))
        self.min_max_per_in_channel.append((cmin, cmax))
        self.min_max_per_out_channel.append((rmin.view(-1, T).min(dim=0)[0], rmax.view(-1, T).max(dim=0)[0]))

    def act_momentum_update(self, x: torch.Tensor, act_range_momentum: float = 0.95) -> None:
        """quant_layer.py:431-446: EMA of the scalar range, δ/z from the EMA range (the reference builds a clipped copy
        of x whose min / max ARE the EMA values and feeds it to Scaler.MINMAX; the result depends on those two only)."""
        assert self.init and self.leaf_param
        x_min, x_max = x.data.min().float(), x.data.max().float()
        self.x_min = self.x_min * act_range_momentum + x_min * (1.0 - act_range_momentum)
        self.x_max = self.x_max * act_range_momentum + x_max * (1.0 - act_range_momentum)
        rng = torch.stack([self.x_min, self.x_max]).to(x.dtype)
        delta, self.zero_point = self.scaler(rng, self.symmetric, self.level, self.always_zero)
        self.delta = nn.Parameter(delta)

    def done_group_num(self, group_num, mode):
        """quant_layer.py:315-429: fold the recorded batches (min of mins / max of maxes), pick the axis with the larger
        spread, K-Means(group_num, random_state=0) on the (min, max) pairs (host, scikit-learn like the reference), one
        (δ, z) per cluster from the cluster's range ('minmax') or centre ('mean'), broadcast shape (1,1,X) / (1,X,1)."""
        if self.min_max_per_in_channel == []:
            self.group_num = -1
            return None
        import numpy as np
        from sklearn.cluster import KMeans
        in_min = torch.stack([m[0] for m in self.min_max_per_in_channel]).min(dim=0)[0]
        in_max = torch.stack([m[1] for m in self.min_max_per_in_channel]).max(dim=0)[0]
        out_min = torch.stack([m[0] for m in self.min_max_per_out_channel]).min(dim=0)[0]
        out_max = torch.stack([m[1] for m in self.min_max_per_out_channel]).max(dim=0)[0]
        delta, zero_point, _ = group_params_from_ranges(in_min, in_max, out_min, out_max, group_num, mode, self.level)
        self.last_ranges = tuple(t.detach().cpu() for t in (in_min, in_max, out_min, out_max))    # kept for inspection / tests
        dev = in_min.device
        self.delta.data = delta.to(dev)
        self.zero_point = zero_point.to(dev)
        self.group_num = -1
        self.min_max_per_in_channel = []
        self.min_max_per_out_channel = []
        return self.delta.data, self.zero_point

    # -- initialisation -------------------------------------------------------------------------
    def _init_quantization_param(self, x: torch.Tensor, channel_wise: bool = False):
        if channel_wise:
            if self.scaler is Scaler.MINMAX:
                return channel_minmax(x, self.level)
            if self.scaler is Scaler.MSE and not self.symmetric and not self.always_zero:
                return channel_mse(x, self.level)
            # any other scaler: the reference's loop over output channels (quant_layer.py:253-264)
            dz = [self.scaler(x[c].detach(), self.symmetric, self.level, self.always_zero) for c in range(x.shape[0])]
            shape = (-1,) + (1,) * (x.dim() - 1)
            delta = torch.stack([torch.as_tensor(d, dtype=x.dtype, device=x.device).reshape(()) for d, _ in dz]).view(shape)
            zp = torch.stack([torch.as_tensor(z, dtype=x.dtype, device=x.device).reshape(()) for _, z in dz]).view(shape)
            return delta, zp
        if self.leaf_param:
            self.x_min, self.x_max = x.data.min(), x.data.max()
        return self.scaler(x, self.symmetric, self.level, self.always_zero)

    def init_from(self, x: torch.Tensor):
        delta, zp = self._init_quantization_param(x, self.channel_wise)
        self.delta = nn.Parameter(delta) if self.leaf_param else delta
        self.zero_point = zp
        self.init = True

    # -- forward --------------------------------------------------------------------------------
    def layout(self, x: torch.Tensor):
        """(x2d, mode, T, D) for dgq_fakequant_rows given this quantizer's δ shape and a contiguous x:
        scalar; (1,1,X) = last dim; (1,X,1) = second-to-last dim (token); [N,1(,1,1)] = dim 0 (weights)."""
        d = self.delta
        C = x.shape[-1]
        if d.numel() == 1:
            return x.view(-1, C), 0, 1, C
        if d.dim() == x.dim() and d.shape[0] == x.shape[0] and d.numel() == x.shape[0]:   # per output channel
            return x.view(x.shape[0], -1), 1, x.shape[0], x[0].numel()
        if d.dim() == 3 and d.shape[0] == 1 and d.shape[1] == 1 and d.shape[2] == C:
            return x.view(-1, C), 2, 1, C
        if d.dim() == 3 and d.shape[0] == 1 and d.shape[2] == 1 and x.dim() >= 2 and d.shape[1] == x.shape[-2]:
            return x.view(-1, C), 1, x.shape[-2], C
        raise NotImplementedError("quantizer δ shape %s on input %s" % (tuple(d.shape), tuple(x.shape)))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.init:
            self.init_from(x)
        if not x.is_cuda:
            raise RuntimeError("dgq_amd: quantizers execute on the GPU only (no CPU fallback)")
        xc = x.contiguous()
        x2d, mode, T, D = self.layout(xc)
        d = self.delta.detach().reshape(-1).float().to(x.device)
        z = torch.as_tensor(self.zero_point).detach().reshape(-1).float().to(x.device)
        if z.numel() == 1 and d.numel() > 1:
            z = z.expand_as(d).contiguous()
        out = torch.empty_like(xc)
        ops.fakequant_rows(x2d, T, D, mode, d, z, 0, self.bits, out=out.view(x2d.shape))
        return out

    def bitwidth_refactor(self, bits: int = 8) -> None:
        self.level = 2 ** bits

    def extra_repr(self) -> str:
        return "level=%d, channel_wise=%s, leaf_param=%s, always_zero=%s" % (
            self.level, self.channel_wise, self.leaf_param, self.always_zero)

    def half(self):
        super().half()
        if self.delta is not None and not isinstance(self.delta, nn.Parameter):
            self.delta = self.delta.half()
        if torch.is_tensor(self.zero_point) and not isinstance(self.zero_point, nn.Parameter):
            self.zero_point = self.zero_point.half()
        return self

    def float(self):
        super().float()
        if self.delta is not None and not isinstance(self.delta, nn.Parameter):
            self.delta = self.delta.float()
        if torch.is_tensor(self.zero_point) and not isinstance(self.zero_point, nn.Parameter):
            self.zero_point = self.zero_point.float()
        return self


#: Test instrumentation (tests/test_gpu_unet.py::teacher-forced checks).  When set to a callable, every call of a quantized
#: layer that runs on the integer path — the plain ``forward`` and the fused entry points the blocks use
#: (``forward_fused`` / ``forward_prenorm`` / ``forward_residual``) — reports
#: ``LAYER_TAP(layer, y, x=..., prologue=bool, residual=..., bias_rows=...)`` and continues with the tensor it returns,
#: so a test can compare every operator of the FUSED graph with the oracle and teacher-force it.  None in production.
LAYER_TAP = None
#: weight-only state on the library's own exact-fp32 kernel (dgq_conv2d_f32w); =0: F.linear / F.conv2d on the dequantised weight
WEIGHT_ONLY_HIP = True
#: ... and the FP state (unquantised layers: conv_in / conv_out of a quantized model) too, outside autograd; =0: MIOpen / rocBLAS
FP_STATE_HIP = True


def _tap(layer, y, **info):
    if LAYER_TAP is None:
        return y
    # (a tensor the tap REPLACES carries no GroupNorm partials: they describe the tensor the GEMM wrote, and the next
    # GroupNorm then takes its statistics from the replacement itself — tests/test_gpu_kernels.py pins the partials path)
    return LAYER_TAP(layer, y, **info)


class SlotRef:
    """Shared by every layer of one QuantModel: which timestep slot's activation tables are live
    (time-aware mode, calibration.py:297-312). ``None`` = use the quantizer modules' own δ/z."""

    def __init__(self):
        self.slot = None


class QuantLayer(nn.Module):
    """Drop-in for the reference's QuantLayer (quant_layer.py:577-702) around nn.Linear / nn.Conv2d."""

    QMAP = {nn.Conv2d: F.conv2d, nn.Linear: F.linear}

    def __init__(self, layer: Union[nn.Conv2d, nn.Linear], wq_params: dict = {}, aq_params: dict = {},
                 disable_aq: bool = False, aq_mode: List[int] = [QMODE.QDIFF.value], quant_emb: bool = False) -> None:
        super().__init__()
        self.wq_params = dict(wq_params)
        self.aq_params = dict(aq_params)
        self.fwd_kwargs = {}
        self.is_conv = isinstance(layer, nn.Conv2d)
        if self.is_conv:
            if layer.groups != 1 or layer.dilation != (1, 1):
                raise NotImplementedError("grouped/dilated convolutions do not occur in the SD/SDXL UNets")
            self.fwd_kwargs = dict(stride=layer.stride, padding=layer.padding, dilation=layer.dilation,
                                   groups=layer.groups)
        self.kwd_func = self.QMAP[type(layer)]
        self.w = layer.weight
        self.original_w = self.w.data.clone()
        self.b = None
        self.original_b = None
        if layer.bias is not None:
            self.b = layer.bias
            self.original_b = self.b.data.clone()
        self.use_wq = False
        self.use_aq = False
        self.disable_aq = disable_aq
        self.aq_mode = aq_mode
        self.quant_emb = quant_emb
        self.wq_params["quant_emb"] = quant_emb
        self.wqtizer = UniformAffineQuantizer(**self.wq_params)
        self.aqtizer = UniformAffineQuantizer(**self.aq_params)
        self.split = 0
        self.act_func = StraightThrough()
        self.ignore_recon = False
        self.extra_repr = layer.extra_repr
        self.use_group_num = False
        # device-side state (not part of the state-dict)
        self._pw = None
        self._pw_key = None
        self._wdq = None
        self._wnat = None
        self._bindings = {}
        self._act_tables = {}            # slot -> (δ, z) CPU tensors from the cali_ckpt
        self._slot_ref: Optional[SlotRef] = None
        #: ff.net.0 of a GEGLU feed-forward (set by QuantBasicTransformerBlock): the packed weight keeps its rows interleaved
        #: (2i = value column i, 2i+1 = gate column i) so that dgq_gemm_wxa8's pair epilogue can form value·gelu(gate)
        self.geglu_rows = False

    # -- geometry ---------------------------------------------------------------------------------
    @property
    def in_channels(self):
        return self.w.shape[1]

    @property
    def taps(self):
        return self.w.shape[2] * self.w.shape[3] if self.is_conv else 1

    # -- weights ------------------------------------------------------------------------------------
    def _weight_key(self):
        q = self.wqtizer
        alpha = getattr(q, "alpha", None)
        return (self.w.data_ptr(), self.w._version, str(self.w.device),
                q.delta.data_ptr() if torch.is_tensor(q.delta) else None,
                q.delta._version if torch.is_tensor(q.delta) else None,
                q.zero_point._version if torch.is_tensor(q.zero_point) else None,
                alpha._version if alpha is not None else None,
                self.b._version if self.b is not None else None)

    def packed_weight(self) -> ops.PackedWeight:
        """Freeze the weight to integer codes (once; re-done only if a parameter was modified)."""
        q = self.wqtizer
        if not q.init:
            q.init_from(self.w.data)
        key = self._weight_key()
        if self._pw is None or key != self._pw_key:
            dev = self.w.device
            w, d, z = self.w.data.float(), q.delta.data.to(dev), torch.as_tensor(q.zero_point).data.to(dev)
            alpha, b = getattr(q, "alpha", None), (self.b.data if self.b is not None else None)
            if self.geglu_rows:
                rp = self._row_perm(dev)
                w, d, z = w[rp], d.reshape(w.shape[0], -1)[rp], z.reshape(w.shape[0], -1)[rp]
                alpha = alpha.data[rp] if alpha is not None else None
                b = b[rp] if b is not None else None
            self._pw = ops.PackedWeight(w, d, z, alpha, b, q.bits, self.in_channels, self.taps)
            self._pw_key = key
            self._bindings = {}
            self._wdq = None
            self._wnat = None
        return self._pw

    def _row_perm(self, dev):
        """packed row 2i = reference row i (value half), 2i+1 = reference row i + N/2 (gate half)"""
        N = self.w.shape[0]
        return torch.stack([torch.arange(N // 2, device=dev), torch.arange(N // 2, device=dev) + N // 2], 1).flatten()

    def _row_unperm(self, dev):
        return torch.argsort(self._row_perm(dev))

    def dequantized_weight(self, dtype):
        """δ·(q − z) as a tensor: weight-only mode feeds it to the library GEMM/conv."""
        pw = self.packed_weight()
        if self._wdq is None or self._wdq.dtype != dtype:
            w = pw.alpha[:, None] * (pw.codes.float() - pw.zp_true[:, None])
            if self.geglu_rows:
                w = w[self._row_unperm(w.device)]
            w = w.view(self.w.shape)
            self._wdq = w.to(dtype)
            if self.is_conv:
                self._wdq = self._wdq.contiguous(memory_format=torch.channels_last)
        return self._wdq

    def dequantized_weight_natural(self):
        """(δ·(q − z) as fp32 [N][taps·C] with K in (tap, c) order, bias as fp32 or None) — the operands of dgq_conv2d_f32w."""
        self.packed_weight()                              # (re-packs, and drops the caches below, when the weight state changed)
        if self._wnat is None:
            w = self.dequantized_weight(torch.float32).float()
            if self.is_conv:
                w = w.permute(0, 2, 3, 1)                    # [N][kh][kw][C]
            self._wnat = (w.reshape(w.shape[0], -1).contiguous(), self.b.float().contiguous() if self.b is not None else None)
        return self._wnat

    # -- activation tables ----------------------------------------------------------------------------
    def set_act_table(self, slot, delta, zero_point):
        self._act_tables[slot] = (delta, zero_point)
        self._bindings.pop(slot, None)

    def _binding(self) -> ops.ActBinding:
        pw = self.packed_weight()
        slot = self._slot_ref.slot if self._slot_ref is not None else None
        if slot is not None and slot in self._act_tables:
            key = slot
            if key not in self._bindings:
                d, z = self._act_tables[slot]
                lay = plan_act(d, z, "conv" if self.is_conv else "linear", self.in_channels, self.taps, self.aqtizer.bits,
                               kw=self.w.shape[3] if self.is_conv else 1)
                self._bindings[key] = ops.ActBinding(lay, pw, self.aqtizer.bits)
            return self._bindings[key]
        a = self.aqtizer
        d = a.delta
        z = torch.as_tensor(a.zero_point)
        key = ("live", d.data_ptr(), d._version, z.data_ptr() if z.numel() else 0, z._version)
        if key not in self._bindings:
            self._bindings = {k: v for k, v in self._bindings.items() if not (isinstance(k, tuple) and k[0] == "live")}
            lay = plan_act(d.data, z.data, "conv" if self.is_conv else "linear", self.in_channels, self.taps, a.bits,
                           kw=self.w.shape[3] if self.is_conv else 1)
            self._bindings[key] = ops.ActBinding(lay, pw, a.bits)
        return self._bindings[key]

    # -- forward ----------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, split: int = 0) -> torch.Tensor:
        quant_act = self.use_aq and not self.disable_aq
        if not self.use_wq:
            w = self.original_w.to(device=x.device, dtype=x.dtype)
            b = self.original_b.to(device=x.device, dtype=x.dtype) if self.original_b is not None else None
            if quant_act:
                x = self.aqtizer(x)
            if (FP_STATE_HIP and x.is_cuda and x.dtype in ops.FLOAT_DTYPES and not torch.is_grad_enabled()
                    and (not self.is_conv or (tuple(self.fwd_kwargs.get("dilation", (1, 1)))[0] == 1 and self.fwd_kwargs.get("groups", 1) == 1))):
                # FP state on the GPU without autograd (conv_in / conv_out of every quantized model, quant_model.py:66-73): the
                # library's exact-fp32 kernel instead of MIOpen / rocBLAS (conv_out's four output channels take its N <= 8 form;
                # step 9.537 vs 9.559 ms with MIOpen)
                self._fp_natural(x.device)
                if self.is_conv:
                    # (gn_out: conv_in's output carries its GroupNorm partials on this path too — with or without an output redirect
                    # the first resnet's norm1 and the last up resnet's see the same statistics)
                    return ops.conv2d_f32w(x, self._wnat_fp[1], self._wnat_fp[2], self.w.shape[2], self.w.shape[3],
                                           self.fwd_kwargs["stride"][0], self.fwd_kwargs["padding"][0], gn_out=True)
                return ops.conv2d_f32w(x, self._wnat_fp[1], self._wnat_fp[2], 1, 1, 1, 0)
            if self.is_conv:
                return F.conv2d(x, w, b, stride=self.fwd_kwargs["stride"], padding=self.fwd_kwargs["padding"])
            return F.linear(x, w, b)
        if not x.is_cuda:
            raise RuntimeError("dgq_amd.QuantLayer: the quantized path runs only through the HIP kernels on the GPU "
                               "(no CPU fallback); got a %s tensor" % x.device)
        if not quant_act:
            if getattr(self.wqtizer, "soft_tgt", False):
                # weight reconstruction in progress (reconstruction.py:37-41): the differentiable soft-rounded weight,
                # re-evaluated every call like the reference's wqtizer(self.w) (quant_layer.py:642-643)
                if not self.wqtizer.init:
                    self.wqtizer.init_from(self.w.data)
                w = self.wqtizer(self.w)
                if self.is_conv:
                    w = w.contiguous(memory_format=torch.channels_last)
            else:
                if (WEIGHT_ONLY_HIP and x.dtype in ops.FLOAT_DTYPES and not (torch.is_grad_enabled() and x.requires_grad)
                        and (not self.is_conv or (tuple(self.fwd_kwargs.get("dilation", (1, 1)))[0] == 1 and self.fwd_kwargs.get("groups", 1) == 1))):
                    # inference in the weight-only state: exact-fp32 MFMA kernel of this library (dgq_conv2d_f32w), not
                    # F.linear / F.conv2d of the vendor libraries
                    wn, bn = self.dequantized_weight_natural()
                    if self.is_conv:
                        return ops.conv2d_f32w(x, wn, bn, self.w.shape[2], self.w.shape[3], self.fwd_kwargs["stride"][0],
                                               self.fwd_kwargs["padding"][0])
                    return ops.conv2d_f32w(x, wn, bn, 1, 1, 1, 0)
                w = self.dequantized_weight(x.dtype)
            b = self.b.to(x.dtype) if self.b is not None else None
            if self.is_conv:
                return F.conv2d(x, w, b, stride=self.fwd_kwargs["stride"], padding=self.fwd_kwargs["padding"])
            return F.linear(x, w, b)
        if not self.aqtizer.init and not (self._slot_ref is not None and self._slot_ref.slot in self._act_tables):
            self.aqtizer.init_from(x)           # first-forward self-initialisation (quant_layer.py:274-278)
        if self.aqtizer.calibrating():          # DGQ calibration: statistics of what the quantizer sees (:284-293, :630-641)
            seen = x
            if self.is_conv and self.use_group_num:
                seen = F.unfold(x, kernel_size=tuple(self.w.shape[2:]), dilation=1, padding=self.fwd_kwargs["padding"][0],
                                stride=self.fwd_kwargs["stride"][0])
            self.aqtizer.observe(seen)
        ab = self._binding()
        if self.is_conv:
            kh, kw = self.w.shape[2], self.w.shape[3]
            return _tap(self, ops.quant_conv2d(x, ab, kh, kw, self.fwd_kwargs["stride"][0], self.fwd_kwargs["padding"][0]),
                        x=x, prologue=False)
        y = ops.quant_linear(x, ab)
        if self.geglu_rows:                     # the plain path returns the reference's column order
            y = y.index_select(-1, self._row_unperm(y.device))
        return _tap(self, y, x=x, prologue=False)

    def _take_redirect(self, x: torch.Tensor, scale: int = 1):
        """(out, out2) of a pending ops.OutputRedirect for this convolution's output on input x (spatially scaled by ``scale``), or
        (None, None): asked only by the calls whose result IS their module's result (``final=True``)."""
        if ops.pending_redirect() is None or not self.is_conv or x.dim() != 4:
            return None, None
        kh, kw = self.w.shape[2], self.w.shape[3]
        st, pd = self.fwd_kwargs["stride"][0], self.fwd_kwargs["padding"][0]
        Ho = (x.shape[2] * scale + 2 * pd - kh) // st + 1
        Wo = (x.shape[3] * scale + 2 * pd - kw) // st + 1
        return ops.take_redirect(x.shape[0] * Ho * Wo, self.w.shape[0], x.dtype)

    def forward_final(self, x: torch.Tensor) -> torch.Tensor:
        """``self(x)`` where the result is the calling module's own result (Downsample2D.conv, the UNet's conv_in): the one place a
        pending ops.OutputRedirect is honoured on the plain path — integer path or FP state of a convolution, no tap, no hooks."""
        if (ops.pending_redirect() is not None and self.is_conv and x.is_cuda and x.dtype in ops.FLOAT_DTYPES and LAYER_TAP is None and not self.aqtizer.calibrating()
                and not self._forward_hooks and not self._forward_pre_hooks and not torch.is_grad_enabled()):
            kh, kw = self.w.shape[2], self.w.shape[3]
            st, pd = self.fwd_kwargs["stride"][0], self.fwd_kwargs["padding"][0]
            if self.on_integer_path(x):
                out, out2 = self._take_redirect(x)
                return ops.quant_conv2d(x, self._binding(), kh, kw, st, pd, out=out, out2=out2)
            if (not self.use_wq and not (self.use_aq and not self.disable_aq) and FP_STATE_HIP
                    and tuple(self.fwd_kwargs.get("dilation", (1, 1)))[0] == 1 and self.fwd_kwargs.get("groups", 1) == 1):
                out, out2 = self._take_redirect(x)
                if out is None:                                   # (the FP kernel takes the second copy only)
                    self._fp_natural(x.device)
                    return ops.conv2d_f32w(x, self._wnat_fp[1], self._wnat_fp[2], kh, kw, st, pd, out2=out2, gn_out=True)
                ops.pending_redirect().taken = False
        return self(x)

    def forward_upsampled(self, x: torch.Tensor, final: bool = False) -> torch.Tensor:
        """``self(F.interpolate(x, scale_factor=2.0, mode="nearest"))`` — Upsample2D.forward (diffusers_rewrite/sd.py).  On the integer
        path of a k x k convolution the quantise-on-load pass reads x through the (h/2, w/2) mapping and the upsampled tensor is
        never written (ops.quant_conv2d(upsample=True)); with a layer tap installed, during calibration or in any other state the
        interpolate runs as written."""
        if (LAYER_TAP is None and self.is_conv and self.on_integer_path(x) and x.dtype in ops.FLOAT_DTYPES and not self.aqtizer.calibrating()
                and not self._forward_hooks and not self._forward_pre_hooks and self.w.shape[2] * self.w.shape[3] > 1):
            kh, kw = self.w.shape[2], self.w.shape[3]
            out, out2 = self._take_redirect(x, 2) if final else (None, None)
            return ops.quant_conv2d(x, self._binding(), kh, kw, self.fwd_kwargs["stride"][0], self.fwd_kwargs["padding"][0], upsample=True,
                                    out=out, out2=out2)
        return self(F.interpolate(x, scale_factor=2.0, mode="nearest"))

    def on_integer_path(self, x: torch.Tensor) -> bool:
        """True when forward(x) would run dgq_quant_act + dgq_gemm_wxa8 (weights and activations quantised, GPU)."""
        return (self.use_wq and self.use_aq and not self.disable_aq and x.is_cuda
                and (self.aqtizer.init or (self._slot_ref is not None and self._slot_ref.slot in self._act_tables)))

    def forward_fused(self, x: torch.Tensor, pre_act: int = 0, residual=None, fq=None, ln=None, geglu: bool = False) -> torch.Tensor:
        """``fq(self(act(x))) + residual`` with the elementwise pieces folded into the two kernels of the layer:
        pre_act 1 = SiLU(x), 2 = GEGLU (x[..., :K]·gelu(x[..., K:])) inside the quantise-on-load pass; ``fq`` (an
        attention-side quantizer, see ops.make_extra) and ``residual`` inside the GEMM epilogue.  Linear layers on the
        integer path only; anything else falls back to the unfused sequence of the same kernels/ops.  ``ln`` = an
        nn.LayerNorm module applied to x first, folded into the same load pass (per-row statistics in the kernel).
        ``geglu``: the layer is the GEGLU projection ff.net.0 and the result is value·gelu(gate), [..., N/2] — formed in the
        GEMM epilogue when the packed rows are interleaved (``geglu_rows``)."""
        if ln is not None and (self.is_conv or not self.on_integer_path(x) or x.dtype not in ops.FLOAT_DTYPES or pre_act
                               or x.shape[-1] % 4 or x.shape[-1] > 2048):
            x = ln(x)                                           # nn.LayerNorm module: unfused
            ln = None
        if self.is_conv or not self.on_integer_path(x) or x.dtype not in ops.FLOAT_DTYPES:
            if pre_act == 1:
                x = F.silu(x)
            elif pre_act == 2:
                a, g = x.chunk(2, dim=-1)
                x = a * F.gelu(g)
            y = self(x)                                           # through __call__: forward hooks (data_utill.py) see it
            if geglu:
                a, g = y.chunk(2, dim=-1)
                y = a * F.gelu(g)
            if fq is not None:
                mode, dd, zz, T, D, skip, bits = fq
                y = y.contiguous()
                ops.fakequant_rows(y.view(-1, y.shape[-1]), T, D, mode - 1, dd, zz, skip, bits)
            return y if residual is None else y + residual
        lnp = (ln.weight, ln.bias, float(ln.eps)) if ln is not None else None
        if geglu and not self.geglu_rows:                         # rows not interleaved: GEGLU as its own step
            y = self.forward_fused(x, pre_act=pre_act, ln=ln)
            a, g = y.chunk(2, dim=-1)
            return a * F.gelu(g)
        return _tap(self, ops.quant_linear(x, self._binding(), pre_act=pre_act, residual=residual, fq=fq, ln=lnp, geglu=geglu),
                    x=x, prologue=bool(pre_act or lnp is not None), residual=residual, fq=fq, geglu=geglu)

    def can_fuse_prenorm(self, x: torch.Tensor) -> bool:
        """True when this layer runs on the integer path, so a preceding GroupNorm(+SiLU) can be folded into its
        quantise-on-load pass (dgq_groupnorm_scale_shift + dgq_quant_act prologue)."""
        return (self.is_conv and self.use_wq and self.use_aq and not self.disable_aq and x.is_cuda
                and x.dtype in ops.FLOAT_DTYPES
                and (self.aqtizer.init or (self._slot_ref is not None and self._slot_ref.slot in self._act_tables)))

    def forward_prenorm(self, x: torch.Tensor, norm: nn.GroupNorm, silu: bool = True, residual=None, bias_rows=None, final: bool = False) -> torch.Tensor:
        """conv(act(GroupNorm(x))) [+ residual | + bias_rows[b, :, None, None]] without materialising the normalised tensor.
        final: the result is the calling module's own result (a pending ops.OutputRedirect may place it)."""
        ab = self._binding()
        kh, kw = self.w.shape[2], self.w.shape[3]
        out, out2 = self._take_redirect(x) if (final and LAYER_TAP is None) else (None, None)
        return _tap(self, ops.quant_conv2d(x, ab, kh, kw, self.fwd_kwargs["stride"][0], self.fwd_kwargs["padding"][0],
                                           norm=(norm.num_groups, norm.eps, norm.weight, norm.bias, 1 if silu else 0),
                                           residual=residual, bias_rows=bias_rows, out=out, out2=out2),
                    x=x, prologue=True, residual=residual, bias_rows=bias_rows)

    # -- Linear projections of a Transformer2D with use_linear_projection (SDXL, sdxl.py: proj_in / proj_out are nn.Linear over the
    #    channels of an NCHW tensor): on the integer path they are 1x1 convolutions of the channels-last tensor, so the
    #    GroupNorm in front of proj_in folds into its quantise-on-load pass, the residual behind proj_out into its GEMM epilogue,
    #    and proj_out leaves GroupNorm partials for the next resnet block — as the Conv2d projections of SD do.
    def can_fuse_tokens(self, x: torch.Tensor) -> bool:
        return (not self.is_conv and self.w.dim() == 2 and self.use_wq and self.use_aq and not self.disable_aq and x.is_cuda
                and x.dtype in ops.FLOAT_DTYPES
                and (self.aqtizer.init or (self._slot_ref is not None and self._slot_ref.slot in self._act_tables)))

    def forward_prenorm_tokens(self, x: torch.Tensor, norm: nn.GroupNorm) -> torch.Tensor:
        """Linear(GroupNorm(x).permute(0, 2, 3, 1).reshape(B, HW, C)) for x [B, C, H, W] -> [B, HW, N]."""
        y = ops.quant_conv2d(x, self._binding(), 1, 1, 1, 0, norm=(norm.num_groups, norm.eps, norm.weight, norm.bias, 0), gn_out=False)
        b, n, hh, ww = y.shape
        return _tap(self, y.permute(0, 2, 3, 1).reshape(b, hh * ww, n), x=x, prologue=True)

    def forward_residual_tokens(self, h: torch.Tensor, residual: torch.Tensor, final: bool = False) -> torch.Tensor:
        """Linear(h).reshape(B, H, W, N).permute(0, 3, 1, 2) + residual for tokens h [B, HW, C] and residual [B, N, H, W].
        final: the result is the calling module's own result (a pending ops.OutputRedirect may place it)."""
        b, n, hh, ww = residual.shape
        x = h.reshape(b, hh, ww, h.shape[-1]).permute(0, 3, 1, 2)
        o1, o2 = ops.take_redirect(b * hh * ww, n, x.dtype) if (final and LAYER_TAP is None) else (None, None)
        y = ops.quant_conv2d(x, self._binding(), 1, 1, 1, 0, residual=residual, out=o1, out2=o2)
        yt = y.permute(0, 2, 3, 1).reshape(b, hh * ww, n)
        rt = residual.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(b, hh * ww, n)
        out = _tap(self, yt, x=h, prologue=False, residual=rt)
        if out is yt:
            return y                                        # (keeps the GroupNorm partials attached by quant_conv2d)
        return out.reshape(b, hh, ww, n).permute(0, 3, 1, 2)

    def can_fuse_prenorm_fp(self, x: torch.Tensor) -> bool:
        """FP-state conv on the GPU outside autograd (conv_out of a quantized model): GroupNorm + SiLU fold into the load of
        dgq_conv2d_f32w."""
        return (FP_STATE_HIP and self.is_conv and not self.use_wq and not (self.use_aq and not self.disable_aq) and x.is_cuda
                and not self._forward_hooks and not self._forward_pre_hooks     # (data capture hooks must see a plain call)
                and x.dtype in ops.FLOAT_DTYPES and not torch.is_grad_enabled()
                and tuple(self.fwd_kwargs.get("dilation", (1, 1)))[0] == 1 and self.fwd_kwargs.get("groups", 1) == 1)

    def forward_prenorm_fp(self, x: torch.Tensor, norm: nn.GroupNorm, silu: bool = True) -> torch.Tensor:
        wn, bn = self._fp_natural(x.device)
        return ops.conv2d_f32w(x, wn, bn, self.w.shape[2], self.w.shape[3], self.fwd_kwargs["stride"][0], self.fwd_kwargs["padding"][0],
                               norm=(norm.num_groups, norm.eps, norm.weight, norm.bias, 1 if silu else 0))

    def _fp_natural(self, device):
        key = (self.original_w.data_ptr(), self.original_w._version, str(device))
        if getattr(self, "_wnat_fp", None) is None or self._wnat_fp[0] != key:
            wf = self.original_w.detach().to(device=device, dtype=torch.float32)
            if self.is_conv:
                wf = wf.permute(0, 2, 3, 1)
            bf = self.original_b.detach().to(device=device, dtype=torch.float32).contiguous() if self.original_b is not None else None
            self._wnat_fp = (key, wf.reshape(wf.shape[0], -1).contiguous(), bf)
        return self._wnat_fp[1], self._wnat_fp[2]

    def forward_residual(self, x: torch.Tensor, residual, final: bool = False) -> torch.Tensor:
        """conv(x) + residual with the add in the GEMM epilogue (integer path), else unfused.
        final: the result is the calling module's own result (a pending ops.OutputRedirect may place it)."""
        if self.is_conv and self.on_integer_path(x) and x.dtype in ops.FLOAT_DTYPES:
            kh, kw = self.w.shape[2], self.w.shape[3]
            out, out2 = self._take_redirect(x) if (final and LAYER_TAP is None) else (None, None)
            return _tap(self, ops.quant_conv2d(x, self._binding(), kh, kw, self.fwd_kwargs["stride"][0],
                                               self.fwd_kwargs["padding"][0], residual=residual, out=out, out2=out2),
                        x=x, prologue=False, residual=residual)
        return self(x) + residual

    # -- state switches (quant_layer.py:663-686) -----------------------------------------------------------
    def set_quant_state(self, use_wq: bool = False, use_aq: bool = False) -> None:
        self.use_wq = use_wq if not self.ignore_recon else False
        self.use_aq = use_aq if not self.ignore_recon else False

    def set_running_stat(self, running_stat: bool) -> None:
        self.aqtizer.running_stat = running_stat

    def set_group_num(self, group_num: int = 1) -> None:
        """quant_layer.py:604-608: start collecting DGQ statistics; convs switch to the unfolded-operand semantics."""
        self.aqtizer.group_num = group_num
        self.use_group_num = True

    def done_group_num(self, group_num, mode) -> None:
        self.aqtizer.done_group_num(group_num, mode)
        self._bindings = {}

    def half(self):
        super().half()
        self.original_w = self.original_w.half()
        if self.original_b is not None:
            self.original_b = self.original_b.half()
        return self

    def float(self):
        super().float()
        self.original_w = self.original_w.float()
        if self.original_b is not None:
            self.original_b = self.original_b.float()
        return self

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self.original_w = fn(self.original_w)
        if self.original_b is not None:
            self.original_b = fn(self.original_b)
        return self
