"""T2ILogQuantizer — log2 quantizer for post-softmax attention probabilities
(reference: quant/quant_layer_text.py:12-138).  Executes as dgq_max_f32 + dgq_logquant_f32 on the GPU."""
import torch
import torch.nn as nn

from .. import ops
from .quant_layer import Scaler


class T2ILogQuantizer(nn.Module):
    def __init__(self, bits: int = 8, symmetric: bool = False, channel_wise: bool = False,
                 scaler: Scaler = Scaler.MINMAX, leaf_param: bool = False, always_zero: bool = True,
                 quant_emb: bool = False, real_time: bool = False, log_max_1: bool = False) -> None:
        super().__init__()
        self.level = 2 ** bits
        self.symmetric = symmetric
        self.channel_wise = channel_wise
        self.scaler = scaler
        self.leaf_param = leaf_param
        if leaf_param:
            self.x_log_max = None
        self.running_stat = False
        self.always_zero = always_zero
        self.delta = None
        self.zero_point = None
        self.init = False
        self.quant_emb = quant_emb
        self.real_time = real_time
        self.NB, self.PB = 0, self.level - 1          # asymmetric / always_zero (quant_layer_text.py:44-45)
        self.log_max_1 = log_max_1

    @property
    def bits(self):
        return int(self.level).bit_length() - 1

    def _init_quantization_param(self, x: torch.Tensor) -> torch.Tensor:
        """Best of the 0.999/0.9999/0.99999 quantiles under an L2 score (quant_layer_text.py:49-76).
        Load-time only (non-real-time mode); torch ops on the device."""
        xc = x.detach().float()
        delta, best = xc.max(), 1e10
        flat = xc.reshape(-1)
        for pct in (0.999, 0.9999, 0.99999):
            try:
                nd = torch.quantile(flat, pct)
            except RuntimeError:            # quantile() input size limit: exact k-th value with interpolation
                pos = pct * (flat.numel() - 1)
                lo = int(pos)
                vals = torch.topk(flat, flat.numel() - lo, sorted=True)[0]
                a, b = vals[-1], vals[-2] if vals.numel() > 1 else vals[-1]
                nd = a + (b - a) * (pos - lo)
            xq = ops.logquant_f32(xc.contiguous().clone(), nd.reshape(1).float(), self.bits)
            score = (xc - xq).abs().pow(2).mean()
            if score < best:
                best, delta = score, nd
        return delta

    def forward(self, x: torch.Tensor, skip_cols: int = 0) -> torch.Tensor:
        """``skip_cols`` > 0: leading key columns bypass the quantizer (start-peak, sd.py:191-195) — the
        reference slices and re-concatenates; here the kernel skips them in place."""
        if not x.is_cuda:
            raise RuntimeError("dgq_amd: T2ILogQuantizer executes on the GPU only (no CPU fallback)")
        xc = x.contiguous().float()
        if not self.init and not self.real_time:
            view = xc[..., skip_cols:] if skip_cols else xc
            d = self._init_quantization_param(view)
            self.delta = nn.Parameter(d) if self.leaf_param else d
            self.init = True
        if self.log_max_1:
            self.delta.data = torch.tensor(1.0, device=x.device)
        if self.running_stat and not self.real_time:
            raise NotImplementedError("act_momentum_update is calibration-time")
        if self.real_time:
            delta = ops.max_f32(xc, skip_cols)                       # δ = x.max() over the whole tensor (:96-97)
        else:
            delta = self.delta.detach().reshape(1).float().to(x.device)
        out = xc if xc.data_ptr() != x.data_ptr() else torch.empty_like(xc)
        return ops.logquant_f32(xc, delta, self.bits, skip_cols, out=out)

    def bitwidth_refactor(self, bits: int = 8) -> None:
        self.level = 2 ** bits

    def extra_repr(self) -> str:
        return "level=%d, real_time=%s" % (self.level, self.real_time)

    def half(self):
        super().half()
        return self

    def float(self):
        super().float()
        return self
