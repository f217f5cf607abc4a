"""AdaRoundQuantizer — weight quantizer with learned rounding; at inference
``δ·(clamp(floor(w/δ) + (α≥0) + z, 0, 2^b−1) − z)`` (reference: quant/adaptive_rounding.py:12-90,
hard mode :51,58-70).  Hard (inference) mode: the codes are produced by dgq_quantize_weight with the α tensor.
Soft mode (``soft_tgt = True``, reconstruction time, adaptive_rounding.py:39-41,55-57): the differentiable
``floor(w/δ) + h(α)`` through dgq_adaround_soft_fwd / _bwd (``ops.adaround_soft``)."""
from enum import Enum

import torch
import torch.nn as nn

from .. import ops
from .quant_layer import UniformAffineQuantizer

RMODE = Enum("RMODE", ("LEARNED_ROUND_SIGMOID", "NEAREST", "NEAREST_STE", "STOCHASTIC", "LEARNED_HARD_SIGMOID"))


class AdaRoundQuantizer(nn.Module):
    def __init__(self, uaqtizer: UniformAffineQuantizer, w: torch.Tensor,
                 rmode: RMODE = RMODE.LEARNED_HARD_SIGMOID) -> None:
        super().__init__()
        if rmode != RMODE.LEARNED_HARD_SIGMOID:
            raise NotImplementedError("only LEARNED_HARD_SIGMOID is used by load_cali_model (calibration.py:20-43)")
        self.level = uaqtizer.level
        self.symmetric = uaqtizer.symmetric
        self.delta = uaqtizer.delta
        self.zero_point = uaqtizer.zero_point
        self.rmode = rmode
        self.soft_tgt = False
        self.gamma, self.zeta = -0.1, 1.1
        self.init = True
        self.channel_wise = True
        self.init_alpha(w.clone())

    @property
    def bits(self):
        return int(self.level).bit_length() - 1

    def init_alpha(self, x: torch.Tensor) -> None:
        """α initialised so that the soft target reproduces the fractional part (adaptive_rounding.py:31-38);
        overwritten by the checkpoint's learned α."""
        delta = self.delta.to(x.device)
        rest = (x / delta) - torch.floor(x / delta)
        self.alpha = nn.Parameter(-torch.log((self.zeta - self.gamma) / (rest - self.gamma) - 1))

    def init_from(self, x):
        pass

    def get_soft_tgt(self) -> torch.Tensor:
        """h(α) = clamp(sigmoid(α)·(ζ − γ) + γ, 0, 1) (adaptive_rounding.py:39-40) — inspection / logging; the loop's
        regulariser evaluates it inside ops.adaround_reg."""
        return torch.clamp(torch.sigmoid(self.alpha) * (self.zeta - self.gamma) + self.gamma, 0, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("dgq_amd: AdaRoundQuantizer executes on the GPU only (no CPU fallback)")
        if self.soft_tgt:
            z = torch.as_tensor(self.zero_point).data.to(x.device)
            if self.alpha.device != x.device:
                self.alpha.data = self.alpha.data.to(x.device)
            return ops.adaround_soft(x, self.delta.data.to(x.device), z, self.alpha, self.bits).to(x.dtype)
        d = self.delta.data.to(x.device)
        z = torch.as_tensor(self.zero_point).data.to(x.device)
        codes = ops.quantize_weight(x.float(), d, z, self.alpha.data.to(x.device), self.bits)
        return (d.reshape(-1, 1).float() * (codes.float() - z.reshape(-1, 1).float())).view(x.shape).to(x.dtype)

    def extra_repr(self) -> str:
        return "level=%d, rmode=%s" % (self.level, self.rmode)
