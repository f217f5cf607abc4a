"""Loss of the weight-PTQ reconstruction loop (reference: quant/reconstruction_util.py:13-198).

``LossFunc`` = output reconstruction loss (MSE / the two Fisher-weighted forms) + the AdaRound rounding regulariser
``w·Σ(1 − |2h(α) − 1|^b)`` with the temperature ``b`` annealed linearly from ``b_range[0]`` to ``b_range[1]`` after the
warm-up.  Same names, arguments and schedule as the reference; the regulariser of every AdaRound layer is ONE fused
forward and ONE fused backward kernel (``ops.adaround_reg``) instead of seven elementwise torch kernels.
"""
import logging
from enum import Enum
from typing import Iterable, Union

import torch

from .. import ops
from .adaptive_rounding import AdaRoundQuantizer
from .quant_block import BaseQuantBlock
from .quant_layer import QuantLayer

logger = logging.getLogger(__name__)

RLOSS = Enum("RLOSS", ("RELAXATION", "MSE", "FISHER_DIAG", "FISHER_FULL", "NONE"))
print_freq = 2000


def lp_loss(pred: torch.Tensor, tgt: torch.Tensor, p: float = 2.0) -> torch.Tensor:
    """Σ over dim 1 of |pred − tgt|^p, mean over the rest (quant_layer.py:199-205, REDUCTION.NONE)."""
    return (pred - tgt).abs().pow(p).sum(1).mean()


class LinearTempDecay:
    """b(t): ``start_b`` until ``rel_start_decay·t_max``, then linear down to ``end_b`` at ``t_max``
    (reconstruction_util.py:176-198)."""

    def __init__(self, t_max: int, rel_start_decay: float = 0.2, start_b: float = 10, end_b: float = 2) -> None:
        self.t_max = t_max
        self.start_decay = rel_start_decay * t_max
        self.start_b = start_b
        self.end_b = end_b

    def __call__(self, t) -> float:
        if t < self.start_decay:
            return self.start_b
        rel_t = (t - self.start_decay) / (self.t_max - self.start_decay)
        return self.end_b + (self.start_b - self.end_b) * max(0.0, 1 - rel_t)


def adaround_layers(o: Union[QuantLayer, BaseQuantBlock]) -> Iterable[QuantLayer]:
    """The layers whose rounding the regulariser covers: the layer itself, or every non-embedding QuantLayer of the block
    that is not excluded from reconstruction (reconstruction_util.py:66-78)."""
    if isinstance(o, QuantLayer):
        return [o]
    return [m for m in o.modules() if isinstance(m, QuantLayer) and not m.quant_emb and not m.ignore_recon]


def round_loss_of(q: AdaRoundQuantizer, b: float) -> torch.Tensor:
    """Σ (1 − |2h(α) − 1|^b) of one quantizer; one fused kernel each way on the GPU."""
    return ops.adaround_reg(q.alpha, b)


class LossFunc:
    def __init__(self, o: Union[QuantLayer, BaseQuantBlock], round_loss: RLOSS = RLOSS.RELAXATION, w: float = 1.0,
                 rec_loss: RLOSS = RLOSS.MSE, max_count: int = 2000, b_range: tuple = (10, 2), decay_start: float = 0.0,
                 warmup: float = 0.0, p: float = 2.0) -> None:
        self.o = o
        self.round_loss = round_loss
        self.w = w
        self.rec_loss = rec_loss
        self.loss_start = max_count * warmup
        self.p = p
        self.temp_decay = LinearTempDecay(t_max=max_count, rel_start_decay=warmup + (1 - warmup) * decay_start,
                                          start_b=b_range[0], end_b=b_range[1])
        self.count = 0
        self.record = False        # tests: keep (count, total, rec, round, b) of every call (costs a host sync per call)
        self.history = []

    def reconstruction(self, pred, tgt, grad=None):
        if self.rec_loss == RLOSS.MSE:
            return lp_loss(pred, tgt, p=self.p)
        if self.rec_loss == RLOSS.FISHER_DIAG:
            return ((pred - tgt).pow(2) * grad.pow(2)).sum(1).mean()
        if self.rec_loss == RLOSS.FISHER_FULL:
            a = (pred - tgt).abs()
            grad = grad.abs()
            batch_dotprod = torch.sum(a * grad, (1, 2, 3)).view(-1, 1, 1, 1)
            return (batch_dotprod * a * grad).mean() / 100
        raise ValueError("Not supported reconstruction loss function: {}".format(self.rec_loss))

    def __call__(self, pred: torch.Tensor, tgt: torch.Tensor, grad: torch.Tensor = None) -> torch.Tensor:
        self.count += 1
        rec_loss = self.reconstruction(pred, tgt, grad)
        b = self.temp_decay(self.count)
        if self.count < self.loss_start or self.round_loss == RLOSS.NONE:
            b = round_loss = 0
        elif self.round_loss == RLOSS.RELAXATION:
            round_loss = 0
            for layer in adaround_layers(self.o):
                if layer.split != 0:
                    raise NotImplementedError("split weight quantizers (LDM ResBlock skip-concat) do not occur in the "
                                              "diffusers SD / SDXL UNets")
                round_loss = round_loss + self.w * round_loss_of(layer.wqtizer, b)
        else:
            raise NotImplementedError
        total_loss = rec_loss + round_loss
        if self.record:
            self.history.append((self.count, float(total_loss.detach()), float(rec_loss.detach()),
                                 float(round_loss.detach()) if torch.is_tensor(round_loss) else float(round_loss), float(b)))
        if self.count % print_freq == 0:
            logger.info("Total loss:\t%.8f (rec:%.8f, round:%.8f)\tb=%.2f\tcount=%d", float(total_loss), float(rec_loss),
                        float(round_loss), b, self.count)
        return total_loss
