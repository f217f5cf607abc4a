"""Weight PTQ by output reconstruction — BRECQ blocks / AdaRound rounding (reference:
quant/reconstruction.py:13-200).

For one target (a ``QuantLayer`` or a quant block) the loop learns the rounding direction α of every weight of the target
so that the target's output on the calibration set — inputs optionally produced by the already-quantised prefix of the
network (``asym``) — matches the FP output: Adam on α, loss = ``LossFunc`` (reconstruction + annealed rounding
regulariser).  (The reference's ``use_aq`` variant — LSQ-style tuning of activation step sizes — belongs to QDiff's
scalar activation calibration, which DGQ replaces by group calibration; it raises here.)

Execution model here: the target runs through torch autograd on the device (rocBLAS / MIOpen for the FP contractions of
a training loop); the DGQ-specific ops — soft-rounded weights and the regulariser, forward and backward — are the fused
HIP kernels of csrc/adaround.hip.  One mini-batch index vector per iteration is drawn with ``torch.randperm`` exactly as
the reference does, so a seeded run visits the same samples.
"""
import logging
from typing import Tuple, Union

import torch

from .adaptive_rounding import AdaRoundQuantizer, RMODE
from .data_utill import save_grad, save_inout
from .quant_block import BaseQuantBlock
from .quant_layer import QuantLayer, StraightThrough
from .quant_model import QuantModel
from .reconstruction_util import RLOSS, LossFunc, adaround_layers

logger = logging.getLogger(__name__)

#: test hook: called with (target, LossFunc) before the loop starts (tests switch on ``record`` to compare trajectories)
ON_LOSS_CREATED = None


def _to_adaround(layer: QuantLayer) -> torch.nn.Parameter:
    """Swap the layer's weight quantizer for an AdaRoundQuantizer in soft mode; returns its α."""
    if layer.split != 0:
        raise NotImplementedError("split weight quantizers (LDM ResBlock skip-concat) do not occur in the diffusers UNets")
    if not layer.wqtizer.init:
        layer.wqtizer.init_from(layer.w.data)                     # reference: done by the weight-initialisation forward
    layer.wqtizer = AdaRoundQuantizer(uaqtizer=layer.wqtizer, rmode=RMODE.LEARNED_HARD_SIGMOID, w=layer.original_w.data)
    layer.wqtizer.soft_tgt = True
    return layer.wqtizer.alpha


def _reconstruct(model: QuantModel, target: Union[QuantLayer, BaseQuantBlock], cali_data: Tuple[torch.Tensor], batch_size: int,
                 iters: int, w: float, opt_mode: RLOSS, asym: bool, include_act_func: bool, b_range: tuple, warmup: float,
                 use_aq: bool, lr: float, p: float, keep_gpu: bool) -> None:
    model.set_quant_state(use_wq=False, use_aq=False)
    target.set_quant_state(use_wq=True, use_aq=use_aq)
    org_act_func = None
    if not include_act_func:
        org_act_func, target.act_func = target.act_func, StraightThrough()

    layers = adaround_layers(target) if isinstance(target, QuantLayer) else \
        [m for m in target.modules() if isinstance(m, QuantLayer) and not m.quant_emb]
    if use_aq:
        # QDiff's LSQ-style tuning of the activation step sizes (reconstruction.py:44-47,139-161) needs a differentiable
        # activation fake-quantiser; DGQ replaces that stage by the group calibration (act_group_quant), so it is not built
        raise NotImplementedError("use_aq reconstruction (activation step-size tuning) is not part of the DGQ recipe")
    opt_params = [_to_adaround(m) for m in layers]
    if not opt_params:
        return
    optimizer = torch.optim.Adam(opt_params)
    scheduler = None
    loss_func = LossFunc(o=target, round_loss=RLOSS.NONE if use_aq else RLOSS.RELAXATION, w=w, max_count=iters,
                         rec_loss=opt_mode, b_range=b_range, decay_start=0.0, warmup=warmup, p=p)
    if ON_LOSS_CREATED is not None:
        ON_LOSS_CREATED(target, loss_func)
    cached_inputs, cached_outputs = save_inout(model, target, cali_data, asym, use_aq, batch_size, keep_gpu)
    cached_grads = save_grad(model, target, cali_data, asym, use_aq, batch_size, keep_gpu) if opt_mode != RLOSS.MSE else None
    device = next(target.parameters()).device
    n = cached_inputs[0].size(0)
    for _ in range(iters):
        idx = torch.randperm(n)[:batch_size]                      # host generator: the reference's sample sequence
        didx = idx.to(cached_inputs[0].device)
        cur_inputs = tuple(x[didx].to(device) for x in cached_inputs)
        cur_outputs = cached_outputs[didx].to(device)
        cur_grads = cached_grads[idx.to(cached_grads.device)].to(device) if cached_grads is not None else None
        optimizer.zero_grad()
        out_quant = target(*cur_inputs)
        err = loss_func(out_quant, cur_outputs, cur_grads)
        err.backward()
        optimizer.step()
        if scheduler:
            scheduler.step()
    for m in layers:
        if isinstance(m.wqtizer, AdaRoundQuantizer):
            m.wqtizer.soft_tgt = False                             # hard rounding from here on: (α >= 0)
    if org_act_func is not None:
        target.act_func = org_act_func


def layer_reconstruction(model: QuantModel, layer: QuantLayer, cali_data: Tuple[torch.Tensor], batch_size: int = 128,
                         iters: int = 20000, w: float = 0.001, opt_mode: RLOSS = RLOSS.MSE, asym: bool = False,
                         include_act_func: bool = True, b_range: tuple = (20, 2), warmup: float = 0.0, use_aq: bool = False,
                         lr: float = 4e-5, p: float = 2.0, multi_gpu: bool = False, keep_gpu=True, **kwargs) -> None:
    """reconstruction.py:13-87 — one layer (first / last convs, proj_in / proj_out, samplers, time embedding)."""
    if multi_gpu:
        raise NotImplementedError("gradient all-reduce across calibration replicas (linklink) — the reference disables it too "
                                  "(src/quantize_weight.py:213-214)")
    _reconstruct(model, layer, cali_data, batch_size, iters, w, opt_mode, asym, include_act_func, b_range, warmup, use_aq, lr, p,
                 keep_gpu)


def block_reconstruction(model: QuantModel, block: BaseQuantBlock, cali_data: Tuple[torch.Tensor], batch_size: int = 32,
                         iters: int = 20000, w: float = 0.01, opt_mode: RLOSS = RLOSS.MSE, asym: bool = False,
                         include_act_func: bool = True, b_range: tuple = (20, 2), warmup: float = 0.0, use_aq: bool = False,
                         lr: float = 4e-5, p: float = 2.0, multi_gpu: bool = False, keep_gpu=True, **kwargs) -> None:
    """reconstruction.py:91-200 — one QuantResnetBlock2D / QuantBasicTransformerBlock, all its layers jointly."""
    if multi_gpu:
        raise NotImplementedError("gradient all-reduce across calibration replicas (linklink) — the reference disables it too "
                                  "(src/quantize_weight.py:213-214)")
    _reconstruct(model, block, cali_data, batch_size, iters, w, opt_mode, asym, include_act_func, b_range, warmup, use_aq, lr, p,
                 keep_gpu)


def tib_reconstruction(*args, **kwargs) -> None:
    raise NotImplementedError("TFMQ temporal-information-block reconstruction (reconstruction.py:203-307) needs "
                              "QuantModel(tib_recon=True); the DGQ recipes run with --tib_recon False")
