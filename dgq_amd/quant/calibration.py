"""``load_cali_model`` — the ``cali_ckpt`` reader (reference: quant/calibration.py:208-327; format SURVEY.md §5.4).

plus ``cali_model`` — the weight-PTQ driver (reference: quant/calibration.py:100-206).  The activation
producer is calibration_group_quantization.py.  Differences in *how* (not what):
  * weight quantizers are initialised by a vectorised per-channel min/max instead of a dummy forward with a
    python loop over channels (same values, calibration.py:224-225 -> quant_layer.py:253-264);
  * every ``act_<slot>`` table is planned and made device-resident once; the per-call reload of
    calibration.py:297-312 becomes an index flip (QuantModel.forward).
"""
import logging
from typing import Tuple

import torch
import torch.nn as nn

from .adaptive_rounding import AdaRoundQuantizer, RMODE
from .quant_block import BaseQuantBlock
from .quant_layer import QuantLayer, UniformAffineQuantizer
from .quant_model import QuantModel

logger = logging.getLogger(__name__)


def uaq2adar(model: nn.Module):
    """Wrap every weight quantizer into an AdaRoundQuantizer (calibration.py:20-43)."""
    for m in model.modules():
        if isinstance(m, QuantLayer) and not m.ignore_recon and not isinstance(m.wqtizer, AdaRoundQuantizer):
            m.wqtizer = AdaRoundQuantizer(m.wqtizer, rmode=RMODE.LEARNED_HARD_SIGMOID, w=m.original_w.data)


def _as_param(v):
    if isinstance(v, nn.Parameter):
        return v
    if not torch.is_tensor(v):
        v = torch.tensor(float(v))
    return nn.Parameter(v.detach().clone(), requires_grad=False)


def load_act_ckpt_with_difference_shape(ckpt, qnn, slot=None):
    """calibration.py:268-291.  For each QuantLayer whose ckpt δ is not the live scalar shape switch on
    ``use_group_num`` (this is what sends convs down the unfold path in the reference); then assign δ / z of every
    module that has ``<name>.delta`` in the ckpt — only if the live quantizer is initialised, like the
    reference (``module.delta is not None``)."""
    non_loaded = list(ckpt.keys())
    for name, module in qnn.named_modules():
        kd = "%s.aqtizer.delta" % name
        if kd in ckpt and isinstance(module, QuantLayer) and not module.use_group_num:
            live = module.aqtizer.delta
            if live is None or tuple(live.shape) != tuple(ckpt[kd].shape):
                module.use_group_num = True
        if "%s.delta" % name in ckpt:
            d, z = ckpt["%s.delta" % name], ckpt["%s.zero_point" % name]
            if getattr(module, "delta", None) is not None:
                module.delta.data = d.to(module.delta.device)
                if getattr(module, "zero_point", None) is not None and torch.is_tensor(module.zero_point):
                    module.zero_point.data = z.to(module.zero_point.device)
                non_loaded.remove("%s.delta" % name)
                non_loaded.remove("%s.zero_point" % name)
    if non_loaded:
        logger.info("keys not loaded: %s", non_loaded)
    return non_loaded


@torch.no_grad()
def load_cali_model(qnn: QuantModel, init_data: Tuple[torch.Tensor], use_aq: bool = False, path: str = None,
                    time_aware_aqtizer: bool = False, num_inference_steps: int = 25, use_group: bool = False,
                    init_forward: bool = True) -> None:
    """Same signature as the reference plus ``init_forward``: with ``False`` the data-dependent scalar
    self-initialisation forward of calibration.py:256-257 is skipped (quantizers not covered by the ckpt
    then initialise on their first real input) — used by CPU-side tests of the loader logic."""
    logger.info("Loading calibration model...")
    try:                                          # zip-format files are memory-mapped: pages are read as tensors are used
        full = torch.load(path, map_location="cpu", mmap=True)
    except (RuntimeError, ValueError):            # legacy (non-zip) serialisation
        full = torch.load(path, map_location="cpu")
    ckpt = full["weight"] if "weight" in full else full
    ckpt = dict(ckpt)

    qnn.set_quant_state(use_wq=True, use_aq=False)
    for m in qnn.model.modules():                 # weight-quantizer self-init (calibration.py:224-225)
        if isinstance(m, QuantLayer) and not m.wqtizer.init:
            m.wqtizer.init_from(m.w.data)
    qnn.disable_out_quantization()
    if any("alpha" in k for k in ckpt):          # BRECQ/AdaRound checkpoints (calibration.py:227-230)
        uaq2adar(qnn)
    for name, module in qnn.model.named_modules():
        if "wqtizer" in name and isinstance(module, (UniformAffineQuantizer, AdaRoundQuantizer)):
            module.zero_point = _as_param(module.zero_point)
            module.delta = _as_param(module.delta)
    for k in [k for k in ckpt if "aqtizer" in k]:     # calibration.py:241-243
        del ckpt[k]
    if "model" in list(ckpt.keys())[0]:
        missing = qnn.load_state_dict(ckpt, strict=False)
    else:
        missing = qnn.model.load_state_dict(ckpt, strict=False)
    logger.info("keys not loaded: %s", missing)
    qnn.set_quant_state(use_wq=True, use_aq=False)

    if use_aq:
        qnn.set_quant_state(use_wq=True, use_aq=True)
        if init_forward:
            dev = qnn.device
            args = [t.to(dev) for t in init_data]
            _ = qnn(*args)                           # scalar self-init of every act quantizer (:256-257)
        else:
            for m in qnn.model.modules():             # placeholders so the shape-adaptive loader can assign
                if isinstance(m, UniformAffineQuantizer) and not m.channel_wise and m.delta is None and m.leaf_param:
                    m.delta = nn.Parameter(torch.tensor(1.0), requires_grad=False)
                    m.zero_point = torch.tensor(0.0)
                    m.init = True
                    m._placeholder = True
            qnn.model.conv_in.aqtizer.delta = None    # never initialised in the reference (disable_aq)
            qnn.model.conv_in.aqtizer.init = False
            qnn.model.conv_out.aqtizer.delta = None
            qnn.model.conv_out.aqtizer.init = False
        for module in qnn.model.modules():
            if isinstance(module, (UniformAffineQuantizer, AdaRoundQuantizer)) and module.delta is not None:
                module.zero_point = _as_param(module.zero_point)
            if isinstance(module, AdaRoundQuantizer):
                module.delta = _as_param(module.delta)

        if use_group:
            load_act_ckpt_with_difference_shape(full["act_0"], qnn)

        if time_aware_aqtizer:                        # TFMQ-style per-timestep tables (calibration.py:297-312)
            slots = sorted(int(k[4:]) for k in full if k.startswith("act_"))
            layers, attn = [], []
            for name, module in qnn.named_modules():
                if isinstance(module, QuantLayer):
                    kd = "%s.aqtizer.delta" % name
                    if kd in full["act_%d" % slots[0]] and module.aqtizer.delta is not None:
                        layers.append((name, module))
                elif isinstance(module, UniformAffineQuantizer) and ("%s.delta" % name) in full["act_%d" % slots[0]] \
                        and ".aqtizer." not in name + "." and module.delta is not None:
                    attn.append((name, module))
            dev = qnn.device
            attn_tables = []
            for name, module in attn:
                tab = {}
                for s in slots:
                    a = full["act_%d" % s]
                    tab[s] = (a["%s.delta" % name].to(dev), a["%s.zero_point" % name].to(dev))
                attn_tables.append((module, tab))
            for name, module in layers:
                for s in slots:
                    a = full["act_%d" % s]
                    d = a["%s.aqtizer.delta" % name]
                    if d.dim() > 0 and not module.use_group_num:
                        module.use_group_num = True
                    module.set_act_table(s, d, a["%s.aqtizer.zero_point" % name])
            qnn.time_aware = dict(num_inference_steps=num_inference_steps, slots=set(slots), attn=attn_tables)
            qnn.ckpt = full
        else:                                         # QDiff-style single table (calibration.py:313-325)
            act = full["act_0"] if "act_0" in full else full
            if "model" in list(act.keys())[0]:
                qnn.load_state_dict(act, strict=False)
            else:
                qnn.model.load_state_dict(act, strict=False)
        if not init_forward:
            # quantizers the ckpt does not cover go back to "uninitialised" (they self-initialise on their
            # first real input instead of on the reference's random dummy batch)
            covered = full.get("act_0", {}) if isinstance(full, dict) else {}
            for name, m in qnn.named_modules():
                if getattr(m, "_placeholder", False):
                    if ("%s.delta" % name) not in covered:
                        m.delta, m.zero_point, m.init = None, None, False
                    del m._placeholder
    qnn._drop_graphs()                               # anything captured before the (re)load is stale
    logger.info("Loading calibration model done.")


def recon_targets(model: nn.Module, prev_name: str = "unet", tib_recon: bool = False):
    """The reconstruction schedule of ``cali_model`` (calibration.py:113-141) as a list of (kind, dotted name, module,
    keep_gpu): depth-first over ``named_children``; a QuantLayer that is not inside a quant block is reconstructed alone
    (``'layer'``), a quant block as a unit (``'block'``); targets flagged ``ignore_recon`` are skipped.  ``keep_gpu`` follows
    the reference's rule: cached tensors stay on the device only under ``down_blocks``."""
    out = []
    for name, module in model.named_children():
        if name == "tib":
            continue
        if name == "time_embedding" and tib_recon:
            raise NotImplementedError("tib_recon (TFMQ time-information block)")
        keep_gpu = "down_blocks" in prev_name
        if isinstance(module, QuantLayer):
            if not module.ignore_recon:
                out.append(("layer", "%s.%s" % (prev_name, name), module, keep_gpu))
        elif isinstance(module, BaseQuantBlock):
            if not module.ignore_recon:
                out.append(("block", "%s.%s" % (prev_name, name), module, keep_gpu))
        else:
            out += recon_targets(module, "%s.%s" % (prev_name, name), tib_recon)
    return out


def cali_model(qnn: QuantModel, w_cali_data: Tuple[torch.Tensor], a_cali_data: Tuple[torch.Tensor] = None, use_aq: bool = False,
               path: str = None, running_stat: bool = False, interval: int = 128, tib_recon: bool = False, **kwargs) -> dict:
    """Weight PTQ driver (calibration.py:100-206): (1) weight-quantizer initialisation by one weight-only forward,
    (2) BRECQ / AdaRound reconstruction of every target in network order — or a resume from ``resume_w`` — and (3) the
    ``<path>_weight_only`` checkpoint ``{'weight': state_dict}`` that ``load_cali_model`` and the activation producer
    (calibration_group_quantization.act_group_quant) consume.  Returns the saved dict.

    kwargs as the reference passes them (src/quantize_weight.py:192-207): iters, batch_size, w, asym, warmup, opt_mode,
    multi_gpu, no_recon, resume_w, plus anything layer_ / block_reconstruction accept.  ``use_aq`` appends QDiff-style scalar
    activation calibration (``cali_model_aq``, calibration.py:45-97) and saves ``path``; DGQ's own recipe calibrates activations
    with act_group_quant instead."""
    from .reconstruction import block_reconstruction, layer_reconstruction
    import os
    if tib_recon:
        raise NotImplementedError("tib_recon (TFMQ time-information block)")
    kwargs = dict(kwargs)
    resume_w = kwargs.pop("resume_w", None)
    no_recon = kwargs.pop("no_recon", False)
    logger.info("weight initialization...")
    dev = qnn.device
    qnn.set_quant_state(use_wq=True, use_aq=False)
    with torch.no_grad():
        qnn(*(x[:min(1, x.shape[0])].to(dev) for x in w_cali_data))      # every wqtizer initialises on its weight
    qnn.disable_out_quantization()
    if resume_w:
        # calibration.py:151-172: the stored weight quantizers replace reconstruction; nothing is written for the weights, and the
        # branch falls through to the ``use_aq`` tail like the reference's does.
        load_cali_model(qnn, init_data=None, use_aq=False, path=resume_w)
        model_dict = {"weight": torch.load(resume_w, map_location="cpu")["weight"]}
        logger.info("quantized model loaded from %s", resume_w)
    else:
        if not no_recon:
            for kind, name, module, keep_gpu in recon_targets(qnn, "unet"):
                logger.info("Reconstruction for %s %s", kind, name)
                fn = layer_reconstruction if kind == "layer" else block_reconstruction
                fn(qnn, module, cali_data=w_cali_data, **dict(kwargs, keep_gpu=keep_gpu))
        qnn.set_quant_state(use_wq=True, use_aq=False)
        for name, module in qnn.model.named_modules():
            if "wqtizer" in name and isinstance(module, (UniformAffineQuantizer, AdaRoundQuantizer)):
                module.zero_point = _as_param(module.zero_point)
                module.delta = _as_param(module.delta)
        state = {k: v.detach().cpu().clone() for k, v in qnn.state_dict().items()}
        model_dict = {"weight": state}
        if path is not None:
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            torch.save(model_dict, "%s_weight_only" % path)
            logger.info("calibrated model saved to %s_weight_only", path)
    if use_aq:                                              # calibration.py:199-206: + one scalar (δ, z) table per interval
        model_dict = cali_model_aq(qnn, a_cali_data, model_dict, running_stat, interval)
        if path is not None:
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            torch.save(model_dict, path)
            logger.info("calibrated model saved to %s", path)
    return model_dict


@torch.no_grad()
def cali_model_aq(qnn: QuantModel, a_cali_data, model_dict, running_stat, interval):
    """QDiff-style scalar activation calibration (calibration.py:45-97), the ``--use_aq`` tail of the weight CLI: per interval
    of the calibration data every activation quantizer forgets its state, initialises itself on one random batch of <= 8 samples
    (its ``Scaler`` on the whole tensor) and — with ``running_stat`` — follows the remaining batches with the EMA of
    ``act_momentum_update``; the (δ, z) pairs go to ``act_<interval>`` under the reference's key names.  DGQ's own recipe
    replaces this by ``calibration_group_quantization.act_group_quant``."""
    import numpy as np
    from .calibration_group_quantization import collect_act_state
    dev = qnn.device
    qnn.eval()
    for time in range(a_cali_data[0].shape[0] // interval):
        t_cali = tuple(x[time * interval: (time + 1) * interval] for x in a_cali_data)
        qnn.set_quant_state(use_wq=True, use_aq=True)
        for name, module in qnn.model.named_modules():
            if "aqtizer" in name and hasattr(module, "init"):
                del module.delta
                module.delta = None
                if isinstance(module, UniformAffineQuantizer):
                    del module.zero_point
                    module.zero_point = None
                module.init = False
        batch_size = min(8, t_cali[0].shape[0])
        inds = np.random.choice(t_cali[0].shape[0], batch_size, replace=False)
        _ = qnn(*(x[inds].to(dev) for x in t_cali))
        if running_stat:
            logger.info("running stat for activation calibration...")
            inds = np.arange(t_cali[0].shape[0])
            np.random.shuffle(inds)
            qnn.set_running_stat(True)
            try:
                for i in range(0, t_cali[0].shape[0], batch_size):
                    _ = qnn(*(x[inds[i: i + batch_size]].to(dev) for x in t_cali))
            finally:
                qnn.set_running_stat(False)
        for name, module in qnn.model.named_modules():
            if "aqtizer" in name and isinstance(module, UniformAffineQuantizer) and module.delta is not None:
                module.zero_point = _as_param(module.zero_point)
        model_dict["act_%d" % time] = collect_act_state(qnn)
    return model_dict
