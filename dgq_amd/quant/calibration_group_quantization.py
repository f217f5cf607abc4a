"""DGQ activation calibration — the PRODUCER of the ``act_<t>`` tables this package consumes
(reference: quant/calibration_group_quantization.py:44-129, driven by src/quantize_act.py:100-164).

Per timestep interval of the calibration set: reset every activation quantizer, one forward to self-initialise the
scalar scales, ``set_group_num(G)``, forwards over the interval's samples recording per-axis min / max in every
quantizer (a HIP reduction, dgq_minmax_rows_cols), ``done_group_num`` (axis choice by spread, K-Means(G,
random_state=0) on the host, per-cluster ranges), then the (δ, z) pairs are collected under the reference's key names.
Same function names and arguments as the reference; the forwards run on the integer path of this package."""
import logging
import os
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn

from .quant_layer import UniformAffineQuantizer
from .quant_model import QuantModel

logger = logging.getLogger(__name__)


def collect_act_state(qnn: QuantModel):
    """``model.<path>.delta`` / ``.zero_point`` of every activation-side quantizer that owns both
    (calibration_group_quantization.py:99-104; real-time log quantizers have no δ and are skipped like there)."""
    out = {}
    for name, module in qnn.model.named_modules():
        if "aqtizer" in name and isinstance(module, UniformAffineQuantizer) and module.delta is not None \
                and module.zero_point is not None:
            out["model." + name + ".delta"] = module.delta.detach().cpu().clone()
            out["model." + name + ".zero_point"] = torch.as_tensor(module.zero_point).detach().cpu().clone()
    return out


@torch.no_grad()
def cali_model_aq(model_type, qnn: QuantModel, a_cali_data, model_dict, group_num, interval, group_mode):
    dev = qnn.device
    qnn.eval()
    cali_data = a_cali_data
    for time in range(cali_data[0].shape[0] // interval):
        t_cali_data = tuple(x[time * interval: (time + 1) * interval] for x in cali_data)
        qnn.set_quant_state(use_wq=True, use_aq=True)
        qnn.disable_out_quantization()
        for name, module in qnn.model.named_modules():      # forget the previous interval (:57-68)
            if "aqtizer" in name and hasattr(module, "init"):
                del module.delta                             # parameter or plain tensor: drop it, then a plain None
                module.delta = None
                if isinstance(module, UniformAffineQuantizer):
                    del module.zero_point
                    module.zero_point = None
                module.init = False
        if model_type in ("sd", "tiny"):
            batch_size = min(8, t_cali_data[0].shape[0])
        elif model_type == "sdxl":
            batch_size = min(4, t_cali_data[0].shape[0])
        else:
            raise ValueError(f"Unknown model type: {model_type}")
        inds = np.random.choice(t_cali_data[0].shape[0], batch_size, replace=False)
        _ = qnn(*(x[inds].to(dev) for x in t_cali_data))     # scalar self-initialisation of every quantizer (:83-85)
        logger.info("group_num: %d running stat for activation calibration...", group_num)
        inds = np.arange(t_cali_data[0].shape[0])
        np.random.shuffle(inds)
        qnn.set_group_num(group_num)
        try:
            for i in range(0, t_cali_data[0].shape[0], batch_size):
                _ = qnn(*(x[inds[i: i + batch_size]].to(dev) for x in t_cali_data))
        except BaseException:
            qnn.restore_fusion()                              # a failed calibration forward must not leave the process unfused
            raise
        qnn.done_group_num(group_num, mode=group_mode)
        for name, module in qnn.model.named_modules():      # zero points become parameters like δ (:92-98)
            if "aqtizer" in name and isinstance(module, UniformAffineQuantizer) and module.delta is not None:
                zp = module.zero_point
                module.zero_point = nn.Parameter(zp if torch.is_tensor(zp) else torch.tensor(float(zp)), requires_grad=False)
        model_dict["act_{}".format(time)] = collect_act_state(qnn)
    return model_dict


def act_group_quant(model_type, qnn: QuantModel, a_cali_data: Tuple[torch.Tensor], path: str = None, group_num: int = 1,
                    interval: int = 128, group_mode="minmax", **kwargs) -> None:
    """Writes ``cali_ckpt_activation_*.pth`` = {'act_0': {...}, 'act_1': {...}, ...}; results/merge.py adds 'weight'."""
    logger.info("Calibrating...")
    model_dict = cali_model_aq(model_type, qnn, a_cali_data, {}, group_num, interval, group_mode=group_mode)
    if os.path.dirname(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(model_dict, path)
    logger.info("calibrated model saved to %s", path)
    return model_dict
