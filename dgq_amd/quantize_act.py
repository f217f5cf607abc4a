"""Drop-in counterpart of the reference CLI ``src/quantize_act.py`` (same flag names, :35-69): DGQ activation calibration —
per-timestep statistics + K-Means grouping — on top of a weight-only checkpoint, writing
``<outdir>/<time>/cali_ckpt_activation_w?a?g?.pth`` with the reference's key names (``act_<slot>``).  ``merge`` below is
``results/merge.py``: it adds the 'weight' entry and writes ``<act ckpt>_merged``, the file ``get_qmodel`` loads.

Pipeline construction and calibration-data sampling are replaced as in dgq_amd/quantize_weight.py; the rest is the reference's
flow (:121-163): quantizer dicts, ``QuantModel``, ``load_cali_model`` of the weight-only file on one dummy sample,
``act_group_quant``."""
import argparse
import logging
import os
import sys

import torch

from .quantize_weight import MODEL_TYPE, build_pipe, setup, str2bool


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="DGQ activation calibration on MI355X")
    p.add_argument("--outdir", type=str, default="results")
    p.add_argument("--weight_only_ckpt", type=str, default=None)
    p.add_argument("--wq", type=int, default=4)
    p.add_argument("--aq", type=int, default=8)
    p.add_argument("--softmax_a_bit", type=int, default=8)
    p.add_argument("--time_aware_aqtizer", type=str2bool)
    p.add_argument("--t2i_log_quant", type=str2bool)
    p.add_argument("--t2i_real_time", type=str2bool)
    p.add_argument("--t2i_start_peak", type=str2bool)
    p.add_argument("--group_num", type=int, default=1)
    p.add_argument("--group_mode", type=str, choices=["mean", "minmax", "test"], default="minmax")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--coco_path", type=str, default=None)
    p.add_argument("--cali_prompt_data_n", default=64, type=int)
    p.add_argument("--cali_data_path", type=str, default="./data/cali_data")
    p.add_argument("--cali_data_size", type=int, default=-1)
    p.add_argument("--step_size", type=int, default=25)
    # additions (no counterpart in the reference)
    p.add_argument("--model_type", default=MODEL_TYPE, choices=["sd", "sdxl", "tiny", "mini"])
    p.add_argument("--unet_weights", default=None, help="HF-keyed UNet state-dict (.pt); synthetic if omitted")
    p.add_argument("--merge", action="store_true", help="also write <act ckpt>_merged (results/merge.py)")
    return p.parse_args(argv)


def merge(weight_path, act_path):
    """results/merge.py:13-18."""
    ck = torch.load(act_path)
    ck["weight"] = torch.load(weight_path)["weight"]
    torch.save(ck, act_path + "_merged")
    return act_path + "_merged"


def main(argv=None):
    opt = parse_args(argv)
    import numpy as np
    from .dataset_generation import calibration_data_generation
    from .quant import QuantModel, Scaler, QMODE, load_cali_model, act_group_quant
    from .quant.load_qmodel_util import setup_pipe_to_calibrate
    mt = opt.model_type
    pipe = build_pipe(opt)
    outpath, logger = setup(opt.seed, opt.outdir)
    np.random.seed(opt.seed)                                   # seed_everything: K-Means draws from numpy's global state
    logger.info("sys.argv: %s", sys.argv)
    w_cali_data, a_cali_data, interval = calibration_data_generation(
        mt, pipe=pipe, cali_data_path=opt.cali_data_path, coco_path=opt.coco_path, cali_prompt_data_n=opt.cali_prompt_data_n,
        step_size=opt.step_size, time_aware_aqtizer=opt.time_aware_aqtizer, cali_data_size=opt.cali_data_size)
    wq_params = {"bits": opt.wq, "channel_wise": True, "scaler": Scaler.MINMAX}
    aq_params = {"bits": opt.aq, "channel_wise": False, "scaler": Scaler.MINMAX, "leaf_param": True}
    softmax_aq_params = {"softmax_a_bit": opt.softmax_a_bit, "t2i_log_quant": opt.t2i_log_quant, "t2i_real_time": opt.t2i_real_time,
                         "t2i_start_peak": opt.t2i_start_peak, "log_max_1": False}
    setup_pipe_to_calibrate(mt, pipe)
    qnn = QuantModel(model=pipe.unet, wq_params=wq_params, aq_params=aq_params, softmax_aq_params=softmax_aq_params,
                     aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value], tib_recon=False).to("cuda").eval()
    dummy = tuple(d[0:1] for d in w_cali_data)
    load_cali_model(qnn, init_data=dummy, use_aq=False, path=opt.weight_only_ckpt)
    path = os.path.join(outpath, "cali_ckpt_activation_w%da%dg%d.pth" % (opt.wq, opt.aq, opt.group_num))
    act_group_quant("sdxl" if mt == "sdxl" else "sd", qnn, a_cali_data=a_cali_data, path=path, group_num=opt.group_num, group_mode=opt.group_mode, interval=interval)
    logger.info("Activation quantization is done")
    if opt.merge:
        logger.info("merged: %s", merge(opt.weight_only_ckpt, path))
    return path


if __name__ == "__main__":
    main()
