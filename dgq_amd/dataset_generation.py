"""Calibration-data side of the two producer CLIs (``src/dataset_generation.py`` in the reference).

Generating calibration data means running a full diffusers pipeline over COCO prompts (``collect_data``,
dataset_generation.py:17-58: tokenizer, CLIP, scheduler, pretrained weights) — out of scope here.  What the
producers need from it is restated: ``cali_data_preprocessing`` (:60-157) turns the dict the pipeline callback recorded into
the ``(latent_model_input, timesteps, prompt_embeds[, add_text_embeds, add_time_ids])`` tuple the calibration loops iterate,
and ``calibration_data_generation`` (:160-200) loads it, picks ``interval`` and moves everything to fp32 on the CPU.  Without a
recorded file the CLIs fall back on synthetic tensors of the same shapes (and say so)."""
import logging
import os

import torch

logger = logging.getLogger(__name__)


def cali_data_preprocessing(model_type, cali_data, cali_data_size, step_size, n_prompts):
    """dataset_generation.py:60-157.  ``cali_data``: dict of per-callback lists, prompt batch major (for each prompt batch the
    T consecutive denoise steps): "latents", "timesteps", "prompt_embeds", "latent_model_input" (+ the SDXL conditioning).
    Returns (tuple rearranged timestep-major and concatenated, interval = samples per timestep).  T = step_size for SDXL,
    step_size + 1 for SD (PNDM calls the UNet once more).  The debug sub-sampling branch (0 < cali_data_size <= 12) raises in
    the reference and does here."""
    T = step_size if model_type == "sdxl" else step_size + 1
    if 0 < cali_data_size <= 12:
        raise NotImplementedError("Only for debug")
    re = {k: [] for k in cali_data}
    for i in range(T):
        for j in range(0, len(cali_data["latents"]), T):
            for k in cali_data:
                re[k].append(cali_data[k][j + i])
    cfg = re["prompt_embeds"][0].shape[0] == 2 * re["latents"][0].shape[0]      # CFG: the UNet saw the doubled batch
    timesteps = [t.repeat(2) for t in re["timesteps"]] if cfg else re["timesteps"]
    out = (torch.cat(re["latent_model_input"], dim=0), torch.cat(timesteps, dim=0), torch.cat(re["prompt_embeds"], dim=0))
    if model_type == "sdxl":
        out = out + (torch.cat(re["add_text_embeds"], dim=0), torch.cat(re["add_time_ids"], dim=0))
    return out, n_prompts * (2 if cfg else 1)


def synthetic_cali_data(model_type, n_timesteps, per_timestep, seed=0):
    """Stand-in for a recorded file: ``per_timestep`` samples at each of ``n_timesteps`` evenly spaced timesteps, descending."""
    from . import synth
    from .diffusers_rewrite import ARCH
    a = ARCH[model_type]
    n = n_timesteps * per_timestep
    res = a["sample_size"]
    ts = torch.tensor([999 - (i // per_timestep) * (1000 // n_timesteps) for i in range(n)], dtype=torch.int64)
    out = (synth.named_randn("cali_x", (n, 4, res, res), seed + 1), ts, synth.named_randn("cali_ctx", (n, 77, a["ctx_dim"]), seed + 2))
    if model_type == "sdxl":
        inp = synth.synth_inputs("sdxl", n, seed + 3, res)
        out = out + (inp["text_embeds"], inp["time_ids"])
    return out


def calibration_data_generation(model_type, pipe=None, cali_data_path=None, coco_path=None, cali_prompt_data_n=64, step_size=25,
                                time_aware_aqtizer=True, cali_data_size=-1, synthetic=(2, 4)):
    """dataset_generation.py:160-200: (w_cali_data, a_cali_data, interval).  ``cali_data_path`` may hold the recorded dict
    (the reference's own file) or an already preprocessed tuple; a missing file means synthetic data of ``synthetic`` =
    (timesteps, samples per timestep)."""
    if cali_data_path and os.path.exists(cali_data_path):
        raw = torch.load(cali_data_path, map_location="cpu")
        if isinstance(raw, dict):
            data, interval = cali_data_preprocessing(model_type, raw, cali_data_size, step_size, cali_prompt_data_n)
        else:
            data = tuple(raw)
            t = data[1]
            interval = int((t == t[0]).sum())              # samples of the first timestep (timestep-major order)
    else:
        logger.warning("no calibration data at %r: synthetic tensors (%d timesteps x %d samples) — collecting real data needs "
                       "the diffusers pipeline, which is not part of this package", cali_data_path, *synthetic)
        data = synthetic_cali_data(model_type, *synthetic)
        interval = synthetic[1]
    if not time_aware_aqtizer:
        interval = data[0].shape[0]
    data = tuple(x.to("cpu").float() if x.is_floating_point() else x.to("cpu") for x in data)
    return data, data, interval
