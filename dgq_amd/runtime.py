"""Denoise loop + synthetic-model builder + one-process-per-GPU sharding helpers.

Multi-GPU: the denoising batch (prompt × CFG pair) shards embarrassingly — each rank owns a
slice of the prompt list (the reference's only multi-GPU mode: ``prompts[rank::world]``-style slicing in
src/gen4eval_SDXL.py:116, one python process per GPU, scripts/gen4eval_SDXL.sh:53-100), weights and quantizer
tables are replicated read-only, and NO collective sits on the data path."""
import os
import types

import torch

from . import synth
from .scheduler import DDIMScheduler


def shard_prompts(n_prompts: int, rank: int, world_size: int):
    """Contiguous rank slice of the prompt indices; the CFG halves of a prompt stay on one GPU."""
    per = (n_prompts + world_size - 1) // world_size
    lo = min(rank * per, n_prompts)
    return list(range(lo, min(lo + per, n_prompts)))


def slot_for_timestep(t: int, num_inference_steps: int) -> int:
    """calibration.py:301-304."""
    return int((1000 - int(t)) // (1000 // num_inference_steps))


@torch.no_grad()
def denoise_loop(unet_fn, latents, ctx_pair, num_inference_steps, guidance=7.5, timesteps=None, extra=None):
    """CFG denoise loop: one UNet call per step on the (uncond ‖ cond) pair, DDIM update on the device."""
    sch = DDIMScheduler(num_inference_steps)
    x = latents
    for t in (timesteps if timesteps is not None else sch.timesteps):
        inp = torch.cat([x, x], dim=0) if guidance > 0 else x
        eps = unet_fn(inp, t, ctx_pair, **(extra or {}))
        if guidance > 0:
            e_u, e_c = eps.chunk(2)
            eps = e_u + guidance * (e_c - e_u)
        x = sch.step(eps, t, x)
    return x


def quant_params(Scaler, wbits, abits, use_aq, log, rt, sp):
    wq = {"bits": wbits, "channel_wise": True, "scaler": Scaler.MINMAX}
    aq = {"bits": abits, "channel_wise": False, "scaler": Scaler.MINMAX, "leaf_param": use_aq}
    sm = {"softmax_a_bit": abits, "t2i_log_quant": log, "t2i_real_time": rt, "t2i_start_peak": sp, "log_max_1": False}
    return wq, aq, sm


def synthetic_fp_unet(arch, cfg, res, batch, slots, ckpt_dir="/tmp", seed=0, rank=0, barrier=None):
    """The rank-shared half of ``build_synthetic_qnn``: rank 0 writes (once) the synthetic reference-format cali_ckpt, every rank
    waits for it at ``barrier`` and reads the FP weights back from the file memory-mapped.  Returns (FP UNet, ckpt path, laps)."""
    import time
    from .diffusers_rewrite import UNet2DConditionModel
    verbose = os.environ.get("DGQ_BUILD_TIMING") == "1"
    t_last = [time.time()]
    laps = {}

    def lap(what):
        now = time.time()
        laps[what] = round(now - t_last[0], 2)
        if verbose:
            print("[build %s] %-28s %.1f s" % (arch, what, now - t_last[0]), flush=True)
        t_last[0] = now
    path = os.path.join(ckpt_dir, "dgq_synth_%s_w%da%dg%d_r%d_b%d_s%s_%s.pth" % (
        arch, cfg["wbits"], cfg["abits"], cfg["G"], res, batch,
        ("%d" % slots) if isinstance(slots, int) else "x".join(str(s) for s in synth.slot_list(slots)),
        "sp" if cfg["sp"] else "nosp"))
    if rank == 0 and not os.path.exists(path):
        tmp = path + ".tmp%d" % os.getpid()
        synth.write_cali_ckpt(tmp, arch, cfg["wbits"], cfg["abits"], cfg["G"], num_slots=slots, seed=seed, batch=batch,
                              res=res, start_peak=cfg["sp"], uniform_softmax=(cfg["use_aq"] and not cfg["log"]),
                              with_act=cfg["use_aq"])
        os.replace(tmp, path)
    lap("write cali_ckpt")
    if barrier is not None:
        barrier()
    unet = UNet2DConditionModel(arch)
    lap("construct FP UNet")
    # FP weights (conv_in / conv_out keep their constructor-time copies): read back from the ckpt file
    # (memory-mapped) — the same name-keyed tensors rank 0 generated for it, without regenerating them on every rank
    unet.load_state_dict(synth.state_dict_from_ckpt(path))
    lap("FP weights from the ckpt (mmap)")
    return unet, path, laps, lap


def build_synthetic_qnn(arch, cfg, res, batch, slots, ckpt_dir="/tmp", seed=0, device="cuda", rank=0, barrier=None):
    """Writes (once) a synthetic reference-format cali_ckpt and builds the QuantModel from it through the same
    entry point the reference's CLI uses (get_qmodel, src/inference_qmodel.py:91).
    cfg keys: wbits, abits, use_aq, G, log, rt, sp, time_aware, steps.  ``slots``: how many act_<s> tables to write
    (int) or the explicit slot ids (a test that visits t = 981 and t = 21 needs act_0 and act_48, not 49 tables)."""
    from .quant import get_qmodel, Scaler
    unet, path, laps, lap = synthetic_fp_unet(arch, cfg, res, batch, slots, ckpt_dir, seed, rank, barrier)
    pipe = types.SimpleNamespace(unet=unet)
    wq, aq, sm = quant_params(Scaler, cfg["wbits"], cfg["abits"], cfg["use_aq"], cfg["log"], cfg["rt"], cfg["sp"])
    qnn = get_qmodel(arch, pipe, path, wq, cfg["use_aq"], aq, sm, cfg["G"] > 1, cfg["steps"],
                     cfg["time_aware"] and cfg["use_aq"], device=device)
    lap("get_qmodel (wrap + load)")
    qnn.float()
    qnn = qnn.to(device)
    qnn.disable_out_quantization()
    qnn._build_laps = laps                   # seconds per build phase (bench.py reports them as "load")
    return qnn, path
