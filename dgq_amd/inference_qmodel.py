"""Drop-in counterpart of the reference CLI ``src/inference_qmodel.py`` (same flag names, :16-44).

The reference builds a diffusers pipeline from pretrained weights (src/utils.py:16-54: needs network + HF
checkpoints, out of scope) and renders PNGs.  No pretrained weights exist in this environment, so this CLI runs the
same quantized-UNet path — ``get_qmodel`` → ``half()/float()`` → ``disable_out_quantization()`` → denoise loop
(src/inference_qmodel.py:91-108) — on synthetic name-keyed weights and latents, and saves the final latents.
With ``--cali_ckpt`` pointing at a real merged checkpoint and ``--unet_weights`` at an HF UNet state-dict the very
same code path serves real weights.  It does not end in ``breakpoint()`` (:110).

Multi-GPU: run under ``python -m torch.distributed.run --nproc-per-node N``; every rank takes its slice of the
prompt list (src/gen4eval_SDXL.py:116), no collective on the data path.
"""
import argparse
import logging
import os
import types

import torch

from . import synth
from .runtime import denoise_loop, shard_prompts

MODEL_TYPE = os.environ.get("DIFFUSERS_REWRITE", "sd")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Quantized UNet inference (DGQ) on MI355X")
    p.add_argument("--use_group", action="store_true", help="Use group quantization")
    p.add_argument("--num_inference_steps", type=int, default=-1)
    p.add_argument("--prompt", type=str, default="a painting of a virus monster playing guitar")
    p.add_argument("--cali_ckpt", type=str, default=None, help="merged calibration checkpoint (synthetic if omitted)")
    p.add_argument("--fp16", action="store_true")
    p.add_argument("--wq", type=int, default=4)
    p.add_argument("--use_aq", action="store_true")
    p.add_argument("--aq", type=int, default=8)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--t2i_log_quant", action="store_true")
    p.add_argument("--t2i_real_time", action="store_true")
    p.add_argument("--t2i_start_peak", action="store_true")
    p.add_argument("--time_aware_aqtizer", action="store_true")
    # additions (no counterpart in the reference)
    p.add_argument("--model_type", default=MODEL_TYPE, choices=["sd", "sdxl", "tiny", "mini"])
    p.add_argument("--unet_weights", default=None, help="HF-keyed UNet state-dict (.pt); synthetic if omitted")
    p.add_argument("--group_num", type=int, default=16, help="G of the synthetic checkpoint")
    p.add_argument("--n_prompts", type=int, default=2, help="synthetic prompts (the reference renders 2 images)")
    p.add_argument("--out", default="latents_{rank}.pt")
    p.add_argument("--graphs", action="store_true", help="replay one hipGraph per timestep slot")
    return p.parse_args(argv)


def main(argv=None):
    opt = parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    from .diffusers_rewrite import UNet2DConditionModel, ARCH
    from .quant import get_qmodel, Scaler
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dev = torch.device("cuda", torch.cuda.current_device())
    mt = opt.model_type
    steps = opt.num_inference_steps if opt.num_inference_steps > 0 else (4 if mt == "sdxl" else 25)
    guidance = 0.0 if mt == "sdxl" else 7.5                     # src/inference_qmodel.py:48-51
    torch.manual_seed(opt.seed)

    unet = UNet2DConditionModel(mt)
    if opt.unet_weights:
        unet.load_state_dict(torch.load(opt.unet_weights, map_location="cpu"))
    else:
        synth.load_synth_weights(unet, mt, 0)
    pipe = types.SimpleNamespace(unet=unet)
    res = ARCH[mt]["sample_size"]
    batch = 2 if guidance > 0 else 1
    ckpt = opt.cali_ckpt
    if ckpt is None:
        ckpt = "/tmp/dgq_cli_%s_w%da%dg%d_s%d.pth" % (mt, opt.wq, opt.aq, opt.group_num if opt.use_group else 1, steps)
        if rank == 0 and not os.path.exists(ckpt):
            synth.write_cali_ckpt(ckpt, mt, opt.wq, opt.aq, opt.group_num if opt.use_group else 1,
                                  num_slots=steps if opt.time_aware_aqtizer else 1, seed=0, batch=batch, res=res,
                                  start_peak=opt.t2i_start_peak, uniform_softmax=opt.use_aq and not opt.t2i_log_quant,
                                  with_act=opt.use_aq)
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("nccl")
            dist.barrier()

    wq_params = {"bits": opt.wq, "channel_wise": True, "scaler": Scaler.MINMAX}
    aq_params = {"bits": opt.aq, "channel_wise": False, "scaler": Scaler.MINMAX, "leaf_param": opt.use_aq}
    softmax_aq_params = {"softmax_a_bit": opt.aq, "t2i_log_quant": opt.t2i_log_quant,
                         "t2i_real_time": opt.t2i_real_time, "t2i_start_peak": opt.t2i_start_peak, "log_max_1": False}
    time_aware = opt.time_aware_aqtizer if opt.use_aq else False
    qnn = get_qmodel(mt, pipe, ckpt, wq_params, opt.use_aq, aq_params, softmax_aq_params, opt.use_group,
                     num_inference_steps=steps, time_aware_aqtizer=time_aware, device=dev)
    qnn = qnn.half() if opt.fp16 else qnn.float()
    qnn = qnn.to(dev)
    qnn.disable_out_quantization()
    if opt.graphs:
        qnn.enable_graphs(True)
    dt = torch.float16 if opt.fp16 else torch.float32

    def unet_fn(x, t, ctx, **extra):
        return qnn(x, t, ctx, **extra)[0]

    outs = {}
    for i in shard_prompts(opt.n_prompts, rank, world):
        lat = synth.named_randn("latent", (1, 4, res, res), opt.seed + i).to(dev, dt)
        ctx = synth.named_randn("ctx|" + opt.prompt, (batch, 77, ARCH[mt]["ctx_dim"]), opt.seed + i).to(dev, dt)
        extra = None
        if mt == "sdxl":
            inp = synth.synth_inputs("sdxl", batch, opt.seed + i, res)
            extra = {"added_cond_kwargs": {"text_embeds": inp["text_embeds"].to(dev, dt), "time_ids": inp["time_ids"].to(dev, dt)}}
        outs[i] = denoise_loop(unet_fn, lat, ctx, steps, guidance=guidance, extra=extra).float().cpu()
        logging.info("prompt %d done: latent absmax %.4f", i, outs[i].abs().max().item())
    out = opt.out.format(rank=rank)
    torch.save(outs, out)
    logging.info("saved %s", out)


if __name__ == "__main__":
    main()
