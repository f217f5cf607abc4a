"""The denoising loop of the diffusers pipelines, exactly as it drives ``pipe.unet`` — the caller side of the drop-in
(SURVEY.md §8(b), §8(f)-3).  HF diffusers is not installed in this image (the reference vendors 0.26.0, which does not
import here), so the ~20 lines of ``StableDiffusionPipeline.__call__`` step 7 / ``StableDiffusionXLPipeline.__call__``
that touch the UNet are restated; everything else of a pipeline (tokenizer, CLIP, VAE, safety checker) is out of scope.

What the loop relies on from the UNet object — and what QuantModel therefore provides (quant/quant_model.py:12-38,113-116):
  * ``unet.config.in_channels / sample_size / time_cond_proj_dim`` (+ ``addition_time_embed_dim`` for SDXL)
    (pipeline_stable_diffusion.py:958,1008; pipeline_stable_diffusion_xl.py:1084);
  * the call ``unet(latent_model_input, t, encoder_hidden_states=..., timestep_cond=..., cross_attention_kwargs=...,
    added_cond_kwargs=..., return_dict=False)[0]`` (pipeline_stable_diffusion.py:1027-1035) with ``t`` a 0-d int64 tensor
    taken from ``scheduler.timesteps``; unknown keywords are swallowed (diffusers_rewrite/sd.py:546-548);
  * SD runs PNDM/PLMS by default: N steps = N + 1 UNet calls, two of them at the same timestep, which the time-aware
    activation tables alias onto one slot (quant/calibration.py:301-304)."""
import torch

from .scheduler import DDIMScheduler, EulerAncestralDiscreteScheduler, PNDMScheduler


@torch.no_grad()
def stable_diffusion_denoise(unet, latents, prompt_embeds, num_inference_steps=25, guidance_scale=7.5, scheduler="pndm",
                             on_call=None):
    """pipeline_stable_diffusion.py:1013-1044.  ``prompt_embeds``: [2B,77,768] (negative ‖ positive) when
    guidance_scale > 1, else [B,77,768].  ``on_call(i, t)`` observes every UNet call."""
    sch = PNDMScheduler(num_inference_steps) if scheduler == "pndm" else DDIMScheduler(num_inference_steps)
    do_cfg = guidance_scale > 1.0
    assert unet.config.in_channels == latents.shape[1]
    timestep_cond = None
    if unet.config.time_cond_proj_dim is not None:
        raise NotImplementedError("guidance-scale embedding (LCM-style UNets) is not part of SD1.4 / SDXL-turbo")
    for i, t in enumerate(sch.timesteps):
        tt = torch.tensor(t, dtype=torch.int64, device=latents.device)
        latent_model_input = torch.cat([latents] * 2) if do_cfg else latents
        if on_call is not None:
            on_call(i, t)
        noise_pred = unet(latent_model_input, tt, encoder_hidden_states=prompt_embeds, timestep_cond=timestep_cond,
                          cross_attention_kwargs=None, added_cond_kwargs=None, return_dict=False)[0]
        if do_cfg and hasattr(sch, "step_guided"):
            latents = sch.step_guided(noise_pred, t, latents, guidance_scale)      # guidance + scheduler step (one launch on the GPU)
            continue
        if do_cfg:
            noise_uncond, noise_text = noise_pred.chunk(2)
            noise_pred = noise_uncond + guidance_scale * (noise_text - noise_uncond)
        latents = sch.step(noise_pred, t, latents)
    return latents


@torch.no_grad()
def sdxl_turbo_denoise(unet, latents, prompt_embeds, text_embeds, time_ids, num_inference_steps=4, generator=None, on_call=None):
    """pipeline_stable_diffusion_xl.py:1170-1200 with guidance_scale = 0 (src/inference_qmodel.py:49): no CFG batch;
    ``added_cond_kwargs = {"text_embeds", "time_ids"}``; SDXL-turbo's EulerAncestralDiscrete scheduler ("trailing" spacing:
    t = 999, 749, 499, 249 for 4 steps) with its ancestral noise drawn from ``generator`` (a CPU torch.Generator, as the
    pipeline's ``generator=`` argument; None = the global RNG).  ``latents``: unit-variance noise, scaled by
    ``init_noise_sigma`` here as ``prepare_latents`` does."""
    assert unet.config.addition_time_embed_dim is not None
    sch = EulerAncestralDiscreteScheduler(num_inference_steps)
    x = latents * sch.init_noise_sigma
    added = {"text_embeds": text_embeds, "time_ids": time_ids}
    for i, t in enumerate(sch.timesteps):
        tt = torch.tensor(t, device=latents.device)              # the scheduler's timesteps are float32 (999., 749., ...)
        inp = sch.scale_model_input(x, t)
        if on_call is not None:
            on_call(i, t)
        eps = unet(inp, tt, encoder_hidden_states=prompt_embeds, timestep_cond=None, cross_attention_kwargs=None,
                   added_cond_kwargs=added, return_dict=False)[0]
        x = sch.step(eps, t, x, generator=generator)
    return x
