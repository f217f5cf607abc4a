"""Deterministic DDIM (η = 0) for the denoise loop around QuantModel.forward.

The reference drives the UNet from the vendored diffusers pipeline (pipeline_stable_diffusion.py:1027-1040); only
the ~30 lines of scheduler arithmetic the N-step loop needs are restated here: "leading" timestep spacing with
steps_offset = 1 (diffusers scheduling_ddim.py:325-330) and the ε-prediction update (:404-450), with the public
SD-v1-4 scheduler constants (scaled_linear β 0.00085→0.012, set_alpha_to_one=False, clip_sample=False)."""
import torch


class DDIMScheduler:
    def __init__(self, num_inference_steps, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 steps_offset=1, device="cpu"):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0].clone()
        self.num_inference_steps = num_inference_steps
        self.num_train_timesteps = num_train_timesteps
        ratio = num_train_timesteps // num_inference_steps
        self.timesteps = [i * ratio + steps_offset for i in range(num_inference_steps)][::-1]
        # per-step scalar coefficients, so the update is two fused multiply-adds on the device
        self._coef = {}
        for t in self.timesteps:
            prev_t = t - ratio
            a_t = self.alphas_cumprod[t]
            a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
            self._coef[t] = (a_t, a_prev)

    def step(self, model_output, t, sample):
        a_t, a_prev = self._coef[int(t)]
        pred_x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        return a_prev ** 0.5 * pred_x0 + (1 - a_prev) ** 0.5 * model_output
