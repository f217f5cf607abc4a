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

    def step_guided(self, noise_pred, t, sample, guidance_scale):
        """Classifier-free guidance on the (unconditional ‖ conditional) prediction followed by ``step`` — the two statements of
        the pipeline loop (pipeline_stable_diffusion.py:1037-1044).  Device tensors take both as ONE launch
        (dgq_cfg_ddim_step: the same fp32 operations in the same order, each rounded to the tensors' type where that is 16-bit —
        bit-identical to the ten eager kernels, including torch's division by a host scalar as a multiplication by its reciprocal);
        anything else runs the torch statements."""
        e_u, e_c = noise_pred.chunk(2)
        if (noise_pred.is_cuda and noise_pred.dtype in (torch.float32, torch.float16, torch.bfloat16) and sample.dtype == noise_pred.dtype
                and e_u.shape == sample.shape):
            from . import ops
            a_t, a_prev = self._coef[int(t)]
            s1, s2 = float((1 - a_t) ** 0.5), (a_t ** 0.5)
            inv_s2 = float(torch.ones((), dtype=torch.float32) / s2)           # fp32 reciprocal, as the device kernel of `/ scalar` forms it
            out = ops.cfg_ddim_step(e_u, e_c, sample, float(guidance_scale), s1, inv_s2, float(a_prev ** 0.5), float((1 - a_prev) ** 0.5))
            if out is not None:
                return out
        return self.step(e_u + guidance_scale * (e_c - e_u), t, sample)


class PNDMScheduler:
    """PLMS (PNDM with ``skip_prk_steps=True``) — the scheduler the reference actually runs SD with: its pipeline is
    loaded from the SD-v1-4 repo whose scheduler_config is PNDM (src/inference_qmodel.py:56-63 -> prepare_pipe,
    src/dataset_generation.py:70).  Restates diffusers scheduling_pndm.py: ``set_timesteps`` "leading" spacing with
    steps_offset = 1 (:185-190) and the skip-PRK timestep list (:204-212), ``step_plms`` (:321-390), ``_get_prev_sample``
    (:407-449); scaled_linear β 0.00085→0.012, set_alpha_to_one=False, ε-prediction.

    N inference steps produce N + 1 UNet calls: the second timestep is visited twice (the first PLMS step is a two-stage
    Heun-like step), e.g. N = 25: [961, 921, 921, 881, ..., 1] — 26 calls that alias onto the 25 time-aware slots."""

    def __init__(self, num_inference_steps, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0].clone()             # set_alpha_to_one=False
        self.num_inference_steps = num_inference_steps
        self.num_train_timesteps = num_train_timesteps
        ratio = num_train_timesteps // num_inference_steps
        base = [i * ratio + steps_offset for i in range(num_inference_steps)]
        self.timesteps = (base[:-1] + base[-2:-1] + base[-1:])[::-1]
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def scale_model_input(self, sample, t=None):
        return sample

    def _prev_sample(self, sample, timestep, prev_timestep, model_output):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        sample_coeff = (a_prev / a_t) ** 0.5
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return sample_coeff * sample - (a_prev - a_t) * model_output / denom

    def step(self, model_output, t, sample):
        t = int(t)
        ratio = self.num_train_timesteps // self.num_inference_steps
        prev_t = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_t = t
            t = t + ratio
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        out = self._prev_sample(sample, t, prev_t, model_output)
        self.counter += 1
        return out


class EulerAncestralDiscreteScheduler:
    """SDXL-turbo's scheduler (``src/inference_qmodel.py`` runs sdxl-turbo with its repo's scheduler_config: Euler ancestral,
    scaled_linear β 0.00085→0.012, "trailing" timestep spacing, ε-prediction).  Restates diffusers
    scheduling_euler_ancestral_discrete.py: ``set_timesteps`` (:262-303: t = round(arange(T, 0, −T/N)) − 1, σ interpolated
    from ((1−ᾱ)/ᾱ)^½ in float64 and stored as float32, a trailing 0), ``init_noise_sigma`` (:219-225), ``scale_model_input``
    (:234-260) and ``step`` (:322-420) with the ancestral noise drawn from an explicit generator on the CPU and moved to the
    sample's device, as diffusers' ``randn_tensor`` does for a CPU generator.  4 steps: t = 999, 749, 499, 249."""

    def __init__(self, num_inference_steps, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        import numpy as np
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        step_ratio = num_train_timesteps / num_inference_steps
        timesteps = np.arange(num_train_timesteps, 0, -step_ratio).round().copy().astype(np.float32) - 1
        sig = np.array(((1 - alphas_cumprod) / alphas_cumprod) ** 0.5)
        sig = np.interp(timesteps, np.arange(0, len(sig)), sig)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = [float(t) for t in timesteps]
        self.num_inference_steps = num_inference_steps
        self.step_index = 0

    @property
    def init_noise_sigma(self):
        return self.sigmas.max()

    def scale_model_input(self, sample, t=None):
        sigma = self.sigmas[self.step_index]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, t, sample, generator=None):
        sigma = self.sigmas[self.step_index]
        sample = sample.to(torch.float32)
        pred_original_sample = sample - sigma * model_output
        sigma_from, sigma_to = self.sigmas[self.step_index], self.sigmas[self.step_index + 1]
        sigma_up = (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5
        sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
        derivative = (sample - pred_original_sample) / sigma
        prev_sample = sample + derivative * (sigma_down - sigma)
        noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype).to(model_output.device)
        prev_sample = prev_sample + noise * sigma_up
        self.step_index += 1
        return prev_sample.to(model_output.dtype)
