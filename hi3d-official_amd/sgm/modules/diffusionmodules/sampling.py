"""Euler sampler of the EDM ODE (reference: sampling.py:21-147, 228-232), keeping the
members the Hi3D scripts touch: `sampler(denoiser, x, cond=, uc=)`, `.step_call(...)`
(pipeline_i2v_eval_v02.py:127-135), `.discretization`, `.get_sigma_gen`, `.num_steps`,
`.device`, `.guider`.

Host-side control only; no host<->device synchronisation happens inside the loop (the
reference's `s_tmin <= sigmas[i] <= s_tmax` test on a device tensor is skipped when
s_churn == 0, where it cannot change the result)."""
import torch

from ...util import append_dims, default, instantiate_from_config
from .sampling_utils import to_d

DEFAULT_GUIDER = {"target": "sgm.modules.diffusionmodules.guiders.IdentityGuider"}


class BaseDiffusionSampler:
    def __init__(self, discretization_config, num_steps=None, guider_config=None, verbose=False, device="cuda"):
        self.num_steps = num_steps
        self.discretization = instantiate_from_config(discretization_config)
        self.guider = instantiate_from_config(default(guider_config, DEFAULT_GUIDER))
        self.verbose = verbose
        self.device = device

    def prepare_sampling_loop(self, x, cond, uc=None, num_steps=None):
        sigmas = self.discretization(self.num_steps if num_steps is None else num_steps, device=self.device)
        uc = default(uc, cond)
        x *= torch.sqrt(1.0 + sigmas[0] ** 2.0)
        return x, x.new_ones([x.shape[0]]), sigmas, len(sigmas), cond, uc

    def denoise(self, x, denoiser, sigma, cond, uc):
        denoised = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc))
        return self.guider(denoised, sigma)

    def get_sigma_gen(self, num_sigmas):
        gen = range(num_sigmas - 1)
        if self.verbose:
            print(f"[sampler] {self.__class__.__name__} / {self.discretization.__class__.__name__} / "
                  f"{self.guider.__class__.__name__}: {num_sigmas - 1} steps")
        return gen


class EDMSampler(BaseDiffusionSampler):
    def __init__(self, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.s_churn, self.s_tmin, self.s_tmax, self.s_noise = s_churn, s_tmin, s_tmax, s_noise
        self._fused_off = False

    def euler_step(self, x, d, dt):
        return x + dt * d

    def possible_correction_step(self, euler_step, x, d, dt, next_sigma, denoiser, cond, uc):
        raise NotImplementedError

    def _gamma(self, sigmas, i, num_sigmas):
        if self.s_churn == 0.0:
            return 0.0
        inside = self.s_tmin <= float(sigmas[i]) <= self.s_tmax
        return min(self.s_churn / (num_sigmas - 1), 2 ** 0.5 - 1) if inside else 0.0

    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None, gamma=0.0):
        sigma_hat = sigma * (gamma + 1.0)
        if gamma > 0:
            noise = torch.randn_like(x) * self.s_noise
            x = x + noise * append_dims(sigma_hat ** 2 - sigma ** 2, x.ndim) ** 0.5
        denoised = self.denoise(x, denoiser, sigma_hat, cond, uc)
        d = to_d(x, sigma_hat, denoised)
        dt = append_dims(next_sigma - sigma_hat, x.ndim)
        return self.possible_correction_step(self.euler_step(x, d, dt), x, d, dt, next_sigma, denoiser, cond, uc)

    def step_call(self, denoiser, x, i, s_in, sigmas, num_sigmas, cond, uc):
        gamma = self._gamma(sigmas, i, num_sigmas)
        if gamma == 0.0 and not self._fused_off and type(self) is EulerEDMSampler:
            from hi3d_hip import fused_step
            if fused_step.StepRequest.eligible(self, x, cond, uc):
                # the whole step (CFG batch build, denoiser scaling, UNet, guidance, Euler update) as one
                # HIP kernel sequence / graph replay: the request rides through the caller's closure to
                # Denoiser.forward (hi3d_hip/fused_step.py)
                req = fused_step.StepRequest(self.guider, x, sigmas, i, cond, uc)
                try:
                    out = denoiser(req, sigmas[i:i + 1], cond)
                except (TypeError, AttributeError) as e:
                    if "StepRequest" not in str(e):
                        raise
                    self._fused_off = True            # the closure is not a pass-through: generic path from now on
                    out = None
                if isinstance(out, fused_step.StepResult):
                    return out.x
                if out is not None:                   # generic Denoiser math ran on the materialised batch
                    sigma = s_in * sigmas[i]
                    d = to_d(x, sigma, self.guider(out, sigma))
                    return self.euler_step(x, d, append_dims(s_in * sigmas[i + 1] - sigma, x.ndim))
        return self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, cond, uc, gamma)

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None):
        x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
        for i in self.get_sigma_gen(num_sigmas):
            x = self.step_call(denoiser, x, i, s_in, sigmas, num_sigmas, cond, uc)
        return x


class EulerEDMSampler(EDMSampler):
    def possible_correction_step(self, euler_step, x, d, dt, next_sigma, denoiser, cond, uc):
        return euler_step
