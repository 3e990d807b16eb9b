"""Noise-level schedules (reference: sgm/modules/diffusionmodules/discretizer.py:17-39)."""
import torch


class Discretization:
    def __call__(self, n, do_append_zero=True, device="cpu", flip=False):
        sigmas = self.get_sigmas(n, device=device)
        if do_append_zero:
            sigmas = torch.cat([sigmas, sigmas.new_zeros(1)])
        return torch.flip(sigmas, (0,)) if flip else sigmas

    def get_sigmas(self, n, device):
        raise NotImplementedError


class EDMDiscretization(Discretization):
    """Karras rho-schedule: sigma_k = (smax^(1/rho) + k/(n-1) (smin^(1/rho) - smax^(1/rho)))^rho."""

    def __init__(self, sigma_min=0.002, sigma_max=80.0, rho=7.0):
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho

    def get_sigmas(self, n, device="cpu"):
        # evaluated on the host in fp32 (identical values on every backend), then moved
        ramp = torch.linspace(0, 1, n)
        lo, hi = self.sigma_min ** (1 / self.rho), self.sigma_max ** (1 / self.rho)
        return ((hi + ramp * (lo - hi)) ** self.rho).to(device)
