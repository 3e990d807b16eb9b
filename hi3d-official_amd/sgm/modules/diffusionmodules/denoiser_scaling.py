"""Preconditioning coefficients (reference: denoiser_scaling.py:40-59)."""
import torch


class VScaling:
    def __call__(self, sigma):
        inv = 1.0 / (sigma * sigma + 1.0)
        root = torch.sqrt(inv)
        return inv, -sigma * root, root, sigma.clone()      # c_skip, c_out, c_in, c_noise


class VScalingWithEDMcNoise(VScaling):
    """v-prediction scaling with the EDM noise conditioning c_noise = ln(sigma)/4."""

    def __call__(self, sigma):
        c_skip, c_out, c_in, _ = super().__call__(sigma)
        return c_skip, c_out, c_in, 0.25 * torch.log(sigma)
