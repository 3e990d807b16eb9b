"""D(x; sigma) = c_skip x + c_out F(c_in x; c_noise)   (reference: denoiser.py:11-39)."""
import torch.nn as nn

from ...util import append_dims, instantiate_from_config


class Denoiser(nn.Module):
    def __init__(self, scaling_config):
        super().__init__()
        self.scaling = instantiate_from_config(scaling_config)

    def possibly_quantize_sigma(self, sigma):
        return sigma

    def possibly_quantize_c_noise(self, c_noise):
        return c_noise

    def forward(self, network, input, sigma, cond, **additional_model_inputs):
        from hi3d_hip import fused_step
        if isinstance(input, fused_step.StepRequest):
            # EDMSampler.step_call asks for the whole step as one kernel sequence / graph replay
            done = fused_step.serve(self, network, input, additional_model_inputs)
            if done is not None:
                return done
            input, sigma, cond = input.materialize()       # not the Hi3D structure: generic math below
        sigma = self.possibly_quantize_sigma(sigma)
        flat_shape = sigma.shape
        c_skip, c_out, c_in, c_noise = self.scaling(append_dims(sigma, input.ndim))
        c_noise = self.possibly_quantize_c_noise(c_noise.reshape(flat_shape))
        net = network(input * c_in, c_noise, cond, **additional_model_inputs)
        return net * c_out + input * c_skip
