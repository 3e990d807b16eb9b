"""Network wrapper that maps conditioning keys onto UNet arguments
(reference: wrappers.py:9-34)."""
import torch
import torch.nn as nn

OPENAIUNETWRAPPER = "sgm.modules.diffusionmodules.wrappers.OpenAIWrapper"


class IdentityWrapper(nn.Module):
    def __init__(self, diffusion_model, compile_model=False):
        super().__init__()
        # no tracing compiler on this backend: the runtime is already a static kernel plan
        self.diffusion_model = diffusion_model

    def forward(self, *args, **kwargs):
        return self.diffusion_model(*args, **kwargs)


class OpenAIWrapper(IdentityWrapper):
    def forward(self, x, t, c, **kwargs):
        concat = c.get("concat", None)
        if concat is not None and concat.numel() > 0:
            x = torch.cat((x, concat.to(x.dtype)), dim=1)
        return self.diffusion_model(x, timesteps=t, context=c.get("crossattn", None), y=c.get("vector", None), **kwargs)
