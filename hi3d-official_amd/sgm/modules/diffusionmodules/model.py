"""VAE Encoder / Decoder parameter containers (reference: sgm/modules/diffusionmodules/model.py:487-748).
As everywhere in this framework the module only carries the reference's parameter names; the
arithmetic is hi3d_hip.runtime_vae (gfx950 kernels)."""
from ...util import ParamTree


class Encoder(ParamTree):
    temporal = False

    def __init__(self, **ddconfig):
        from ...models.autoencoder import encoder_param_shapes
        self.ddconfig = dict(ddconfig)
        super().__init__(encoder_param_shapes(self.ddconfig, prefix=""))

    def forward(self, *a, **k):
        raise RuntimeError("run through AutoencoderKL / AutoencodingEngine (MI355X runtime)")


class Decoder(ParamTree):
    temporal = False

    def __init__(self, **ddconfig):
        from ...models.autoencoder import decoder_param_shapes
        self.ddconfig = {k: v for k, v in ddconfig.items() if k not in ("video_kernel_size", "alpha", "merge_strategy", "time_mode")}
        super().__init__(decoder_param_shapes(self.ddconfig, prefix="", temporal=self.temporal,
                                              vks=tuple(getattr(self, "video_kernel_size", None) or (3, 1, 1))))

    def forward(self, *a, **k):
        raise RuntimeError("run through AutoencoderKL / AutoencodingEngine (MI355X runtime)")
