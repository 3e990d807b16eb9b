"""DiffusionEngine: owner of UNet wrapper / denoiser / sampler / conditioner / first
stage, with the chunked VAE calls (reference: sgm/models/diffusion.py:19-150).

Inference-only: the reference class is a LightningModule whose training hooks
(shared_step, configure_optimizers, EMA, logging) are out of scope of this framework."""
import math

import torch
import torch.nn as nn

from ..modules import UNCONDITIONAL_CONFIG
from ..modules.diffusionmodules.wrappers import OPENAIUNETWRAPPER
from ..util import default, disabled_train, get_obj_from_str, instantiate_from_config


class DiffusionEngine(nn.Module):
    def __init__(
        self,
        network_config,
        denoiser_config,
        first_stage_config,
        conditioner_config=None,
        sampler_config=None,
        optimizer_config=None,
        scheduler_config=None,
        loss_fn_config=None,
        network_wrapper=None,
        ckpt_path=None,
        use_ema=False,
        ema_decay_rate=0.9999,
        scale_factor=1.0,
        disable_first_stage_autocast=False,
        input_key="jpg",
        log_keys=None,
        no_cond_log=False,
        compile_model=False,
        en_and_decode_n_samples_a_time=None,
    ):
        super().__init__()
        if use_ema:
            raise NotImplementedError("use_ema is a training feature (configs ship use_ema: false)")
        self.log_keys, self.input_key = log_keys, input_key
        network = instantiate_from_config(network_config)
        self.model = get_obj_from_str(default(network_wrapper, OPENAIUNETWRAPPER))(network, compile_model=compile_model)
        self.denoiser = instantiate_from_config(denoiser_config)
        self.sampler = instantiate_from_config(sampler_config) if sampler_config is not None else None
        self.conditioner = instantiate_from_config(default(conditioner_config, UNCONDITIONAL_CONFIG))
        first_stage = instantiate_from_config(first_stage_config).eval()
        first_stage.train = disabled_train.__get__(first_stage)
        for p in first_stage.parameters():
            p.requires_grad = False
        self.first_stage_model = first_stage
        self.loss_fn = None          # training loss: out of scope
        self.use_ema = False
        self.scale_factor = scale_factor
        self.disable_first_stage_autocast = disable_first_stage_autocast
        self.no_cond_log = no_cond_log
        self.en_and_decode_n_samples_a_time = en_and_decode_n_samples_a_time
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path)

    @property
    def device(self):
        return next(self.parameters()).device

    def init_from_ckpt(self, path):
        if path.endswith("ckpt"):
            sd = torch.load(path, map_location="cpu")["state_dict"]
        elif path.endswith("safetensors"):
            from safetensors.torch import load_file
            sd = load_file(path)
        else:
            raise NotImplementedError(path)
        missing, unexpected = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")

    def get_input(self, batch):
        return batch[self.input_key]

    def _chunks(self, n):
        step = default(self.en_and_decode_n_samples_a_time, n)
        return [(i, min(n, i + step)) for i in range(0, n, step)]

    def _build_first_stage_runtime(self, device, which):
        """The first-stage model's HIP runtime (weight re-layout) built NOW, on the caller's stream, before the chunk loop lets
        side streams run (hi3d_hip.runtime_vae.run_chunks); a no-op once built and for models without the hook."""
        if device.type != "cuda":
            return
        m = self.first_stage_model
        fn = getattr(m, "runtime" if which == "decode" else "encoder_runtime", None)
        if fn is not None:
            fn(device)

    @torch.no_grad()
    def decode_first_stage(self, z):
        from hi3d_hip.runtime_vae import run_chunks
        z = z * (1.0 / self.scale_factor)

        def one(lo, hi):
            kwargs = {}
            if getattr(self.first_stage_model, "is_video_decoder", False):
                kwargs["timesteps"] = hi - lo
            return self.first_stage_model.decode(z[lo:hi], **kwargs)

        return torch.cat(run_chunks(one, self._chunks(z.shape[0]), z.device, prepare=lambda: self._build_first_stage_runtime(z.device, "decode")), dim=0)

    @torch.no_grad()
    def encode_first_stage_with_noise(self, x, noise=None):
        """encode_first_stage with explicit posterior noise (tests / reproducible runs)."""
        return self.scale_factor * self.first_stage_model.encode(x, noise=noise)

    @torch.no_grad()
    def encode_first_stage(self, x):
        from hi3d_hip.runtime_vae import run_chunks
        outs = run_chunks(lambda lo, hi: self.first_stage_model.encode(x[lo:hi]), self._chunks(x.shape[0]), x.device,
                          prepare=lambda: self._build_first_stage_runtime(x.device, "encode"))
        return self.scale_factor * torch.cat(outs, dim=0)
