"""Conditioner container (reference: sgm/modules/encoders/modules.py:71-184).

The Hi3D conditioner runs ONCE per clip (OpenCLIP ViT-H image tower, MiDaS depth, VAE
encoder of the conditioning frame ...) and is outside the denoising hot path this
framework covers (SURVEY.md section 8, rank-3 "next").  Built here: the scalar embedders and
the conditioning-frame VAE embedder (it reuses the gfx950 VAE encoder); the CLIP and MiDaS
towers are not.  What the hot path consumes is its
OUTPUT: `c` / `uc` dicts with keys crossattn [B,1,1024], vector [B,adm], concat
[T,Cc,h,w].  This container keeps the reference's combining rules for embedders that
are available and reports the ones that are not, by name, when it is asked to run them.
"""
import torch
import torch.nn as nn

from ...util import instantiate_from_config


class AbstractEmbModel(nn.Module):
    is_trainable = False
    ucg_rate = 0.0
    input_key = None


class ConcatTimestepEmbedderND(AbstractEmbModel):
    """Sinusoidal embedding of each scalar, concatenated (reference :913-929)."""

    def __init__(self, outdim):
        super().__init__()
        self.outdim = outdim

    def forward(self, x):
        from hi3d_hip import ops
        if x.ndim == 1:
            x = x[:, None]
        b, dims = x.shape
        emb = ops.timestep_embedding(x.reshape(-1).float(), self.outdim)
        return emb.reshape(b, dims * self.outdim)


class VideoPredictionEmbedderWithEncoder(AbstractEmbModel):
    """The `concat` conditioning of stage 1: the conditioning frame through the first-stage VAE encoder
    (mode of the posterior), scaled, repeated for the n_copies views (reference :951-1025).

    Inference surface only: `sigma_sampler_config` / `sigma_cond_config` (training-time noise
    augmentation inside the embedder) are refused.  The encoder is this framework's AutoencoderKL
    mirror, i.e. the gfx950 VAE encoder runtime (hi3d_hip.runtime_vae.VAEEncoderRuntime); its
    parameters keep the reference's names under `encoder.`."""

    def __init__(self, n_cond_frames, n_copies, encoder_config, sigma_sampler_config=None, sigma_cond_config=None,
                 is_ae=False, scale_factor=1.0, disable_encoder_autocast=False, en_and_decode_n_samples_a_time=None):
        super().__init__()
        if sigma_sampler_config is not None or sigma_cond_config is not None:
            raise NotImplementedError("VideoPredictionEmbedderWithEncoder: sigma_sampler / sigma_cond are training-time options")
        if not is_ae:
            raise NotImplementedError("VideoPredictionEmbedderWithEncoder: only is_ae=True (encoder.encode) is wired")
        self.n_cond_frames, self.n_copies = n_cond_frames, n_copies
        self.encoder = instantiate_from_config(encoder_config)
        self.is_ae, self.scale_factor = is_ae, scale_factor
        self.disable_encoder_autocast = disable_encoder_autocast
        self.en_and_decode_n_samples_a_time = en_and_decode_n_samples_a_time

    def forward(self, vid):
        """vid [(b n_cond_frames), 3, H, W] -> [(b n_copies), n_cond_frames*4, H/8, W/8]."""
        n = vid.shape[0]
        step = self.en_and_decode_n_samples_a_time or n
        z = torch.cat([self.encoder.encode(vid[i:i + step]) for i in range(0, n, step)], dim=0)
        z = z * self.scale_factor
        bt, c, h, w = z.shape
        b = bt // self.n_cond_frames
        z = z.reshape(b, 1, self.n_cond_frames * c, h, w)                 # (b t) c h w -> b () (t c) h w
        return z.expand(b, self.n_copies, self.n_cond_frames * c, h, w).reshape(b * self.n_copies, self.n_cond_frames * c, h, w)


class _Unavailable(AbstractEmbModel):
    def __init__(self, target, error):
        super().__init__()
        self.target, self.error = target, error

    def forward(self, *a, **k):
        raise NotImplementedError(
            f"conditioner embedder {self.target} is not part of the MI355X hot-path framework "
            f"({self.error}); feed precomputed `c`/`uc` dicts (see hi3d_hip.synth.synth_conditioning "
            "for the shapes) or run the reference conditioner once per clip")


class GeneralConditioner(nn.Module):
    OUTPUT_DIM2KEYS = {2: "vector", 3: "crossattn", 4: "concat", 5: "concat"}
    KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1}

    def __init__(self, emb_models):
        super().__init__()
        embedders = []
        for cfg in emb_models or []:
            try:
                emb = instantiate_from_config(cfg)
            except (ImportError, AttributeError, ModuleNotFoundError) as e:
                emb = _Unavailable(cfg["target"], f"{type(e).__name__}: {e}")
            emb.is_trainable = cfg.get("is_trainable", False)
            emb.ucg_rate = cfg.get("ucg_rate", 0.0)
            if "input_key" in cfg:
                emb.input_key = cfg["input_key"]
            elif "input_keys" in cfg:
                emb.input_keys = cfg["input_keys"]
            else:
                raise KeyError("need either 'input_key' or 'input_keys' for embedder " + cfg["target"])
            embedders.append(emb)
        self.embedders = nn.ModuleList(embedders)

    def forward(self, batch, force_zero_embeddings=None):
        output = {}
        force_zero_embeddings = force_zero_embeddings or []
        for emb in self.embedders:
            with torch.no_grad():
                out = emb(batch[emb.input_key]) if getattr(emb, "input_key", None) is not None \
                    else emb(*[batch[k] for k in emb.input_keys])
            outs = out if isinstance(out, (list, tuple)) else [out]
            for o in outs:
                key = self.OUTPUT_DIM2KEYS[o.dim()]
                if getattr(emb, "input_key", None) in force_zero_embeddings:
                    o = torch.zeros_like(o)
                output[key] = torch.cat((output[key], o), self.KEY2CATDIM[key]) if key in output else o
        return output

    def get_unconditional_conditioning(self, batch_c, batch_uc=None, force_uc_zero_embeddings=None):
        c = self(batch_c)
        uc = self(batch_c if batch_uc is None else batch_uc, force_uc_zero_embeddings or [])
        return c, uc
