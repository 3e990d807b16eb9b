"""Conditioner container and embedders (reference: sgm/modules/encoders/modules.py:71-184, 570-728, 913-1046).

The Hi3D conditioner runs ONCE per clip.  All of its towers are built on the gfx950 kernels:
the scalar embedders (`ConcatTimestepEmbedderND`), the conditioning-frame VAE embedder (`VideoPredictionEmbedderWithEncoder`
on hi3d_hip.runtime_vae.VAEEncoderRuntime), the CLIP vision towers (`FrozenOpenCLIPImage(Prediction)Embedder` here,
`AesEmbedder` in vtdm/encoders.py, both on hi3d_hip.runtime_vit) and the MiDaS DPT-hybrid depth tower (`vtdm.encoders.DepthEmbedder`
on hi3d_hip.runtime_dpt).  open_clip / clip / timm / kornia are absent from this image: those towers and the 224 x 224 resize
are pinned against independent implementations of the same published architectures / algorithms and declared UNPINNED
against the packages themselves.  What the hot path consumes is the container's OUTPUT: `c` / `uc` dicts with
keys crossattn [B,1,1024], vector [B,adm], concat [T,Cc,h,w].  An embedder whose target cannot be imported is kept as a
named placeholder that raises when it is asked to run (`_Unavailable`).
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


# CLIP vision-tower geometries (open_clip model_configs/ViT-H-14.json; OpenAI clip ViT-L/14): width, layers, heads,
# patch, image size, output dim, MLP activation
CLIP_VISUAL_ARCHS = {
    "ViT-H-14": dict(width=1280, layers=32, heads=16, patch=14, image=224, out_dim=1024, act="gelu"),
    "ViT-L-14": dict(width=1024, layers=24, heads=16, patch=14, image=224, out_dim=768, act="quick_gelu"),
    # reduced towers for tests (same code path: head dim 80 / 64, same output widths)
    "ViT-tiny-H": dict(width=320, layers=2, heads=4, patch=14, image=224, out_dim=1024, act="gelu"),
    "ViT-tiny-L": dict(width=128, layers=2, heads=2, patch=14, image=224, out_dim=768, act="quick_gelu"),
}
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_visual_shapes(width, layers, patch, image, out_dim, **_):
    """{open_clip / OpenAI-clip visual state_dict key: shape}"""
    g = image // patch
    s = {"conv1.weight": (width, 3, patch, patch), "class_embedding": (width,), "positional_embedding": (1 + g * g, width),
         "proj": (width, out_dim)}
    for n in ("ln_pre", "ln_post"):
        s[n + ".weight"], s[n + ".bias"] = (width,), (width,)
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        for n in ("ln_1", "ln_2"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (width,), (width,)
        s[p + "attn.in_proj_weight"], s[p + "attn.in_proj_bias"] = (3 * width, width), (3 * width,)
        s[p + "attn.out_proj.weight"], s[p + "attn.out_proj.bias"] = (width, width), (width,)
        s[p + "mlp.c_fc.weight"], s[p + "mlp.c_fc.bias"] = (4 * width, width), (4 * width,)
        s[p + "mlp.c_proj.weight"], s[p + "mlp.c_proj.bias"] = (width, 4 * width), (width,)
    return s


class _ClipVisualTower(nn.Module):
    """Parameters of a CLIP vision tower under `<name>.visual.*` (the reference keeps the whole open_clip / clip
    model minus its text transformer; only `visual.*` carries weights the image path uses) + the packed gfx950
    runtime, rebuilt when any parameter changes."""

    def __init__(self, arch_cfg):
        super().__init__()
        from ...util import ParamTree
        self.cfg = dict(arch_cfg)
        self.visual = ParamTree(clip_visual_shapes(**self.cfg))
        self._rt = None

    def runtime(self, device):
        from hi3d_hip.runtime_vit import ViTRuntime
        from ...util import params_key
        key = params_key(self, device)
        if self._rt is None or self._rt[0] != key:
            sd = {k: v for k, v in self.state_dict().items()}
            self._rt = (key, ViTRuntime(sd, "visual.", self.cfg["heads"], self.cfg["act"], device))
        return self._rt[1]


class FrozenOpenCLIPImageEmbedder(AbstractEmbModel):
    """OpenCLIP vision-transformer image encoder (reference :570-728) on the gfx950 ViT runtime.

    The reference downloads / loads `version` through open_clip.create_model_and_transforms; here the tower's
    parameters are a ParamTree under the same names (`model.visual.*`): load them with load_state_dict from an
    open_clip checkpoint (`init_from_open_clip_ckpt`), nothing is fetched.  Inference surface: `output_tokens`
    and multi-crop inputs are refused (unused by the Hi3D configs)."""

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True,
                 antialias=True, ucg_rate=0.0, unsqueeze_dim=False, repeat_to_max_len=False, num_image_crops=0,
                 output_tokens=False, init_device=None):
        super().__init__()
        if arch not in CLIP_VISUAL_ARCHS:
            raise NotImplementedError(f"FrozenOpenCLIPImageEmbedder: arch {arch} (known: {sorted(CLIP_VISUAL_ARCHS)})")
        if output_tokens or num_image_crops:
            raise NotImplementedError("FrozenOpenCLIPImageEmbedder: output_tokens / num_image_crops are not wired")
        self.model = _ClipVisualTower(CLIP_VISUAL_ARCHS[arch])
        self.version, self.device, self.max_length = version, device, max_length
        self.antialias, self.ucg_rate, self.unsqueeze_dim = antialias, ucg_rate, unsqueeze_dim
        self.repeat_to_max_len, self.max_crops, self.output_tokens = repeat_to_max_len, 0, False
        self.register_buffer("mean", torch.tensor(CLIP_MEAN), persistent=False)
        self.register_buffer("std", torch.tensor(CLIP_STD), persistent=False)
        if freeze:
            self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for p in self.parameters():
            p.requires_grad = False

    def init_from_open_clip_ckpt(self, path):
        """open_clip_pytorch_model.bin / a clip state_dict: keeps `visual.*`, drops the text tower."""
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
        vis = {k: v.float() for k, v in sd.items() if k.startswith("visual.")}
        missing, unexpected = self.model.load_state_dict(vis, strict=False)
        if missing:
            raise KeyError(f"open_clip checkpoint lacks {len(missing)} visual keys, e.g. {missing[:3]}")

    def preprocess(self, x):
        """[-1,1] image -> 224 x 224 -> [0,1] -> CLIP mean / std (reference :619-628), on the GPU in two banded passes of
        `hi3d_resample_axis` with the affine fused into the second: kornia.geometry.resize(bicubic, align_corners=True,
        antialias) is kornia 0.6.9's Gaussian pre-blur + torch's bicubic, restated in hi3d_hip/resample.py (kornia itself is
        absent from this image: unpinned against the package).  antialias=False (the reference accepts it) or an up-scale:
        plain bicubic taps, no pre-blur.  Runs with x's GPU as the current device (a conditioner on cuda:1 in a process whose
        current device is cuda:0: the kernels go to the current stream of the CURRENT device)."""
        from hi3d_hip import ops
        size = self.model.cfg["image"]
        if size != 224:
            raise NotImplementedError("FrozenOpenCLIPImageEmbedder.preprocess: built for the shipped 224 x 224 towers")
        with torch.cuda.device(x.device):
            mean, std = self.mean.to(x.device, torch.float32), self.std.to(x.device, torch.float32)
            x = x.float()
            if tuple(x.shape[-2:]) == (size, size):             # kornia returns the input untouched (affwarp.py: size == input_size)
                return ((x + 1.0) / 2.0 - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)
            # ((y + 1) / 2 - mean) / std  ==  y * (0.5 / std) + (0.5 - mean) / std
            return ops.resample_image(x, "clip224" if self.antialias else "clip224_noaa",
                                      scale=(0.5 / std).contiguous(), shift=((0.5 - mean) / std).contiguous())

    def forward(self, image, no_dropout=False):
        dev = image.device if image.is_cuda else torch.device(self.device)
        z = self.model.runtime(dev).forward(self.preprocess(image.to(dev))).to(image.dtype)
        if self.ucg_rate > 0.0 and not no_dropout:
            z = torch.bernoulli((1.0 - self.ucg_rate) * torch.ones(z.shape[0], device=z.device))[:, None] * z
        if self.unsqueeze_dim:
            z = z[:, None, :]
        if self.repeat_to_max_len:
            z_ = z[:, None, :] if z.dim() == 2 else z
            return z_.expand(z_.shape[0], self.max_length, z_.shape[-1]), z
        return z

    def encode(self, image):
        return self(image)


class FrozenOpenCLIPImagePredictionEmbedder(AbstractEmbModel):
    """`crossattn` conditioning: CLIP embedding of the conditioning frame(s), "(b t) d -> b t d", repeated for the
    n_copies views (reference :1028-1046)."""

    def __init__(self, open_clip_embedding_config=None, n_cond_frames=1, n_copies=1):
        super().__init__()
        self.n_cond_frames, self.n_copies = n_cond_frames, n_copies
        cfg = open_clip_embedding_config or {"target": "sgm.modules.encoders.modules.FrozenOpenCLIPImageEmbedder"}
        self.open_clip = instantiate_from_config(cfg)

    def forward(self, vid):
        z = self.open_clip(vid)
        z = z.reshape(-1, self.n_cond_frames, z.shape[-1])
        return z.repeat_interleave(self.n_copies, dim=0)


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
            except (ImportError, AttributeError, ModuleNotFoundError, NotImplementedError) as e:
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

    def _embed(self, batch):
        """every embedder's outputs on `batch`, in order: [(embedder, [tensors])]"""
        res = []
        for emb in self.embedders:
            with torch.no_grad():
                out = emb(batch[emb.input_key]) if getattr(emb, "input_key", None) is not None \
                    else emb(*[batch[k] for k in emb.input_keys])
            res.append((emb, list(out) if isinstance(out, (list, tuple)) else [out]))
        return res

    def _assemble(self, embedded, force_zero_embeddings, copy=False):
        output = {}
        for emb, outs in embedded:
            for o in outs:
                key = self.OUTPUT_DIM2KEYS[o.dim()]
                if getattr(emb, "input_key", None) in force_zero_embeddings:
                    o = torch.zeros_like(o)
                elif copy:
                    o = o.clone()                      # (the second dict never aliases the first)
                output[key] = torch.cat((output[key], o), self.KEY2CATDIM[key]) if key in output else o
        return output

    def forward(self, batch, force_zero_embeddings=None):
        return self._assemble(self._embed(batch), force_zero_embeddings or [])

    def get_unconditional_conditioning(self, batch_c, batch_uc=None, force_uc_zero_embeddings=None):
        """(c, uc) as the reference's GeneralConditioner.get_unconditional_conditioning (encoders/modules.py:143-156: both
        passes with every ucg_rate at 0, i.e. deterministic).  When uc is derived from the SAME batch -- both Hi3D pipelines
        (pipeline_i2v_eval_v01.py:74-78, v02.py:106-110) -- the embedders run ONCE and the two dicts are assembled from the
        same outputs, the forced ones zeroed for uc: identical values, half the conditioner time (at stage 2 that pass holds
        the VAE encoder on 16 frames of 1024^2, the CLIP tower and the depth network: 0.14 s per clip)."""
        force = force_uc_zero_embeddings or []
        if batch_uc is None or batch_uc is batch_c:
            embedded = self._embed(batch_c)
            return self._assemble(embedded, []), self._assemble(embedded, force, copy=True)
        return self(batch_c), self(batch_uc, force)
