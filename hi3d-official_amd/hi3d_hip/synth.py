"""Deterministic synthetic weights and conditioning for the Hi3D hot path.

There are no checkpoints in the reference tree (README.md:35-39 are download links)
and a freshly constructed VideoUNet outputs exactly 0 because of its zero-initialised
layers (openaimodel.py:296-304, attention.py:693-699, video_model.py:436-440;
SURVEY.md section 0.6).  Parity tests and the benchmark therefore use weights drawn
here: every tensor is generated from a seed derived from ITS OWN KEY, so the values do
not depend on parameter registration order -- the reference modules (in the build
container) and this package's modules (anywhere) get bit-identical tensors.
"""
import zlib

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) * 2654435761 + seed * 1000003) & 0x7FFFFFFFFFFF)
    return g


def synth_tensor(key: str, shape, seed: int = 1) -> torch.Tensor:
    """fp32 CPU tensor for state_dict entry `key`."""
    shape = tuple(shape)
    g = _gen(key, seed)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if key.endswith("mix_factor"):
        return 0.5 + 0.75 * r                      # alpha = sigmoid(.) spread over (0.2, 0.9)
    if len(shape) >= 2:                            # conv / linear weight
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return r * (fan_in ** -0.5)
    if key.endswith("weight"):                     # norm gain
        return 1.0 + 0.1 * r
    return 0.05 * r                                # biases


def synth_state_dict(shapes: dict, seed: int = 1) -> dict:
    """shapes: {key: shape} -> {key: fp32 tensor}"""
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()}


# Residual-branch tails of the depth network (BiT bottlenecks' last GroupNorm gain, the ViT blocks' output projections,
# the fusion blocks' second convolutions).  With every tensor drawn at unit scale a 70-layer random network amplifies a
# perturbation of its input by ~3x per stage (measured: 2 % noise into BiT stage 2 -> 16 % out), so a bf16 run cannot
# be told from a wrong one; trained networks do not behave like that (timm even initialises those gains to zero).
# The depth fixtures therefore damp these tensors -- the same rule for the reference run, the oracle and the HIP path.
RESIDUAL_TAILS = (".norm3.weight", ".attn.proj.weight", ".attn.proj.bias", ".mlp.fc2.weight", ".mlp.fc2.bias",
                  "resConfUnit1.conv2.weight", "resConfUnit1.conv2.bias", "resConfUnit2.conv2.weight", "resConfUnit2.conv2.bias")


def damp_residual_tails(sd: dict, factor: float) -> dict:
    """{key: tensor} with the residual-branch tails (RESIDUAL_TAILS) multiplied by `factor`."""
    return {k: (v * factor if k.endswith(RESIDUAL_TAILS) else v) for k, v in sd.items()}


def fill_module_on_device_(module: torch.nn.Module, seed: int = 1, prefix: str = "") -> None:
    """Benchmark-only variant: same per-key distributions as `synth_tensor`, drawn with the
    DEVICE generator directly into the parameters (no 6 GB host round trip per rank).  Values
    differ from the CPU stream, so parity tests never use this."""
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            g = torch.Generator(device=v.device)
            g.manual_seed((zlib.crc32((prefix + k).encode()) * 2654435761 + seed * 1000003) & 0x7FFFFFFFFFFF)
            v.normal_(generator=g)
            if k.endswith("mix_factor"):
                v.mul_(0.75).add_(0.5)
            elif v.ndim >= 2:
                v.mul_(float(v[0].numel()) ** -0.5)
            elif k.endswith("weight"):
                v.mul_(0.1).add_(1.0)
            else:
                v.mul_(0.05)


def fill_module_(module: torch.nn.Module, seed: int = 1, prefix: str = "") -> None:
    """Overwrite every parameter/buffer of `module` in place (keys = state_dict keys).  Every tensor has its own generator (seeded
    by its key), so the draws are independent of each other and of the order: large modules are drawn on a thread pool (torch's
    CPU normal sampler is serial per call and releases the GIL: the 1.5 B parameters of the full UNet took ~40 s on one thread)."""
    sd = module.state_dict()
    keys = [k for k, v in sd.items() if v.dtype.is_floating_point]
    draw = lambda k: synth_tensor(prefix + k, sd[k].shape, seed).to(sd[k].dtype)
    if sum(sd[k].numel() for k in keys) > (1 << 24):
        import concurrent.futures as cf
        import os
        with cf.ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as ex:
            new = dict(zip(keys, ex.map(draw, keys)))
    else:
        new = {k: draw(k) for k in keys}
    module.load_state_dict(new, strict=False)


def synth_conditioning(T: int, h: int, w: int, stage: int = 1, seed: int = 0, context_dim: int = 1024,
                       adm_in: int = None):
    """Synthetic (c, uc) with the shapes the Hi3D conditioner emits:
    crossattn [1,1,1024] (uc: zeros), vector [1,adm] (same in uc), concat [T,Cc,h,w]
    (uc: zeros); Cc = 4 (stage 1) or 9+4 (stage 2: depth ~U[0,1] then latent)."""
    g = torch.Generator().manual_seed(seed)
    adm = adm_in if adm_in is not None else (768 if stage == 1 else 512)
    cc = 4 if stage == 1 else 13
    concat = torch.randn((T, cc, h, w), generator=g)
    if stage == 2:
        concat[:, :9] = torch.rand((T, 9, h, w), generator=g)
    c = {"crossattn": torch.randn((1, 1, context_dim), generator=g),
         "vector": torch.randn((1, adm), generator=g),
         "concat": concat}
    uc = {"crossattn": torch.zeros_like(c["crossattn"]), "vector": c["vector"].clone(),
          "concat": torch.zeros_like(concat)}
    x = torch.randn((T, 4, h, w), generator=g)
    return x, c, uc


def synth_unet_inputs(cfg, T: int, hw: int, seed: int):
    """Seeded VideoUNet.forward arguments for a CFG-doubled batch of 2 clips x T frames (the inputs
    of the `unet_*` golden fixtures; CPU generator, so identical wherever it is drawn)."""
    g = torch.Generator().manual_seed(seed)
    B = 2 * T
    return dict(
        x=torch.randn((B, cfg["in_channels"], hw, hw), generator=g),
        timesteps=0.25 * torch.log(torch.rand((B,), generator=g) * 50 + 0.01),
        context=torch.randn((2, 1, cfg["context_dim"]), generator=g),
        y=torch.randn((2, cfg["adm_in_channels"]), generator=g),
        image_only_indicator=torch.zeros(2, T),
    )
