"""The two Hi3D denoising procedures as functions of an engine and its conditioning (`c`, `uc`: the output of
`model.conditioner.get_unconditional_conditioning(...)` -- the once-per-clip conditioner of sgm/modules/encoders and
vtdm/encoders.py, which runs on the same kernels -- or synthetic dicts of the same shapes, hi3d_hip.synth.synth_conditioning):

  stage1_denoise  -- pipeline_i2v_eval_v01.py:62-98   (noise -> 25 Euler-EDM steps -> VAE decode)
  stage2_refine   -- pipeline_i2v_eval_v02.py:77-141  (per-frame VAE encode of the stage-1 video,
                     SDEdit-style re-noising blend before every step, then decode)
Host control flow only; every tensor op inside is a hi3d_hip kernel or the engine's modules."""
import math

import torch

from . import ops
from .fused_step import StepRequest


def _denoiser_fn(model, T, device):
    """The closure both reference pipelines build (pipeline_i2v_eval_v01.py:80-88).  image_only_indicator
    has one row per clip of the batch the network sees: 2 for the CFG-doubled batch, 1 per rank under
    hi3d_hip.parallel.SplitCFGGuider (each rank evaluates one half)."""
    ioi = {n: torch.zeros(n, T, device=device) for n in (1, 2)}

    def denoiser(inp, sigma, c):
        clips = 2 if isinstance(inp, StepRequest) else max(1, inp.shape[0] // T)
        if clips not in ioi:
            ioi[clips] = torch.zeros(clips, T, device=device)
        return model.denoiser(model.model, inp, sigma, c, image_only_indicator=ioi[clips], num_video_frames=T)
    return denoiser


def _decode(model, latents, decode_group):
    """decode_first_stage of the clip (sgm/models/diffusion.py:117-135).  decode_group = a torch.distributed process group
    (or True: the default group) over which the FRAMES are sharded: every rank decodes T / world frames with the HIP VAE and
    ONE all-gather (RCCL on the GPUs) reassembles the clip on every rank -- north_star's "RCCL all-gather at VAE-decode
    hand-off" (hi3d_hip.parallel.decode_sharded).  None: this rank decodes all frames."""
    if decode_group is None or decode_group is False:
        return model.decode_first_stage(latents)
    from .parallel import decode_sharded
    return decode_sharded(model.decode_first_stage, latents, None if decode_group is True else decode_group)


@torch.no_grad()
def stage1_denoise(model, c, uc, T, h, w, noise=None, decode=True, decode_group=None):
    dev = model.device
    x = torch.randn((T, 4, h, w), device=dev) if noise is None else noise.to(dev, torch.float32).clone()
    samples = model.sampler(_denoiser_fn(model, T, dev), x, cond=c, uc=uc)
    return _decode(model, samples, decode_group) if decode else samples


def v02_alpha(i, num_steps, alpha_pow=40.0):
    """alpha_i = (0.5 (1 + cos(i / num_steps)))^40 -- no pi in the reference (v02.py:128-129)."""
    return math.pow(0.5 * (1.0 + math.cos(i * 1.0 / num_steps)), alpha_pow)


@torch.no_grad()
def stage2_refine(model, frames, c, uc, init_noise=None, encode_noise=None, decode=True, z_frames=None, decode_group=None):
    """frames: [3, T, H, W] stage-1 video in [-1, 1] (or pass z_frames [T,4,h,w] directly).
    decode_group: shard the final VAE decode over a process group (see _decode)."""
    dev = model.device
    sampler = model.sampler
    sigmas = sampler.discretization(sampler.num_steps, device=dev)
    sig_host = sigmas.float().cpu().tolist()
    num_sigmas = len(sigmas)
    if z_frames is None:
        T = frames.shape[1]
        zs = []
        for t in range(T):                                       # per-frame encode (v02.py:96-101)
            n = None if encode_noise is None else encode_noise[t:t + 1]
            zs.append(model.encode_first_stage_with_noise(frames[:, t].unsqueeze(0).to(dev), n))
        z_frames = torch.cat(zs, 0)
    z_frames = z_frames.to(dev, torch.float32).contiguous()
    T, _, h, w = z_frames.shape
    init = (torch.randn((T, 4, h, w), device=dev) if init_noise is None else init_noise.to(dev, torch.float32)).contiguous()
    latents = (init * math.sqrt(1.0 + sig_host[0] ** 2)).contiguous()
    s_in = latents.new_ones([T])
    den = _denoiser_fn(model, T, dev)
    for i in sampler.get_sigma_gen(num_sigmas):
        ops.v02_blend(latents, init, z_frames, v02_alpha(i, sampler.num_steps), sig_host[i])
        latents = sampler.step_call(den, latents, i, s_in, sigmas, num_sigmas, c, uc).contiguous()
    return _decode(model, latents, decode_group) if decode else latents
