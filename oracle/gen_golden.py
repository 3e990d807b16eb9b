"""TEST INFRASTRUCTURE -- generate tests/golden/*.pt by running the REFERENCE classes.

Runs only in the build container (needs /root/reference; see oracle/ref_import.py for the
four inert stand-in modules).  Each fixture stores the seeded inputs, the config, the
weight seed (weights are re-derived per key by hi3d_hip.synth, never stored) and the
reference's fp32 CPU output.  Deviations from the shipped YAMLs, all numerically neutral
(SURVEY.md 8c): attention type `softmax` (SDPA) instead of xformers, VAE attn `vanilla`,
sampler device cpu, use_checkpoint False, zero-initialised parameters re-drawn.

usage:  python oracle/gen_golden.py [--only NAME] [--full]
"""
import argparse
import hashlib
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd", "hi3d_hip"))  # synth only (no package import: name clash with reference `sgm`)
import ref_import  # noqa: E402
import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
UNET_PREFIX = "model.diffusion_model."
VAE_PREFIX = "first_stage_model."


def unet_cfg(stage, mc=320):
    return dict(in_channels=8 if stage == 1 else 17, model_channels=mc, out_channels=4, num_res_blocks=2,
                attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4], num_head_channels=64,
                use_linear_in_transformer=True, transformer_depth=1, context_dim=1024,
                spatial_transformer_attn_type="softmax", extra_ff_mix_layer=True, use_spatial_context=True,
                merge_strategy="learned_with_images", video_kernel_size=[3, 1, 1], num_classes="sequential",
                adm_in_channels=768 if stage == 1 else 512, use_checkpoint=False)


def vae_ddconfig(ch=128):
    return dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3,
                ch=ch, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def shapes_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(f"{k}:{tuple(sd[k].shape)};".encode())
    return h.hexdigest()


def build_unet(cfg, seed):
    VideoUNet = ref_import.ref("sgm.modules.diffusionmodules.video_model.VideoUNet")
    m = VideoUNet(**cfg).eval()
    synth.fill_module_(m, seed, prefix=UNET_PREFIX)
    return m


def unet_inputs(cfg, T, hw, seed):
    return synth.synth_unet_inputs(cfg, T, hw, seed)


def gen_unet(name, cfg, T, hw, wseed=1, iseed=0, ioi=None, compact=False, halves=False):
    """compact: the inputs are NOT stored (they are re-drawn from `input_seed` by
    synth.synth_unet_inputs, pinned by `input_probe`) and the output is stored in fp16 -- for the
    full-size stage-2 forward, whose fp32 input alone is 36 MB."""
    t0 = time.time()
    m = build_unet(cfg, wseed)
    inp = unet_inputs(cfg, T, hw, iseed)
    if ioi is not None:
        inp["image_only_indicator"] = ioi
    with torch.no_grad():
        if halves:
            # the two clips of the CFG-doubled batch one after the other (they never mix inside the network: every rearrange
            # keeps b outermost) -- halves the reference's peak memory: BASELINE config 4 at its real size
            # (2 x 32 frames, latent 128 x 128) needs ~90 GB in one call, ~45 GB this way (the build container has 62)
            outs = []
            for b in (0, 1):
                fs = slice(b * T, (b + 1) * T)
                outs.append(m(inp["x"][fs], inp["timesteps"][fs], context=inp["context"][b:b + 1], y=inp["y"][b:b + 1], time_context=None,
                              num_video_frames=T, image_only_indicator=inp["image_only_indicator"][b:b + 1]))
                print(f"{name}: clip {b} done ({time.time() - t0:.0f}s)", flush=True)
            out = torch.cat(outs, 0)
            del outs
        else:
            out = m(inp["x"], inp["timesteps"], context=inp["context"], y=inp["y"], time_context=None,
                    num_video_frames=T, image_only_indicator=inp["image_only_indicator"])
    sd = m.state_dict()
    fx = dict(kind="unet", cfg=cfg, T=T, two_calls=bool(halves), weight_seed=wseed, key_prefix=UNET_PREFIX, inputs=inp, output=out,
              n_tensors=len(sd), shapes_sha256=shapes_digest(sd), shapes={k: tuple(v.shape) for k, v in sd.items()},
              probe={k: sd[k].flatten()[:4].clone() for k in list(sd)[:3]})
    if compact:
        x = inp["x"]
        fx["inputs"] = None
        fx["input_seed"], fx["hw"] = iseed, hw
        fx["input_probe"] = dict(head=x.flatten()[:16].clone(), sum=float(x.double().sum()), abs_sum=float(x.double().abs().sum()))
        fx["output_absmax"], fx["output_std"] = float(out.abs().max()), float(out.std())
        fx["output"] = out.to(torch.float16)
    torch.save(fx, os.path.join(GOLD, name + ".pt"))
    print(f"{name}: out {tuple(out.shape)} absmax {out.abs().max():.4f} std {out.std():.4f}  ({time.time() - t0:.1f}s)", flush=True)


def gen_sampler(name, cfg, T, hw, steps, max_scale, stage, wseed=1, iseed=0):
    t0 = time.time()
    unet = build_unet(cfg, wseed)
    Wrapper = ref_import.ref("sgm.modules.diffusionmodules.wrappers.OpenAIWrapper")
    Denoiser = ref_import.ref("sgm.modules.diffusionmodules.denoiser.Denoiser")
    Sampler = ref_import.ref("sgm.modules.diffusionmodules.sampling.EulerEDMSampler")
    model = Wrapper(unet)
    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = Sampler(
        num_steps=steps, verbose=False, device="cpu",
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization",
                               "params": {"sigma_max": 700.0}},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": max_scale, "min_scale": 1.0}})
    x0, c, uc = synth.synth_conditioning(T, hw, hw, stage=stage, seed=iseed, adm_in=cfg["adm_in_channels"])
    extra = dict(image_only_indicator=torch.zeros(2, T), num_video_frames=T)
    traj = []

    def denoiser(inp, sigma, cc):
        return den(model, inp, sigma, cc, **extra)

    with torch.no_grad():
        # same loop as EDMSampler.__call__, unrolled through step_call (the v02 entry point)
        # so the per-step states can be recorded
        x, s_in, sigmas, num_sigmas, cond, ucond = sampler.prepare_sampling_loop(x0.clone(), c, uc, steps)
        for i in sampler.get_sigma_gen(num_sigmas):
            x = sampler.step_call(denoiser, x, i, s_in, sigmas, num_sigmas, cond, ucond)
            traj.append(x.clone())
        whole = sampler(denoiser, x0.clone(), cond=c, uc=uc)
    assert torch.equal(whole, x)
    fx = dict(kind="sampler", cfg=cfg, T=T, steps=steps, max_scale=max_scale, stage=stage, weight_seed=wseed,
              key_prefix=UNET_PREFIX, x0=x0, c=c, uc=uc, sigmas=sigmas, traj=torch.stack(traj), output=x,
              shapes={k: tuple(v.shape) for k, v in unet.state_dict().items()})
    torch.save(fx, os.path.join(GOLD, name + ".pt"))
    print(f"{name}: final absmax {x.abs().max():.4f} std {x.std():.4f} ({time.time() - t0:.1f}s)")


def _ref_sampler(unet, T, steps, max_scale):
    Wrapper = ref_import.ref("sgm.modules.diffusionmodules.wrappers.OpenAIWrapper")
    Denoiser = ref_import.ref("sgm.modules.diffusionmodules.denoiser.Denoiser")
    Sampler = ref_import.ref("sgm.modules.diffusionmodules.sampling.EulerEDMSampler")
    model = Wrapper(unet)
    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = Sampler(
        num_steps=steps, verbose=False, device="cpu",
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization",
                               "params": {"sigma_max": 700.0}},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": max_scale, "min_scale": 1.0}})
    extra = dict(image_only_indicator=torch.zeros(2, T), num_video_frames=T)

    def denoiser(inp, sigma, cc):
        return den(model, inp, sigma, cc, **extra)

    return sampler, denoiser


def _host_facts(threads):
    import platform
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(threads=threads, cores=os.cpu_count(), cpu=model, torch=torch.__version__, python=platform.python_version(),
                dtype="fp32", where="build container (no GPU)")


def _ref_decode_frames(z, frames, ch=128, wseed=1):
    """decode_first_stage (models/diffusion.py:117-135: z / scale_factor, AutoencoderKL.decode) + the pipeline's rearrange +
    tensor2vid (pipeline_i2v_eval_v01.py:95-99, vtdm/util.py:13-21) of the REFERENCE on the frames `frames` of the clip z."""
    import einops
    AE = ref_import.ref("sgm.models.autoencoder.AutoencoderKL")
    tensor2vid = ref_import.ref_vtdm_util("tensor2vid")
    ae = AE(embed_dim=4, ddconfig=vae_ddconfig(ch), lossconfig={"target": "torch.nn.Identity"}).eval()
    synth.fill_module_(ae, wseed, prefix=VAE_PREFIX)
    imgs = []
    with torch.no_grad():
        for f in frames:                       # en_and_decode_n_samples_a_time: frames are independent in AutoencoderKL
            imgs.append(ae.decode(z[f:f + 1] / 0.18215))
    img = torch.cat(imgs, 0)
    vid = tensor2vid(einops.rearrange(img.clone(), "(b t) c h w -> b c t h w", t=len(frames)))
    return img, torch.from_numpy(__import__("numpy").stack(vid))


def gen_sampler_at_size(name, cfg, T, hw, steps, n_run, max_scale, stage, wseed=1, iseed=0, decode_frames=None):
    """The reference's EulerEDMSampler.step_call (sampling.py:93-147) + LinearPredictionGuider (guiders.py:78-99) + Denoiser
    (denoiser.py:23-39) + OpenAIWrapper + VideoUNet at a BENCHMARKED size: the first `n_run` steps of the `steps`-step schedule.

    What is stored per step is the guided denoised estimate D_i (the return value of EDMSampler.denoise, sampling.py:54-57, in
    fp16: |D| is O(1)), NOT the state: at sigma_0 = 700 the state is ~2800 in magnitude and carries D with weight
    1 - sigma_1/sigma_0, so a state stored in fp16 (resolution 2.0 at that magnitude) would not see the network at all, and a
    state compared in relative max-abs terms would pass with a garbage network.  The test rebuilds the reference states from
    x0 and the D_i with the Euler update (exact up to D's fp16 rounding, pinned against `last_state`, stored in fp32).
    Compact: x0 / c / uc are re-drawn from `input_seed` by synth.synth_conditioning (pinned by `x0_probe`).  Records the wall time
    of every step: this IS the metric's unit (one CFG-doubled denoise step) on the reference classes."""
    t0 = time.time()
    threads = torch.get_num_threads()
    unet = build_unet(cfg, wseed)
    t_build = time.time() - t0
    sampler, denoiser = _ref_sampler(unet, T, steps, max_scale)
    x0, c, uc = synth.synth_conditioning(T, hw, hw, stage=stage, seed=iseed, adm_in=cfg["adm_in_channels"])
    Ds, step_s = [], []
    ref_denoise = sampler.denoise

    def recording_denoise(*a, **k):
        d = ref_denoise(*a, **k)
        Ds.append(d.to(torch.float16))
        return d
    sampler.denoise = recording_denoise
    with torch.no_grad():
        x, s_in, sigmas, num_sigmas, cond, ucond = sampler.prepare_sampling_loop(x0.clone(), c, uc, steps)
        for i in sampler.get_sigma_gen(num_sigmas):
            if i >= n_run:
                break
            t1 = time.time()
            x = sampler.step_call(denoiser, x, i, s_in, sigmas, num_sigmas, cond, ucond)
            step_s.append(time.time() - t1)
            print(f"{name}: step {i} {step_s[-1]:.1f}s |x| {x.abs().max():.3f} |D| {Ds[-1].float().abs().max():.3f}", flush=True)
    assert len(Ds) == n_run
    fx = dict(kind="sampler_at_size", cfg=cfg, T=T, hw=hw, steps=steps, n_run=n_run, max_scale=max_scale, stage=stage,
              weight_seed=wseed, key_prefix=UNET_PREFIX, input_seed=iseed,
              x0_probe=dict(head=x0.flatten()[:16].clone(), sum=float(x0.double().sum()), abs_sum=float(x0.double().abs().sum())),
              sigmas=sigmas, denoised_f16=torch.stack(Ds), last_state=x.clone(), last_step=n_run - 1,
              ref_step_seconds=step_s, ref_build_seconds=t_build, ref_host=_host_facts(threads),
              shapes_sha256=shapes_digest(unet.state_dict()))
    if decode_frames:
        del unet
        t1 = time.time()
        img, vid = _ref_decode_frames(x, decode_frames)
        fx.update(decode_frames=list(decode_frames), decoded_f16=img.to(torch.float16), decoded_u8=vid,
                  ref_decode_seconds=time.time() - t1)
    torch.save(fx, os.path.join(GOLD, name + ".pt"))
    print(f"{name}: {n_run} steps, mean {sum(step_s) / len(step_s):.1f}s/step, total {time.time() - t0:.0f}s", flush=True)


DECODED_CROP = 512


def compact_decoded(name):
    """One-off for a fixture written before the crop above existed (the 4-hour v02_s2_full_25step job of round 6 was already
    running): keep the centre crop of its fp16 image, as gen_v02_at_size now does.  Nothing is recomputed."""
    path = os.path.join(GOLD, name + ".pt")
    fx = torch.load(path, weights_only=False)
    if "decoded_crop" in fx:
        return
    img = fx["decoded_f16"]
    c0 = (img.shape[-1] - DECODED_CROP) // 2
    fx.update(decoded_crop=(c0, DECODED_CROP), decoded_absmax=float(img.float().abs().max()),
              decoded_f16=img[..., c0:c0 + DECODED_CROP, c0:c0 + DECODED_CROP].clone())
    torch.save(fx, path)
    print(f"{name}: decoded_f16 cropped to {tuple(fx['decoded_f16'].shape)}")


def gen_v02_at_size(name, cfg, T, hw, steps, max_scale, keep, wseed=1, iseed=0, decode_frames=None):
    """BASELINE config 3 end to end on the reference classes: the stage-2 refine loop of pipeline_i2v_eval_v02.py:103-135 (re-noising
    blend alpha_i = (0.5 (1 + cos(i / 25)))^40 with the per-frame encoded latents, then EulerEDMSampler.step_call with CFG 1 -> 2)
    at its REAL size -- 16 views, latent 128 x 128, 17 input channels, all 25 steps (~10 min per step on 8 cores, ~45 GB) -- then
    the reference's decode + tensor2vid of `decode_frames`.  Stored: the guided denoised estimate D_i (fp16) of the steps in `keep`,
    the final latents (fp32), the decoded frames; inputs re-drawn from `input_seed` (init / c / uc: synth.synth_conditioning;
    z frames: N(0, 0.8^2) from seed + 100, as gen_v02)."""
    import math
    t0 = time.time()
    threads = torch.get_num_threads()
    unet = build_unet(cfg, wseed)
    sampler, denoiser = _ref_sampler(unet, T, steps, max_scale)
    append_dims = ref_import.ref("sgm.util.append_dims")
    init, c, uc = synth.synth_conditioning(T, hw, hw, stage=2, seed=iseed, adm_in=cfg["adm_in_channels"])
    g = torch.Generator().manual_seed(iseed + 100)
    z_list = [torch.randn((1, 4, hw, hw), generator=g) * 0.8 for _ in range(T)]
    Ds, step_s, kept = [], [], {}
    ref_denoise = sampler.denoise

    def recording_denoise(*a, **k):
        d = ref_denoise(*a, **k)
        Ds.append(d)
        return d
    sampler.denoise = recording_denoise
    with torch.no_grad():
        sigmas = sampler.discretization(sampler.num_steps, device="cpu")
        num_sigmas = len(sigmas)
        s_in = init.new_ones([T])
        latents = init.clone()
        latents *= torch.sqrt(1.0 + sigmas[0] ** 2.0)
        z = z_list[0]
        for i in sampler.get_sigma_gen(num_sigmas):
            t1 = time.time()
            alpha = math.pow(0.5 * (1 + math.cos(i * 1.0 / sampler.num_steps)), 40.0)
            for t in range(T):
                latents[t:t + 1] = latents[t:t + 1] * (1 - alpha) + (init[t:t + 1] * append_dims(sigmas[i], z.ndim) + z_list[t]) * alpha
            latents = sampler.step_call(denoiser, latents, i, s_in, sigmas, num_sigmas, c, uc)
            step_s.append(time.time() - t1)
            if i in keep:
                kept[i] = Ds[-1].to(torch.float16)
            Ds.clear()
            print(f"{name}: step {i} {step_s[-1]:.1f}s |x| {latents.abs().max():.3f}", flush=True)
            # (partial fixture after every step: a 4-hour job that is cut short still leaves what it finished)
            torch.save(dict(kind="v02_at_size_partial", steps_done=i + 1, kept_steps=sorted(kept), ref_step_seconds=list(step_s)),
                       os.path.join(GOLD, name + ".partial.pt"))
    fx = dict(kind="v02_at_size", cfg=cfg, T=T, hw=hw, steps=steps, max_scale=max_scale, weight_seed=wseed, key_prefix=UNET_PREFIX,
              input_seed=iseed, init_probe=dict(head=init.flatten()[:16].clone(), sum=float(init.double().sum()),
                                                abs_sum=float(init.double().abs().sum())),
              z_probe=dict(head=z_list[0].flatten()[:16].clone(), sum=float(torch.cat(z_list).double().sum())),
              sigmas=sigmas, kept_steps=sorted(kept), denoised_f16=torch.stack([kept[i] for i in sorted(kept)]),
              output=latents.clone(), ref_step_seconds=step_s, ref_host=_host_facts(threads),
              shapes_sha256=shapes_digest(unet.state_dict()))
    if decode_frames:
        del unet
        t1 = time.time()
        img, vid = _ref_decode_frames(latents, decode_frames)
        # the fp16 image is kept for a centre crop only (PSNR of the unclamped decoder output: 2 x 3 x 1024^2 fp16 would be 12.6 MB);
        # the uint8 frames of tensor2vid are kept whole
        c0 = (img.shape[-1] - DECODED_CROP) // 2
        fx.update(decode_frames=list(decode_frames), decoded_crop=(c0, DECODED_CROP),
                  decoded_f16=img[..., c0:c0 + DECODED_CROP, c0:c0 + DECODED_CROP].to(torch.float16).clone(), decoded_u8=vid,
                  decoded_absmax=float(img.abs().max()), ref_decode_seconds=time.time() - t1)
    torch.save(fx, os.path.join(GOLD, name + ".pt"))
    try:
        os.remove(os.path.join(GOLD, name + ".partial.pt"))
    except OSError:
        pass
    print(f"{name}: {steps} steps, mean {sum(step_s) / len(step_s):.1f}s/step, total {time.time() - t0:.0f}s", flush=True)


def gen_decode_of(name, src, ch=128):
    """Full-width end-to-end image golden: the REFERENCE's decode_first_stage + tensor2vid of the final latents an existing
    reference-class sampler fixture holds (`src`.output) -- sample -> decode -> uint8 frames all from reference classes."""
    t0 = time.time()
    z = torch.load(os.path.join(GOLD, src + ".pt"), weights_only=False)["output"]
    frames = list(range(z.shape[0]))
    img, vid = _ref_decode_frames(z, frames, ch=ch)
    fx = dict(kind="decode_of", src=src, ddconfig=vae_ddconfig(ch), weight_seed=1, key_prefix=VAE_PREFIX,
              z_head=z.flatten()[:16].clone(), decoded=img, decoded_u8=vid)
    torch.save(fx, os.path.join(GOLD, name + ".pt"))
    print(f"{name}: img {tuple(img.shape)} absmax {img.abs().max():.3f} u8 {tuple(vid.shape)} ({time.time() - t0:.1f}s)")


def gen_vae(name, ch, n, hw, wseed=1, iseed=0, compact=False):
    """compact: z is re-drawn from `input_seed` by the test (pinned by `z_head`), the image is stored in fp16."""
    t0 = time.time()
    AE = ref_import.ref("sgm.models.autoencoder.AutoencoderKL")
    dd = vae_ddconfig(ch)
    ae = AE(embed_dim=4, ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}).eval()
    synth.fill_module_(ae, wseed, prefix=VAE_PREFIX)
    g = torch.Generator().manual_seed(iseed)
    z = torch.randn((n, 4, hw, hw), generator=g)
    with torch.no_grad():
        out = ae.decode(z / 0.18215)      # decode_first_stage does z / scale_factor first (diffusion.py:119)
    sd = ae.state_dict()
    fx = dict(kind="vae_decode", ddconfig=dd, weight_seed=wseed, key_prefix=VAE_PREFIX, z=z, output=out,
              shapes={k: tuple(v.shape) for k, v in sd.items()}, shapes_sha256=shapes_digest(sd))
    if compact:
        fx.update(z=None, input_seed=iseed, z_shape=tuple(z.shape), z_head=z.flatten()[:16].clone(),
                  output=out.to(torch.float16), output_absmax=float(out.abs().max()))
    torch.save(fx, os.path.join(GOLD, name + ".pt"))
    print(f"{name}: out {tuple(out.shape)} absmax {out.abs().max():.4f} ({time.time() - t0:.1f}s)")


def gen_vae_encode(name, ch, n, hw, wseed=1, iseed=0, compact=False):
    """compact: x is re-drawn from `input_seed` by the test (pinned by `x_head`)."""
    t0 = time.time()
    AE = ref_import.ref("sgm.models.autoencoder.AutoencoderKL")
    dd = vae_ddconfig(ch)
    ae = AE(embed_dim=4, ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}).eval()
    synth.fill_module_(ae, wseed, prefix=VAE_PREFIX)
    g = torch.Generator().manual_seed(iseed)
    x = torch.rand((n, 3, hw, hw), generator=g) * 2 - 1
    noise = torch.randn((n, 4, hw // 8, hw // 8), generator=g)
    with torch.no_grad():
        moments = ae.quant_conv(ae.encoder(x))
        torch.manual_seed(1234)                       # posterior.sample() draws from the global CPU generator
        z_sampled = ae.encode(x)
        torch.manual_seed(1234)
        ref_noise = torch.randn(z_sampled.shape)
    sd = ae.state_dict()
    fx = dict(kind="vae_encode", ddconfig=dd, weight_seed=wseed, key_prefix=VAE_PREFIX, x=x, moments=moments,
              z_sampled=z_sampled, sample_noise=ref_noise, noise=noise,
              shapes={k: tuple(v.shape) for k, v in sd.items()})
    if compact:
        fx.update(x=None, input_seed=iseed, x_shape=tuple(x.shape), x_head=x.flatten()[:16].clone())
    torch.save(fx, os.path.join(GOLD, name + ".pt"))
    print(f"{name}: moments {tuple(moments.shape)} absmax {moments.abs().max():.4f} ({time.time() - t0:.1f}s)")


def gen_v02(name, cfg, T, hw, steps, max_scale, wseed=1, iseed=0):
    """pipeline_i2v_eval_v02.py:103-135 with the reference sampler/denoiser/guider classes."""
    import math
    t0 = time.time()
    unet = build_unet(cfg, wseed)
    Wrapper = ref_import.ref("sgm.modules.diffusionmodules.wrappers.OpenAIWrapper")
    Denoiser = ref_import.ref("sgm.modules.diffusionmodules.denoiser.Denoiser")
    Sampler = ref_import.ref("sgm.modules.diffusionmodules.sampling.EulerEDMSampler")
    append_dims = ref_import.ref("sgm.util.append_dims")
    model = Wrapper(unet)
    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = Sampler(
        num_steps=steps, verbose=False, device="cpu",
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization",
                               "params": {"sigma_max": 700.0}},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": max_scale, "min_scale": 1.0}})
    init, c, uc = synth.synth_conditioning(T, hw, hw, stage=2, seed=iseed, adm_in=cfg["adm_in_channels"])
    g = torch.Generator().manual_seed(iseed + 100)
    z_list = [torch.randn((1, 4, hw, hw), generator=g) * 0.8 for _ in range(T)]
    extra = dict(image_only_indicator=torch.zeros(2, T), num_video_frames=T)

    def denoiser(inp, sigma, cc):
        return den(model, inp, sigma, cc, **extra)

    with torch.no_grad():
        sigmas = sampler.discretization(sampler.num_steps, device="cpu")
        num_sigmas = len(sigmas)
        s_in = init.new_ones([T])
        latents = init.clone()
        latents *= torch.sqrt(1.0 + sigmas[0] ** 2.0)
        z = z_list[0]
        for i in sampler.get_sigma_gen(num_sigmas):
            alpha = math.pow(0.5 * (1 + math.cos(i * 1.0 / sampler.num_steps)), 40.0)
            for t in range(T):
                latents[t:t + 1] = latents[t:t + 1] * (1 - alpha) + (init[t:t + 1] * append_dims(sigmas[i], z.ndim) + z_list[t]) * alpha
            latents = sampler.step_call(denoiser, latents, i, s_in, sigmas, num_sigmas, c, uc)
    fx = dict(kind="v02", cfg=cfg, T=T, steps=steps, max_scale=max_scale, weight_seed=wseed, key_prefix=UNET_PREFIX,
              init=init, c=c, uc=uc, z_frames=torch.cat(z_list, 0), output=latents,
              shapes={k: tuple(v.shape) for k, v in unet.state_dict().items()})
    torch.save(fx, os.path.join(GOLD, name + ".pt"))
    print(f"{name}: final absmax {latents.abs().max():.4f} ({time.time() - t0:.1f}s)")


def gen_video_decode(name, ch, b, T, hw, wseed=1, iseed=0, vks=(3, 1, 1)):
    """vks: VideoDecoder's video_kernel_size -- [3, 1, 1] is what SVD / Hi3D ship; the int 3 is the reference class's DEFAULT
    (temporal_ae.py:24,87-92,299: an int makes time_stack and time_mix_conv isotropic Conv3d(3, padding 1))."""
    t0 = time.time()
    vks = vks if isinstance(vks, int) else list(vks)
    AE = ref_import.ref("sgm.models.autoencoder.AutoencodingEngine")
    dd = vae_ddconfig(ch)
    ae = AE(encoder_config={"target": "sgm.modules.diffusionmodules.model.Encoder", "params": dd},
            decoder_config={"target": "sgm.modules.autoencoding.temporal_ae.VideoDecoder",
                            "params": dict(dd, video_kernel_size=vks)},
            loss_config={"target": "torch.nn.Identity"},
            regularizer_config={"target": "sgm.modules.autoencoding.regularizers.DiagonalGaussianRegularizer"}).eval()
    synth.fill_module_(ae, wseed, prefix=VAE_PREFIX)
    g = torch.Generator().manual_seed(iseed)
    z = torch.randn((b * T, 4, hw, hw), generator=g)
    with torch.no_grad():
        out = ae.decode(z, timesteps=T)
    sd = ae.state_dict()
    fx = dict(kind="video_decode", ddconfig=dd, T=T, weight_seed=wseed, key_prefix=VAE_PREFIX, z=z, output=out, video_kernel_size=vks,
              shapes={k: tuple(v.shape) for k, v in sd.items()})
    torch.save(fx, os.path.join(GOLD, name + ".pt"))
    print(f"{name}: out {tuple(out.shape)} absmax {out.abs().max():.4f} ({time.time() - t0:.1f}s)")


def gen_engine_encode(name, ch, n, hw, wseed=1, iseed=0):
    """AutoencodingEngine.encode (models/autoencoder.py:196-209) of the REFERENCE: raw moments (unregularized=True), the
    sampled posterior with the CPU generator seeded, and the noise that draw used."""
    t0 = time.time()
    AE = ref_import.ref("sgm.models.autoencoder.AutoencodingEngine")
    dd = vae_ddconfig(ch)
    ae = AE(encoder_config={"target": "sgm.modules.diffusionmodules.model.Encoder", "params": dd},
            decoder_config={"target": "sgm.modules.diffusionmodules.model.Decoder", "params": dd},
            loss_config={"target": "torch.nn.Identity"},
            regularizer_config={"target": "sgm.modules.autoencoding.regularizers.DiagonalGaussianRegularizer"}).eval()
    synth.fill_module_(ae, wseed, prefix=VAE_PREFIX)
    g = torch.Generator().manual_seed(iseed)
    x = torch.rand((n, 3, hw, hw), generator=g) * 2 - 1
    with torch.no_grad():
        moments, _ = ae.encode(x, unregularized=True)
        torch.manual_seed(4321)
        z_sampled, reg_log = ae.encode(x, return_reg_log=True)
        torch.manual_seed(4321)
        ref_noise = torch.randn(z_sampled.shape)
    sd = ae.state_dict()
    fx = dict(kind="engine_encode", ddconfig=dd, weight_seed=wseed, key_prefix=VAE_PREFIX, x=x, moments=moments,
              z_sampled=z_sampled, sample_noise=ref_noise, shapes={k: tuple(v.shape) for k, v in sd.items()})
    torch.save(fx, os.path.join(GOLD, name + ".pt"))
    print(f"{name}: moments {tuple(moments.shape)} absmax {moments.abs().max():.4f} z {tuple(z_sampled.shape)} ({time.time() - t0:.1f}s)")


def gen_schedule(name):
    Disc = ref_import.ref("sgm.modules.diffusionmodules.discretizer.EDMDiscretization")
    Scal = ref_import.ref("sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise")
    d = Disc(sigma_max=700.0)
    sig = {n: d(n, device="cpu") for n in (5, 25)}
    s = torch.tensor([700.0, 134.85, 15.59, 0.67815, 0.002])
    fx = dict(kind="schedule", sigmas=sig, scaling_in=s, scaling_out=[t.clone() for t in Scal()(s)])
    torch.save(fx, os.path.join(GOLD, name + ".pt"))
    print(name, sig[5])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--full", action="store_true", help="also the full-size stage-1 UNet forward (~3 min, 8 cores)")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(a.threads)
    jobs = {
        "schedule": lambda: gen_schedule("schedule"),
        "unet_tiny_s1": lambda: gen_unet("unet_tiny_s1", unet_cfg(1, 64), T=4, hw=16),
        "unet_tiny_s2_ioi": lambda: gen_unet("unet_tiny_s2_ioi", unet_cfg(2, 64), T=3, hw=8, iseed=3,
                                              ioi=torch.tensor([[0., 1., 0.], [0., 0., 1.]])),
        "sampler_tiny_s1": lambda: gen_sampler("sampler_tiny_s1", unet_cfg(1, 64), T=4, hw=16, steps=5,
                                               max_scale=2.5, stage=1),
        "sampler_tiny_s2": lambda: gen_sampler("sampler_tiny_s2", unet_cfg(2, 64), T=4, hw=8, steps=4,
                                               max_scale=2.0, stage=2, iseed=5),
        "vae_tiny": lambda: gen_vae("vae_tiny", 64, 2, 8),
        "vae_full_lat8": lambda: gen_vae("vae_full_lat8", 128, 1, 8, iseed=2),
        "vae_enc_tiny": lambda: gen_vae_encode("vae_enc_tiny", 64, 2, 64),
        "vae_enc_full_64": lambda: gen_vae_encode("vae_enc_full_64", 128, 1, 64, iseed=4),
        "engine_enc_tiny": lambda: gen_engine_encode("engine_enc_tiny", 64, 2, 64, iseed=8),
        "videodec_tiny": lambda: gen_video_decode("videodec_tiny", 64, 2, 3, 8),
        "videodec_full_lat8": lambda: gen_video_decode("videodec_full_lat8", 128, 1, 4, 8, iseed=3),
        # video_kernel_size = 3, the reference class's default: isotropic 3 x 3 x 3 time_stack / time_mix_conv (two clips of 3
        # frames at reduced width; one clip of 5 frames at full width, 128 x 128 pixels)
        "videodec_tiny_k3": lambda: gen_video_decode("videodec_tiny_k3", 64, 2, 3, 8, iseed=21, vks=3),
        "videodec_full_lat16_k3": lambda: gen_video_decode("videodec_full_lat16_k3", 128, 1, 5, 16, iseed=22, vks=3),
        "v02_tiny": lambda: gen_v02("v02_tiny", unet_cfg(2, 64), T=4, hw=8, steps=4, max_scale=2.0, iseed=6),
        "unet_s1_lat16": lambda: gen_unet("unet_s1_lat16", unet_cfg(1), T=4, hw=16, iseed=1),
        "unet_s2_lat16": lambda: gen_unet("unet_s2_lat16", unet_cfg(2), T=4, hw=16, iseed=2),
    }
    if a.full:
        jobs["unet_s1_full"] = lambda: gen_unet("unet_s1_full", unet_cfg(1), T=16, hw=64, iseed=7)
        # the benchmarked shape (BASELINE config[2]): B = 2x16 frames, latent 128x128, 17 input channels
        # (~15-25 min on 8 cores, ~45 GB peak)
        # bf16 error accumulation over the whole 25-step schedule at full width (~4 min)
        jobs["sampler_s1_w320_25step"] = lambda: gen_sampler("sampler_s1_w320_25step", unet_cfg(1), T=4, hw=16, steps=25,
                                                             max_scale=2.5, stage=1, iseed=9)
        # the stage-2 refine loop (re-noising blend + Euler-EDM + CFG, pipeline_i2v_eval_v02.py:103-135) over its whole
        # 25-step schedule at full width, 17 input channels (~5 min)
        jobs["v02_w320_25step"] = lambda: gen_v02("v02_w320_25step", unet_cfg(2), T=4, hw=16, steps=25, max_scale=2.0, iseed=11)
        # decode_first_stage of one frame at the two shipped resolutions, full-width decoder
        # BASELINE config 4's frame count through the full-width stage-2 UNet: temporal attention / Conv3d / 3-D GroupNorm over
        # 32 frames at every width (latent 16 x 16 so the reference finishes in about a minute)
        jobs["unet_s2_lat16_t32"] = lambda: gen_unet("unet_s2_lat16_t32", unet_cfg(2), T=32, hw=16, iseed=77, compact=True)
        # ... and the same 32 views at latent 64 x 64 (4096-token spatial attention, 262144-row GEMMs; ~6 min, ~25 GB)
        jobs["unet_s2_lat64_t32"] = lambda: gen_unet("unet_s2_lat64_t32", unet_cfg(2), T=32, hw=64, iseed=78, compact=True)
        # the temporal VideoDecoder (time_mode conv-only) at full width on a 4-frame clip of 256 x 256
        jobs["videodec_full_lat32"] = lambda: gen_video_decode("videodec_full_lat32", 128, 1, 4, 32, iseed=13)
        # encode_first_stage of one frame at the stage-2 resolution, full-width encoder (16384-token mid-block attention)
        jobs["vae_enc_full_1024"] = lambda: gen_vae_encode("vae_enc_full_1024", 128, 1, 1024, iseed=12, compact=True)
        jobs["vae_full_512"] = lambda: gen_vae("vae_full_512", 128, 1, 64, iseed=5, compact=True)
        jobs["vae_full_1024"] = lambda: gen_vae("vae_full_1024", 128, 1, 128, iseed=6, compact=True)
        jobs["unet_s2_full"] = lambda: gen_unet("unet_s2_full", unet_cfg(2), T=16, hw=128, iseed=8, compact=True)
        # BASELINE config 4 at its REAL size: 2 x 32 views, latent 128 x 128 (M = 1,048,576 token rows), the reference's own classes,
        # one clip per call (~20 min per clip on 8 cores, ~45 GB peak)
        jobs["unet_s2_full_t32"] = lambda: gen_unet("unet_s2_full_t32", unet_cfg(2), T=32, hw=128, iseed=88, compact=True, halves=True)
        # ---- round 6: the thing the bench times, at the size it times it (reference classes end to end) ----
        # stage 2, 16 views, latent 128 x 128, CFG 1 -> 2: the first 3 Euler-EDM steps of the 25-step schedule through
        # EulerEDMSampler.step_call (~15-25 min per step on 8 cores, ~45 GB peak); per-step wall time recorded
        jobs["sampler_s2_full_3step"] = lambda: gen_sampler_at_size("sampler_s2_full_3step", unet_cfg(2), T=16, hw=128, steps=25, n_run=3,
                                                                    max_scale=2.0, stage=2, iseed=31)
        # BASELINE config 1 exactly: stage 1, 16 views @ 512 x 512 (latent 64 x 64), all 25 steps, CFG 1 -> 2.5, then the reference
        # decode_first_stage + tensor2vid of frames 0 / 5 / 10 / 15 (~2 min per step on 8 cores)
        jobs["sampler_s1_full_25step"] = lambda: gen_sampler_at_size("sampler_s1_full_25step", unet_cfg(1), T=16, hw=64, steps=25, n_run=25,
                                                                     max_scale=2.5, stage=1, iseed=32,
                                                                     decode_frames=(0, 5, 10, 15))
        # BASELINE config 3 exactly, end to end: the stage-2 refine loop, 16 views, latent 128 x 128, all 25 steps (~4 h on 8 cores,
        # ~45 GB), the reference decode + tensor2vid of frames 0 and 8
        jobs["v02_s2_full_25step"] = lambda: gen_v02_at_size("v02_s2_full_25step", unet_cfg(2), T=16, hw=128, steps=25, max_scale=2.0,
                                                             keep=(0, 1, 2, 6, 12, 18, 24), iseed=41, decode_frames=(0, 8))
        # reference decode + tensor2vid of the final latents of the two full-width 25-step fixtures (4 frames of 128 x 128)
        jobs["sampler_s1_w320_25step_img"] = lambda: gen_decode_of("sampler_s1_w320_25step_img", "sampler_s1_w320_25step")
        jobs["v02_w320_25step_img"] = lambda: gen_decode_of("v02_w320_25step_img", "v02_w320_25step")
    jobs["compact_v02_s2_full_25step"] = lambda: compact_decoded("v02_s2_full_25step")      # (explicit: --only)
    for k, fn in jobs.items():
        if (a.only is None and not k.startswith("compact_")) or a.only == k:
            fn()


if __name__ == "__main__":
    main()
