"""The thing bench.py times, at the size it times it, against the REFERENCE's own classes.

Fixtures (oracle/gen_golden.py --full, the reference's EulerEDMSampler.step_call + LinearPredictionGuider + Denoiser +
OpenAIWrapper + VideoUNet on the build container's CPU in fp32):
  sampler_s2_full_3step   BASELINE config[2], the headline: stage 2, 16 views @ 1024^2 (latent 128^2, 17 input channels, CFG
                          batch 32), the first 3 steps of the 25-step schedule
  sampler_s1_full_25step  BASELINE config[0]/[1] exactly: stage 1, 16 views @ 512^2 (latent 64^2), all 25 steps, then the
                          reference's decode_first_stage + tensor2vid of frames 0 / 5 / 10 / 15
both run here through the launch mode the benchmark times: hi3d_hip.fused_step (one kernel sequence per step) replayed as a HIP
graph, with the two CFG halves on two streams where bench.py's headline has them.

What is compared per step is the guided denoised estimate D_i = x_i - sigma_i (x_{i+1} - x_i) / (sigma_{i+1} - sigma_i), NOT
the state: at sigma_0 = 700 the state is ~2800 in magnitude and a step moves it by 0.22 (x - D), so a relative max-abs
comparison of states passes with a garbage network (|D| ~ 5: the network's whole contribution is 4e-4 of the state).  D is
the network's output after denoiser scaling and guidance -- what the reference's `EDMSampler.denoise` returns
(sampling.py:54-57) -- and recovering it from two fp32 states costs 2.4e-4 * 2800 / 0.22 / 2800 ~ 1e-3 of |D| at most.

Tolerances (bf16 storage, fp32 accumulate vs fp32): per step max-abs error of D <= 2.5e-2 x max |D_ref|, cosine >= 0.999;
final latents cosine >= 0.999; decoded frames PSNR >= 35 dB (peak = max |ref|), uint8 frames: stated below.
"""
import math
import os
import tempfile

import pytest
import torch
import torch.nn.functional as F
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
D_TOL, D_COS = 2.5e-2, 0.999
D_TOL_TRAJ = 3.5e-2        # denoised estimate taken on the product's own trajectory after >= 3 steps of the full-size stage-2 loop


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def cos(a, b):
    return F.cosine_similarity(a.float().cpu().flatten(), b.float().cpu().flatten(), dim=0).item()


def load(name):
    path = os.path.join(GOLD, name + ".pt")
    if not os.path.exists(path):
        pytest.skip(f"{name}.pt not generated (oracle/gen_golden.py --full --only {name})")
    return torch.load(path, weights_only=False)


def seeded_inputs(fx):
    """x0 / c / uc of the fixture, re-drawn from its seed and pinned by its probe"""
    from hi3d_hip import synth
    x0, c, uc = synth.synth_conditioning(fx["T"], fx["hw"], fx["hw"], stage=fx["stage"], seed=fx["input_seed"],
                                         adm_in=fx["cfg"]["adm_in_channels"])
    pr = fx["x0_probe"]
    assert torch.equal(x0.flatten()[:16], pr["head"]) and abs(float(x0.double().sum()) - pr["sum"]) < 1e-6 * pr["abs_sum"], \
        "seeded inputs are not the ones the golden was generated from"
    return x0, c, uc


def reference_states(fx, x0):
    """the reference's states x_0 .. x_n rebuilt from x0 and its denoised estimates (Euler update, sampling.py:93-107,
    sampling_utils.py:34); pinned against the fp32 state the fixture holds"""
    sig = fx["sigmas"]
    xs = [x0 * torch.sqrt(1.0 + sig[0] ** 2.0)]                       # prepare_sampling_loop, sampling.py:46
    for i in range(fx["n_run"]):
        D = fx["denoised_f16"][i].float()
        xs.append(xs[-1] + (sig[i + 1] - sig[i]) * (xs[-1] - D) / sig[i])
    assert relerr(xs[-1], fx["last_state"]) < 1e-3
    return xs


def denoised_from_states(x, x_next, sig, i):
    """D_i from two consecutive states of an Euler step (gamma = 0)"""
    if float(sig[i + 1]) == 0.0:
        return x_next
    r = (sig[i + 1] / sig[i]).double()
    return ((x_next.double() - x.double() * r) / (1.0 - r)).float()


def make_sampler(fx, dev):
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    return EulerEDMSampler(
        num_steps=fx["steps"], device=dev,
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"num_frames": fx["T"], "max_scale": fx["max_scale"], "min_scale": 1.0}})


def test_stage2_headline_sampler_steps_through_graph_and_two_streams_match_reference(dev):
    """The benchmark's timed region itself: one sampler step = CFG batch build + denoiser scaling + VideoUNet at B = 2 x 16
    frames, latent 128 x 128 + guidance + Euler update (guiders.py:78-99, denoiser.py:23-39, wrappers.py:23-34,
    video_model.py:442-501, sampling.py:93-107), the first 3 steps of the 25-step schedule, against the reference classes.
    Pass 0 runs them as the product does from a cold start (two eager steps, the third captured and replayed); pass 1 repeats
    the same three steps, now ALL as replays of the captured graph -- the launch mode of every timed bench step."""
    from conftest import synth_unet
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    fx = load("sampler_s2_full_3step")
    T, hw = fx["T"], fx["hw"]
    assert (fx["stage"], T, hw, fx["steps"], fx["max_scale"]) == (2, 16, 128, 25, 2.0)
    x0, c, uc = seeded_inputs(fx)
    ref_x = reference_states(fx, x0)
    unet = synth_unet(fx, dev)
    rt = unet.runtime(dev)
    model = OpenAIWrapper(unet)
    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = make_sampler(fx, dev)
    cd, ucd = {k: v.to(dev) for k, v in c.items()}, {k: v.to(dev) for k, v in uc.items()}
    extra = dict(image_only_indicator=torch.zeros(2, T, device=dev), num_video_frames=T)

    def denoiser(inp, sigma, cc):
        return den(model, inp, sigma, cc, **extra)

    finals = []
    for run in (0, 1):
        x, s_in, sigmas, num_sigmas, cond, ucond = sampler.prepare_sampling_loop(x0.clone().to(dev), cd, ucd)
        assert torch.allclose(sigmas.cpu(), fx["sigmas"], rtol=1e-6, atol=0)
        for i in range(fx["n_run"]):
            stepper = rt.steppers.get((T, hw, hw))
            if run == 1:
                assert stepper is not None and stepper.graph is not None, "pass 1 must be graph replays"
            x_next = sampler.step_call(denoiser, x, i, s_in, sigmas, num_sigmas, cond, ucond)
            D = denoised_from_states(x, x_next, sigmas, i)
            ref_D = fx["denoised_f16"][i].float()
            rel, cs, xrel = relerr(D, ref_D), cos(D, ref_D), relerr(x_next, ref_x[i + 1])
            print(f"stage-2 headline step {i} (pass {run}: {'graph replay' if rt.steppers[(T, hw, hw)].graph is not None and (run or i >= 2) else 'eager'}): "
                  f"denoised rel {rel:.4f} cos {cs:.6f}; state rel {xrel:.2e}")
            assert rel < D_TOL and cs > D_COS, f"pass {run} step {i}: denoised rel {rel:.4f} cos {cs:.6f}"
            assert xrel < 1e-4            # (what a state comparison can see at sigma ~ 500: the network at 4e-4 of the state)
            x = x_next
        stepper = rt.steppers[(T, hw, hw)]
        assert stepper.graph is not None, "the third step must have been captured into a HIP graph"
        assert rt.last_forward_two_stream, "the headline shape runs its two CFG halves on two streams"
        finals.append(x.clone())
    # eager (steps 0, 1) and replayed steps issue the same kernels on the same data
    assert relerr(finals[1], finals[0]) < 1e-6


def _stage1_model(dev, T):
    """create_model(inference-v01.yaml) at FULL width (UNet 320 .. 1280 channels, VAE ch 128) with the CLIP towers reduced (the
    conditioning of this test is synthetic), the fixture's seeded weights in the UNet and the first stage"""
    from conftest import shrink_conditioner, synth_fill_cached
    from sgm.util import ParamTree
    from vtdm.model import create_model
    y = shrink_conditioner(yaml.safe_load(open(os.path.join(ROOT, "hi3d-official_amd", "configs", "inference-v01.yaml"))))
    P = y["model"]["params"]
    P["sampler_config"]["params"]["verbose"] = False
    assert P["num_samples"] == T and P["sampler_config"]["params"]["num_steps"] == 25
    assert P["sampler_config"]["params"]["guider_config"]["params"]["max_scale"] == 2.5
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as fh:
        yaml.safe_dump(y, fh)
    ParamTree.skip_init = True
    try:
        model = create_model(fh.name)
    finally:
        ParamTree.skip_init = False
        os.unlink(fh.name)
    model = model.to(dev)
    model.sampler.device = dev
    synth_fill_cached(model.model.diffusion_model, "model.diffusion_model.", 1, dev)
    synth_fill_cached(model.first_stage_model, "first_stage_model.", 1, dev)
    return model


def test_stage1_clip_full_size_25_steps_and_decode_match_reference_end_to_end(dev):
    """BASELINE config 1 exactly, end to end from reference classes (VERDICT r5 missing 2 and 4): create_model(inference-v01.yaml)
    -> the 25 Euler-EDM + CFG 1 -> 2.5 steps at T = 16, latent 64 x 64, full width (pipeline_i2v_eval_v01.py:85-92,
    sampling.py:109-147) -> decode_first_stage (diffusion.py:117-135) -> '(b t) c h w -> b c t h w' -> tensor2vid
    (vtdm/util.py:13-21); every step's denoised estimate, the final latents, the decoded frames 0 / 5 / 10 / 15 and their uint8
    export against the reference's."""
    from vtdm.util import tensor2vid
    fx = load("sampler_s1_full_25step")
    T, hw, n = fx["T"], fx["hw"], fx["n_run"]
    assert (fx["stage"], T, hw, fx["steps"], n, fx["max_scale"]) == (1, 16, 64, 25, 25, 2.5)
    x0, c, uc = seeded_inputs(fx)
    ref_x = reference_states(fx, x0)
    model = _stage1_model(dev, T)
    rt = model.model.diffusion_model.runtime(dev)
    cd, ucd = {k: v.to(dev) for k, v in c.items()}, {k: v.to(dev) for k, v in uc.items()}
    extra = {"image_only_indicator": torch.zeros(2, T, device=dev), "num_video_frames": T}

    def denoiser(inp, sigma, cc):                          # pipeline_i2v_eval_v01.py:85-88
        return model.denoiser(model.model, inp, sigma, cc, **extra)

    sampler = model.sampler
    x, s_in, sigmas, num_sigmas, cond, ucond = sampler.prepare_sampling_loop(x0.clone().to(dev), cd, ucd)
    worst = (0.0, 1.0)
    for i in sampler.get_sigma_gen(num_sigmas):
        x_next = sampler.step_call(denoiser, x, i, s_in, sigmas, num_sigmas, cond, ucond)
        D, ref_D = denoised_from_states(x, x_next, sigmas, i), fx["denoised_f16"][i].float()
        rel, cs = relerr(D, ref_D), cos(D, ref_D)
        worst = (max(worst[0], rel), min(worst[1], cs))
        assert rel < D_TOL and cs > D_COS, f"step {i}: denoised rel {rel:.4f} cos {cs:.6f}"
        x = x_next
    stepper = rt.steppers[(T, hw, hw)]
    assert stepper.graph is not None and num_sigmas - 1 == 25
    lat_rel, lat_cos = relerr(x, fx["last_state"]), cos(x, fx["last_state"])
    print(f"stage-1 config-1 clip, 25 steps at 16 x 64^2 through the fused graph step: worst denoised rel {worst[0]:.4f} cos {worst[1]:.6f}; "
          f"final latents rel {lat_rel:.4f} cos {lat_cos:.6f}")
    assert lat_rel < D_TOL and lat_cos > 0.999
    # the one-call form the pipeline uses (v01.py:92) replays the same graph on the same inputs
    whole = sampler(denoiser, x0.clone().to(dev), cond=cd, uc=ucd)
    assert relerr(whole, x) < 1e-6
    # ---- decode_first_stage -> rearrange -> tensor2vid
    images = model.decode_first_stage(whole)                                        # v01.py:94
    assert images.shape == (T, 3, 8 * hw, 8 * hw)
    fr = fx["decode_frames"]
    img, ref_img = images[fr].float().cpu(), fx["decoded_f16"].float()
    mse = ((img - ref_img) ** 2).mean().item()
    psnr = 10 * math.log10(ref_img.abs().max().item() ** 2 / mse)
    img_rel = relerr(img, ref_img)
    video = images.reshape(1, T, 3, 8 * hw, 8 * hw).permute(0, 2, 1, 3, 4)          # '(b t) c h w -> b c t h w', v01.py:96
    frames = tensor2vid(video.clone())
    assert len(frames) == T and frames[0].shape == (8 * hw, 8 * hw, 3) and frames[0].dtype.name == "uint8"
    got_u8 = torch.stack([torch.from_numpy(frames[f]) for f in fr]).int()
    d8 = (got_u8 - fx["decoded_u8"].int()).abs()
    print(f"stage-1 config-1 clip decoded (frames {fr}): image rel {img_rel:.4f} PSNR {psnr:.1f} dB; uint8 frames: max |diff| {int(d8.max())}, "
          f"mean {d8.float().mean():.3f}, {100.0 * (d8 > 2).float().mean():.2f} % of the values differ by more than 2")
    assert psnr > 35.0 and img_rel < 8e-2
    assert d8.float().mean() < 1.5 and (d8 > 8).float().mean() < 1e-2     # (measured: PSNR 51.3 dB, max 10, mean 0.91, 6.5 % of the values off by more than 2)


@pytest.mark.parametrize("name,src", [("sampler_s1_w320_25step_img", "sampler_s1_w320_25step"), ("v02_w320_25step_img", "v02_w320_25step")])
def test_full_width_25_step_latents_decode_to_the_reference_images(dev, name, src):
    """The images end of the two full-width 25-step trajectories (4 frames of 128 x 128): the REFERENCE's final latents through
    the product's decode_first_stage arithmetic + tensor2vid against the reference's decode + tensor2vid of the same latents
    (diffusion.py:117-135, vtdm/util.py:13-21)."""
    from conftest import synth_fill_cached
    from sgm.models.autoencoder import AutoencoderKL
    from vtdm.util import tensor2vid
    fx, z = load(name), load(src)["output"]
    assert torch.equal(z.flatten()[:16], fx["z_head"])
    ae = AutoencoderKL(embed_dim=4, ddconfig=fx["ddconfig"], lossconfig={"target": "torch.nn.Identity"}).to(dev)
    synth_fill_cached(ae, fx["key_prefix"], fx["weight_seed"], dev)
    img = ae.decode(z.to(dev) / 0.18215).float().cpu()
    ref = fx["decoded"]
    psnr = 10 * math.log10(ref.abs().max().item() ** 2 / ((img - ref) ** 2).mean().item())
    n = img.shape[0]
    frames = tensor2vid(img.reshape(1, n, *img.shape[1:]).permute(0, 2, 1, 3, 4).clone())
    d8 = (torch.stack([torch.from_numpy(f) for f in frames]).int() - fx["decoded_u8"].int()).abs()
    print(f"{name}: image rel {relerr(img, ref):.4f} PSNR {psnr:.1f} dB; uint8 max |diff| {int(d8.max())} mean {d8.float().mean():.3f}")
    assert relerr(img, ref) < 4e-2 and psnr > 35.0 and int(d8.max()) <= 8 and d8.float().mean() < 1.0      # (measured: max 5, mean 0.52 -- astype(uint8) truncates, so sub-LSB differences flip values)


def test_stage2_refine_loop_full_size_25_steps_and_decode_match_reference_end_to_end(dev):
    """BASELINE config 3 exactly, end to end from reference classes: the stage-2 refine loop of pipeline_i2v_eval_v02.py:103-135
    -- re-noising blend with the per-frame latents, Euler-EDM, CFG 1 -> 2 -- at 16 views, latent 128 x 128, all 25 steps, through
    the product loop (hi3d_v02_blend + the fused graph-replayed two-stream step), then decode_first_stage one frame per call
    (en_and_decode_n_samples_a_time = 1) and tensor2vid, against `v02_s2_full_25step` (oracle/gen_golden.py: ~4 h of the reference
    on 8 cores).  The guided denoised estimate of the kept steps, the final latents, the decoded frames."""
    from conftest import synth_fill_cached, synth_unet
    from hi3d_hip import ops, synth
    from hi3d_hip.pipelines import v02_alpha
    from sgm.models.autoencoder import AutoencoderKL
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    from vtdm.util import tensor2vid
    fx = load("v02_s2_full_25step")
    T, hw, steps = fx["T"], fx["hw"], fx["steps"]
    assert (T, hw, steps, fx["max_scale"]) == (16, 128, 25, 2.0)
    init, c, uc = synth.synth_conditioning(T, hw, hw, stage=2, seed=fx["input_seed"], adm_in=fx["cfg"]["adm_in_channels"])
    pr = fx["init_probe"]
    assert torch.equal(init.flatten()[:16], pr["head"]) and abs(float(init.double().sum()) - pr["sum"]) < 1e-6 * pr["abs_sum"]
    g = torch.Generator().manual_seed(fx["input_seed"] + 100)
    z_frames = torch.cat([torch.randn((1, 4, hw, hw), generator=g) * 0.8 for _ in range(T)], 0)
    assert torch.equal(z_frames[0].flatten()[:16], fx["z_probe"]["head"])
    unet = synth_unet(dict(fx, stage=2), dev)
    rt = unet.runtime(dev)
    model = OpenAIWrapper(unet)
    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = make_sampler(dict(fx, stage=2), dev)
    cd, ucd = {k: v.to(dev) for k, v in c.items()}, {k: v.to(dev) for k, v in uc.items()}
    extra = dict(image_only_indicator=torch.zeros(2, T, device=dev), num_video_frames=T)

    def denoiser(inp, sigma, cc):
        return den(model, inp, sigma, cc, **extra)

    sigmas = sampler.discretization(sampler.num_steps, device=dev)
    sig_host = sigmas.float().cpu().tolist()
    num_sigmas = len(sigmas)
    initd, zd = init.to(dev).contiguous(), z_frames.to(dev).contiguous()
    latents = (initd * math.sqrt(1.0 + sig_host[0] ** 2)).contiguous()
    s_in = latents.new_ones([T])
    kept = {k: n for n, k in enumerate(fx["kept_steps"])}
    worst = (0.0, 1.0)
    for i in sampler.get_sigma_gen(num_sigmas):
        ops.v02_blend(latents, initd, zd, v02_alpha(i, sampler.num_steps), sig_host[i])          # v02.py:127-131
        x_next = sampler.step_call(denoiser, latents, i, s_in, sigmas, num_sigmas, cd, ucd).contiguous()
        if i in kept:
            D, ref_D = denoised_from_states(latents, x_next, sigmas, i), fx["denoised_f16"][kept[i]].float()
            rel, cs = relerr(D, ref_D), cos(D, ref_D)
            worst = (max(worst[0], rel), min(worst[1], cs))
            print(f"stage-2 refine loop at full size, step {i}: denoised rel {rel:.4f} cos {cs:.6f}")
            # steps 0 - 2: the single-step bound (no drift yet); later steps are taken on the product's OWN trajectory, whose state
            # has drifted from the reference's by then: the trajectory bound D_TOL_TRAJ (measured: 2.2 - 2.5e-2 at steps 6 - 24)
            tol = D_TOL if i < 3 else D_TOL_TRAJ
            assert rel < tol and cs > D_COS, f"step {i}: denoised rel {rel:.4f} cos {cs:.6f} (bound {tol})"
        latents = x_next
    assert rt.steppers[(T, hw, hw)].graph is not None and rt.last_forward_two_stream
    lat_rel, lat_cos = relerr(latents, fx["output"]), cos(latents, fx["output"])
    print(f"stage-2 refine loop at full size (config 3), 25 steps: worst denoised rel {worst[0]:.4f} cos {worst[1]:.6f}; "
          f"final latents rel {lat_rel:.4f} cos {lat_cos:.6f}")
    assert lat_rel < D_TOL_TRAJ and lat_cos > 0.999
    # ---- decode_first_stage (one frame per call) -> tensor2vid, frames of the fixture
    ae = AutoencoderKL(embed_dim=4, ddconfig=dict(attn_type="vanilla-xformers", double_z=True, z_channels=4, resolution=256, in_channels=3,
                                                  out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0),
                       lossconfig={"target": "torch.nn.Identity"}).to(dev)
    synth_fill_cached(ae, "first_stage_model.", 1, dev)
    fr = fx["decode_frames"]
    img = torch.cat([ae.decode(latents[f:f + 1] / 0.18215) for f in fr], 0).float().cpu()
    ref_img = fx["decoded_f16"].float()
    c0, cs_ = fx["decoded_crop"]                      # (the fixture keeps the fp16 image for a centre crop, the uint8 frames whole)
    img_c = img[..., c0:c0 + cs_, c0:c0 + cs_]
    psnr = 10 * math.log10(ref_img.abs().max().item() ** 2 / ((img_c - ref_img) ** 2).mean().item())
    frames = tensor2vid(img.reshape(1, len(fr), *img.shape[1:]).permute(0, 2, 1, 3, 4).clone())
    d8 = (torch.stack([torch.from_numpy(f) for f in frames]).int() - fx["decoded_u8"].int()).abs()
    print(f"stage-2 refine loop at full size decoded (frames {fr} at 1024 x 1024): image rel {relerr(img_c, ref_img):.4f} PSNR {psnr:.1f} dB; "
          f"uint8 max |diff| {int(d8.max())}, mean {d8.float().mean():.3f}")
    assert psnr > 35.0 and d8.float().mean() < 1.5 and (d8 > 8).float().mean() < 1e-2


@pytest.mark.parametrize("mode", ["fp8qk", "fp8"])
def test_stage2_headline_sampler_steps_fp8_attention_match_reference(dev, monkeypatch, mode):
    """BASELINE config 5 ("fp8 MFMA attention + bf16 conv") on the same three headline steps, through the same launch mode (fused
    step, HIP-graph replay, two streams): every spatial attention with the score product (fp8qk) or both products (fp8) on the e4m3
    matrix path, against the reference's fp32 classes.  Reduced precision, separately stated bounds on the guided denoised
    estimate (the single-forward bounds of these modes, tests/test_at_size_gpu.py): fp8qk 8e-2 / cosine >= 0.998, fp8 1.2e-1 /
    cosine >= 0.995 (measured: 2.0 - 2.4e-2 / 0.99990 for both -- at the level of the network's output the e4m3 attention is not
    distinguishable from the bf16 one); the runtime's dispatch flags are asserted (the fp8 kernels are the ones that run)."""
    from conftest import synth_unet
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    monkeypatch.setenv("HI3D_ATTN_FP8QK", "1" if mode == "fp8qk" else "0")
    monkeypatch.setenv("HI3D_ATTN_FP8", "1" if mode == "fp8" else "0")
    fx = load("sampler_s2_full_3step")
    T, hw = fx["T"], fx["hw"]
    x0, c, uc = seeded_inputs(fx)
    unet = synth_unet(fx, dev)
    rt = unet.runtime(dev)
    assert rt.attn_fp8qk == (mode == "fp8qk") and rt.attn_fp8 == (mode == "fp8")
    model = OpenAIWrapper(unet)
    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = make_sampler(fx, dev)
    cd, ucd = {k: v.to(dev) for k, v in c.items()}, {k: v.to(dev) for k, v in uc.items()}
    extra = dict(image_only_indicator=torch.zeros(2, T, device=dev), num_video_frames=T)

    def denoiser(inp, sigma, cc):
        return den(model, inp, sigma, cc, **extra)

    tol, cmin = {"fp8qk": (8e-2, 0.998), "fp8": (1.2e-1, 0.995)}[mode]
    for run in (0, 1):                                   # (pass 1: all three steps as graph replays)
        x, s_in, sigmas, num_sigmas, cond, ucond = sampler.prepare_sampling_loop(x0.clone().to(dev), cd, ucd)
        for i in range(fx["n_run"]):
            x_next = sampler.step_call(denoiser, x, i, s_in, sigmas, num_sigmas, cond, ucond)
            D, ref_D = denoised_from_states(x, x_next, sigmas, i), fx["denoised_f16"][i].float()
            rel, cs = relerr(D, ref_D), cos(D, ref_D)
            if run:
                print(f"stage-2 headline step {i}, {mode} attention (graph replay): denoised rel {rel:.4f} cos {cs:.6f}")
            assert rel < tol and cs > cmin, f"{mode} pass {run} step {i}: denoised rel {rel:.4f} cos {cs:.6f}"
            x = x_next
    assert rt.steppers[(T, hw, hw)].graph is not None and rt.last_forward_two_stream
