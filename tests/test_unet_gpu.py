"""End-to-end parity of the HIP VideoUNet / sampler against
  (a) golden outputs of the REFERENCE classes (tests/golden, CPU fp32), and
  (b) the CPU oracle on fresh seeded inputs at other sizes.
The HIP path stores activations in bf16 (8-bit mantissa) through ~100 sequential layers;
the oracle is fp32.  Tolerances (stated, north_star "within a stated fp tolerance"):
  UNet output  : max-abs error <= 4e-2 x max-abs reference, cosine >= 0.9995
  sampler      : per-step latents cosine >= 0.999, max-abs rel <= 6e-2
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)


def stats(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    rel = ((a - b).abs().max() / b.abs().max()).item()
    cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
    return rel, cos


def build_unet(fx, dev):
    from conftest import synth_unet          # seeded weights drawn once per session, a fresh module per call
    return synth_unet(fx, dev)


@pytest.mark.parametrize("name", ["unet_tiny_s1", "unet_tiny_s2_ioi", "unet_s1_lat16", "unet_s2_lat16"])
def test_unet_matches_reference_golden(dev, name):
    fx = load(name)
    m = build_unet(fx, dev)
    i = {k: v.to(dev) for k, v in fx["inputs"].items()}
    out = m(i["x"], i["timesteps"], context=i["context"], y=i["y"], num_video_frames=fx["T"],
            image_only_indicator=i["image_only_indicator"])
    assert out.shape == fx["output"].shape and out.dtype == i["x"].dtype
    rel, cos = stats(out, fx["output"])
    print(f"{name}: rel {rel:.4f} cos {cos:.6f}")
    assert rel < 4e-2 and cos > 0.9995


def test_unet_full_size_stage1_matches_reference_golden(dev):
    """BASELINE config[1] shape: B = 2x16 frames, latent 64x64, full width (1.52 B params)."""
    path = os.path.join(GOLD, "unet_s1_full.pt")
    if not os.path.exists(path):
        pytest.skip("full-size golden not generated")
    fx = load("unet_s1_full")
    m = build_unet(fx, dev)
    i = {k: v.to(dev) for k, v in fx["inputs"].items()}
    out = m(i["x"], i["timesteps"], context=i["context"], y=i["y"], num_video_frames=fx["T"],
            image_only_indicator=i["image_only_indicator"])
    rel, cos = stats(out, fx["output"])
    print(f"unet_s1_full: rel {rel:.4f} cos {cos:.6f}")
    assert rel < 4e-2 and cos > 0.9995


@pytest.mark.parametrize("name", ["sampler_tiny_s1", "sampler_tiny_s2"])
def test_sampler_matches_reference_golden(dev, name):
    """The reference's own call pattern: EulerEDMSampler(denoiser closure, x, cond, uc) with
    Denoiser + OpenAIWrapper + VideoUNet, compared per step with the reference trajectory."""
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    fx = load(name)
    T = fx["T"]
    model = OpenAIWrapper(build_unet(fx, dev))
    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = EulerEDMSampler(
        num_steps=fx["steps"], device=dev,
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": fx["max_scale"], "min_scale": 1.0}})
    c = {k: v.to(dev) for k, v in fx["c"].items()}
    uc = {k: v.to(dev) for k, v in fx["uc"].items()}
    extra = dict(image_only_indicator=torch.zeros(2, T, device=dev), num_video_frames=T)

    def denoiser(inp, sigma, cc):
        return den(model, inp, sigma, cc, **extra)

    x, s_in, sigmas, num_sigmas, cond, ucond = sampler.prepare_sampling_loop(fx["x0"].clone().to(dev), c, uc)
    assert torch.allclose(sigmas.cpu(), fx["sigmas"])
    for i in sampler.get_sigma_gen(num_sigmas):
        x = sampler.step_call(denoiser, x, i, s_in, sigmas, num_sigmas, cond, ucond)
        rel, cos = stats(x, fx["traj"][i])
        print(f"{name} step {i}: rel {rel:.4f} cos {cos:.6f}")
        assert rel < 6e-2 and cos > 0.999
    whole = sampler(denoiser, fx["x0"].clone().to(dev), cond=c, uc=uc)
    assert torch.equal(whole, x), "step_call loop and __call__ must be the same computation (and deterministic)"


def test_unet_vs_oracle_other_shape(dev):
    """Fresh seeds, non-square latent, T=6 (not a power of two), one image-only frame."""
    from hi3d_hip import synth
    from oracle import hi3d_oracle as O
    fx = load("unet_tiny_s1")
    cfg, T, H, W = fx["cfg"], 6, 8, 24
    m = build_unet(fx, dev)
    sd = {fx["key_prefix"] + k: v.float().cpu() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    x = torch.randn((2 * T, 8, H, W), generator=g)
    ts = 0.25 * torch.log(torch.rand((2 * T,), generator=g) * 100 + 0.01)
    ctx, y = torch.randn((2, 1, 1024), generator=g), torch.randn((2, 768), generator=g)
    ioi = torch.zeros(2, T); ioi[1, 2] = 1.0
    ref = O.video_unet(sd, cfg, x, ts, ctx, y, T, ioi, prefix=fx["key_prefix"])
    out = m(x.to(dev), ts.to(dev), context=ctx.to(dev), y=y.to(dev), num_video_frames=T, image_only_indicator=ioi.to(dev))
    rel, cos = stats(out, ref)
    print(f"oracle shape test: rel {rel:.4f} cos {cos:.6f}")
    assert rel < 4e-2 and cos > 0.9995


def test_stage2_refine_loop_matches_reference_golden(dev):
    """pipeline_i2v_eval_v02.py:103-135 through hi3d_hip.pipelines.stage2_refine
    (fused v02 blend kernel + sampler.step_call) vs the reference's loop."""
    from types import SimpleNamespace
    from hi3d_hip.pipelines import stage2_refine
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    fx = load("v02_tiny")
    T = fx["T"]
    model = SimpleNamespace(
        device=dev, model=OpenAIWrapper(build_unet(fx, dev)),
        denoiser=Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}),
        sampler=EulerEDMSampler(
            num_steps=fx["steps"], device=dev,
            discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
            guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                           "params": {"num_frames": T, "max_scale": fx["max_scale"], "min_scale": 1.0}}))
    c = {k: v.to(dev) for k, v in fx["c"].items()}
    uc = {k: v.to(dev) for k, v in fx["uc"].items()}
    out = stage2_refine(model, None, c, uc, init_noise=fx["init"], decode=False, z_frames=fx["z_frames"])
    rel, cos = stats(out, fx["output"])
    print(f"v02 loop: rel {rel:.4f} cos {cos:.6f}")
    assert rel < 6e-2 and cos > 0.999


def test_unet_32_views_vs_oracle(dev):
    """BASELINE config 4 shape family: T = 32 views (temporal attention over 32 frames, 3-D
    GroupNorm / Conv3d over 32 frames), reduced width and latent so the oracle runs in seconds."""
    from oracle import hi3d_oracle as O
    fx = load("unet_tiny_s2_ioi")
    cfg, T, H, W = fx["cfg"], 32, 8, 8
    m = build_unet(fx, dev)
    sd = {fx["key_prefix"] + k: v.float().cpu() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(21)
    x = torch.randn((2 * T, cfg["in_channels"], H, W), generator=g)
    ts = 0.25 * torch.log(torch.rand((2 * T,), generator=g) * 100 + 0.01)
    ctx, y = torch.randn((2, 1, 1024), generator=g), torch.randn((2, cfg["adm_in_channels"]), generator=g)
    ioi = torch.zeros(2, T)
    ref = O.video_unet(sd, cfg, x, ts, ctx, y, T, ioi, prefix=fx["key_prefix"])
    out = m(x.to(dev), ts.to(dev), context=ctx.to(dev), y=y.to(dev), num_video_frames=T, image_only_indicator=ioi.to(dev))
    rel, cos = stats(out, ref)
    print(f"32 views: rel {rel:.4f} cos {cos:.6f}")
    assert rel < 4e-2 and cos > 0.9995


def _sampler_stack(fx, dev):
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    T = fx["T"]
    unet = build_unet(fx, dev)
    model = OpenAIWrapper(unet)
    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = EulerEDMSampler(
        num_steps=fx["steps"], device=dev,
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": fx["max_scale"], "min_scale": 1.0}})
    extra = dict(image_only_indicator=torch.zeros(2, T, device=dev), num_video_frames=T)

    def denoiser(inp, sigma, cc):
        return den(model, inp, sigma, cc, **extra)
    return unet, sampler, denoiser


def test_two_clips_through_one_model_do_not_share_conditioning(dev):
    """pipeline_i2v_eval_v01.py:106-118 runs clip after clip through one model.  Round 1 cached the
    cross-attention vectors on context.data_ptr(): a second clip whose context landed on the recycled
    address silently reused the first clip's.  Clip 2 (different crossattn / vector / concat, the first
    clip's tensors freed) must match the ORACLE on its own conditioning, on both step paths."""
    from hi3d_hip import synth
    from oracle import hi3d_oracle as O
    fx = load("sampler_tiny_s1")
    T, cfg = fx["T"], fx["cfg"]
    unet, sampler, denoiser = _sampler_stack(fx, dev)
    sd = {fx["key_prefix"] + k: v.float().cpu() for k, v in unet.state_dict().items()}
    h = fx["x0"].shape[-1]

    def oracle_step0(x0, c, uc, sigmas):
        """first Euler step on the CPU oracle (denoiser.py:23-39, guiders.py:78-99, sampling.py:93-107)"""
        s0, s1 = float(sigmas[0]), float(sigmas[1])
        x = x0 * math.sqrt(1 + s0 * s0)
        c_skip, c_out, c_in = 1 / (s0 * s0 + 1), -s0 / math.sqrt(s0 * s0 + 1), 1 / math.sqrt(s0 * s0 + 1)
        xin = torch.cat([torch.cat([x, x]) * c_in, torch.cat([uc["concat"], c["concat"]])], 1)
        net = O.video_unet(sd, cfg, xin, torch.full((2 * T,), 0.25 * math.log(s0)), torch.cat([uc["crossattn"], c["crossattn"]]),
                           torch.cat([uc["vector"], c["vector"]]), T, torch.zeros(2, T), prefix=fx["key_prefix"])
        den = net * c_out + torch.cat([x, x]) * c_skip
        du, dc = den.chunk(2)
        scale = torch.linspace(1.0, fx["max_scale"], T)[:, None, None, None]
        d = du + scale * (dc - du)
        return x + (s1 - s0) * (x - d) / s0

    import math
    for fused in ("1", "0"):
        os.environ["HI3D_FUSED_STEP"] = fused
        try:
            for clip_seed in (100, 200, 300):
                x0, c, uc = synth.synth_conditioning(T, h, h, stage=1, seed=clip_seed, adm_in=cfg["adm_in_channels"])
                cd = {k: v.to(dev) for k, v in c.items()}
                ucd = {k: v.to(dev) for k, v in uc.items()}
                x, s_in, sigmas, num_sigmas, cond, ucond = sampler.prepare_sampling_loop(x0.clone().to(dev), cd, ucd)
                got = sampler.step_call(denoiser, x, 0, s_in, sigmas, num_sigmas, cond, ucond)
                ref = oracle_step0(x0, c, uc, sigmas.cpu())
                rel, cos = stats(got, ref)
                print(f"fused={fused} clip {clip_seed}: rel {rel:.4f} cos {cos:.6f}")
                assert rel < 6e-2 and cos > 0.999
                del cd, ucd, cond, ucond, x, got            # free the clip's tensors: the allocator may recycle them
        finally:
            os.environ.pop("HI3D_FUSED_STEP", None)


def test_fused_graph_step_equals_generic_step(dev):
    """The one-graph-replay step (hi3d_hip/fused_step.py) against the generic per-op path
    (HI3D_FUSED_STEP=0: guider cat -> Denoiser math -> OpenAIWrapper cat -> VideoUNet -> guider -> Euler)
    on the same model: same kernels inside the UNet, so the two agree to fp32 rounding of the
    elementwise tail; and the fused path must really have been taken and captured."""
    fx = load("sampler_tiny_s2")
    unet, sampler, denoiser = _sampler_stack(fx, dev)
    c = {k: v.to(dev) for k, v in fx["c"].items()}
    uc = {k: v.to(dev) for k, v in fx["uc"].items()}
    outs = {}
    for fused in ("1", "0"):
        os.environ["HI3D_FUSED_STEP"] = fused
        try:
            outs[fused] = sampler(denoiser, fx["x0"].clone().to(dev), cond=c, uc=uc)
        finally:
            os.environ.pop("HI3D_FUSED_STEP", None)
    st = list(unet.runtime(dev).steppers.values())
    assert len(st) == 1 and st[0].graph is not None, "the fused path did not run / was not captured"
    rel, cos = stats(outs["1"], outs["0"])
    print(f"fused vs generic: rel {rel:.2e}")
    assert rel < 2e-3
    rel, cos = stats(outs["1"], fx["output"])
    assert rel < 6e-2 and cos > 0.999


@pytest.mark.parametrize("mode", ["fp8qk", "fp8"])
def test_unet_fp8_attention_paths(dev, monkeypatch, mode):
    """BASELINE config 5 (fp8 MFMA attention + bf16 conv): the full-width UNet with every spatial attention's score product
    (fp8qk) or both products (fp8) on the e4m3 / e8m0-scaled matrix path.  Separately stated tolerances for these
    reduced-precision options: fp8qk cosine >= 0.998, max-abs error <= 8e-2 x max-abs reference; fp8 0.995 / 1.2e-1 (bf16
    default: 0.9995 / 4e-2).  The option really switches kernels: the output differs from the bf16 run's."""
    fx = load("unet_s1_lat16")
    i = {k: v.to(dev) for k, v in fx["inputs"].items()}
    run = lambda m: m(i["x"], i["timesteps"], context=i["context"], y=i["y"], num_video_frames=fx["T"],
                      image_only_indicator=i["image_only_indicator"])
    monkeypatch.setenv("HI3D_ATTN_FP8QK", "0"); monkeypatch.setenv("HI3D_ATTN_FP8", "0")
    base = run(build_unet(fx, dev))
    monkeypatch.setenv("HI3D_ATTN_FP8QK", "1" if mode == "fp8qk" else "0")
    monkeypatch.setenv("HI3D_ATTN_FP8", "1" if mode == "fp8" else "0")
    m = build_unet(fx, dev)
    rt = m.runtime(dev)
    assert rt.attn_fp8qk == (mode == "fp8qk") and rt.attn_fp8 == (mode == "fp8")
    out = run(m)
    rel, cos = stats(out, fx["output"])
    d = (out.float() - base.float()).abs().max().item()
    print(f"{mode} UNet: rel {rel:.4f} cos {cos:.6f}; max |out - bf16 run| {d:.3e}")
    assert d > 0
    tol, cmin = {"fp8qk": (8e-2, 0.998), "fp8": (1.2e-1, 0.995)}[mode]
    assert rel < tol and cos > cmin


@pytest.mark.parametrize("name", ["unet_tiny_s2_ioi", "unet_s1_lat16"])
def test_unet_skip_concat_in_place_equals_materialised(dev, monkeypatch, name):
    """Round 4: the decoder's `h = th.cat([h, hs.pop()], dim=1)` (video_model.py:490-499) is read in place by its two
    consumers (two-source GroupNorm, two K segments of the 1x1 skip_connection) instead of being written.  Against the same
    model with the copy (HI3D_CAT_FUSED=0): bit-identical, and the attention's row-major-V form against the transpose pass
    (HI3D_ATTN_VROW=0) likewise -- neither change touches a single arithmetic operation."""
    import importlib
    from hi3d_hip import ops
    fx = load(name)
    i = {k: v.to(dev) for k, v in fx["inputs"].items()}
    run = lambda m: m(i["x"], i["timesteps"], context=i["context"], y=i["y"], num_video_frames=fx["T"],
                      image_only_indicator=i["image_only_indicator"])
    new = run(build_unet(fx, dev))
    monkeypatch.setenv("HI3D_CAT_FUSED", "0")
    m = build_unet(fx, dev)
    assert m.runtime(dev).cat_fused is False
    monkeypatch.setattr(ops, "ATTN_VROW", False)
    old = run(m)
    assert torch.equal(new, old)
    # the GroupNorm statistics taken from the producing conv's accumulators (HI3D_GN_FUSED) are fp32 sums of the UNROUNDED
    # results: same network to rounding noise, not bit for bit
    monkeypatch.setattr(ops, "GN_FUSED", False)
    sep = run(build_unet(fx, dev))
    rel, c = stats(new, sep)
    print(f"{name}: producer-side GroupNorm statistics vs separate passes: rel {rel:.2e} cos {c:.6f}")
    assert rel < 1e-2
    # HI3D_GN_FOLD=1 (opt-in): the transformer's GroupNorm as a per-frame rescaling of proj_in's weights -- the weights, not the
    # activations, take the bf16 rounding of the scale: same network to rounding noise (tokens per frame % 256 == 0 only)
    monkeypatch.setattr(ops, "GN_FUSED", True)
    monkeypatch.setattr(ops, "GN_FOLD", True)
    fold = run(build_unet(fx, dev))
    rel, c = stats(new, fold)
    print(f"{name}: GroupNorm folded into proj_in vs the norm pass: rel {rel:.2e} cos {c:.6f}")
    assert rel < 3e-2 and c > 0.9995            # (the size of the two-stream / tile-variant differences on these fixtures)


@pytest.mark.parametrize("how", ["new_tensor", "in_place"])
def test_guidance_scale_change_after_graph_capture_takes_effect(dev, how):
    """ADVICE r2 / VERDICT r3 item 7: the captured HIP graph of the sampler step reads the guidance scale from a buffer that
    is refreshed when the guider's tensor is REPLACED (`guider.scale = ...`, a CFG sweep, a second sampler) or WRITTEN in
    place -- after the capture.  Reference behaviour: LinearPredictionGuider reads self.scale on every call
    (guiders.py:60-86).  The fused / graph path after the change against the generic per-op path with the same scale."""
    fx = load("sampler_tiny_s2")
    unet, sampler, denoiser = _sampler_stack(fx, dev)
    c = {k: v.to(dev) for k, v in fx["c"].items()}
    uc = {k: v.to(dev) for k, v in fx["uc"].items()}
    x, s_in, sigmas, num_sigmas, cond, ucond = sampler.prepare_sampling_loop(fx["x0"].clone().to(dev), c, uc)
    n = num_sigmas - 1
    for i in range(4):                                   # the third fused step captures the graph; the fourth replays it
        x = sampler.step_call(denoiser, x, i % n, s_in, sigmas, num_sigmas, cond, ucond)
    st = list(unet.runtime(dev).steppers.values())
    assert len(st) == 1 and st[0].graph is not None
    g = sampler.guider
    before = sampler.step_call(denoiser, x, 1, s_in, sigmas, num_sigmas, cond, ucond)     # replay, old scale
    if how == "new_tensor":
        g.scale = torch.linspace(3.0, 7.0, g.num_frames).unsqueeze(0)
    else:
        g.scale.mul_(2.5).add_(0.75)
    after = sampler.step_call(denoiser, x, 1, s_in, sigmas, num_sigmas, cond, ucond)      # replay, new scale
    os.environ["HI3D_FUSED_STEP"] = "0"
    try:
        generic = sampler.step_call(denoiser, x, 1, s_in, sigmas, num_sigmas, cond, ucond)
    finally:
        os.environ.pop("HI3D_FUSED_STEP", None)
    rel, _ = stats(after, generic)
    moved, _ = stats(after, before)
    print(f"scale change ({how}): graph replay vs generic path rel {rel:.2e}; moved the step by {moved:.2e}")
    assert rel < 2e-3 and moved > 1e-2


@pytest.mark.parametrize("name", ["unet_tiny_s2_ioi", "unet_s1_lat16"])
def test_unet_two_stream_cfg_halves_match_single_stream(dev, monkeypatch, name):
    """HI3D_TWO_STREAM=1: the unconditional and the conditional half of the batch run through the large resolution levels as
    two kernel sequences on two HIP streams (they never mix inside the network; runtime_unet.forward_tokens), joint below.
    Same arithmetic per frame; a half-batch launch may take another tile variant / GroupNorm blocking, so the comparison with
    the single-stream forward is to rounding noise (<= 3e-2), and with the reference golden to the UNet tolerance."""
    fx = load(name)
    i = {k: v.to(dev) for k, v in fx["inputs"].items()}
    run = lambda m: m(i["x"], i["timesteps"], context=i["context"], y=i["y"], num_video_frames=fx["T"],
                      image_only_indicator=i["image_only_indicator"])
    one = run(build_unet(fx, dev))
    monkeypatch.setenv("HI3D_TWO_STREAM", "1")
    m = build_unet(fx, dev)
    rt = m.runtime(dev)
    assert rt.two_stream == "1" and rt._split_plan()[0] > 0
    two = run(m)
    two2 = run(m)
    torch.cuda.synchronize()
    assert torch.equal(two, two2)
    rel, c = stats(two, one)
    relg, cg = stats(two, fx["output"])
    print(f"{name}: two streams vs one: rel {rel:.2e}; vs reference golden rel {relg:.4f} cos {cg:.6f}")
    # (a half-batch launch takes other tile variants / split-K / GroupNorm blockings at these small sizes: the two forwards
    # differ by accumulated bf16 rounding noise -- measured 1.7e-2 at full width, latent 16 -- the golden bound is what matters)
    assert rel < 3e-2 and relg < 4e-2 and cg > 0.9995


def test_two_stream_step_graph_matches_single_stream(dev, monkeypatch):
    """The captured sampler step with the two-stream forward inside (fork / join of the side stream inside the HIP graph):
    several replays, against the single-stream sampler on the same inputs and against the reference golden."""
    fx = load("sampler_tiny_s2")
    c = {k: v.to(dev) for k, v in fx["c"].items()}
    uc = {k: v.to(dev) for k, v in fx["uc"].items()}
    outs = {}
    for two in ("0", "1"):
        monkeypatch.setenv("HI3D_TWO_STREAM", two)
        unet, sampler, denoiser = _sampler_stack(fx, dev)
        outs[two] = sampler(denoiser, fx["x0"].clone().to(dev), cond=c, uc=uc)
        st = list(unet.runtime(dev).steppers.values())
        assert len(st) == 1 and (st[0].graph is not None) == (fx["steps"] >= 3)
    rel, _ = stats(outs["1"], outs["0"])
    relg, cg = stats(outs["1"], fx["output"])
    print(f"two-stream sampler: vs single stream rel {rel:.2e}; vs golden rel {relg:.4f} cos {cg:.6f}")
    assert rel < 2e-2 and relg < 6e-2 and cg > 0.999
