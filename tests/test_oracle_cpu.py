"""Pin the CPU oracle (oracle/hi3d_oracle.py) against golden vectors produced by the
reference's own classes (oracle/gen_golden.py).  fp32 vs fp32 on the same torch build:
the only differences are summation order inside identical ATen kernels."""
import glob
import os

import pytest
import torch

from hi3d_hip import synth
from oracle import hi3d_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-4      # max-abs error / max-abs reference, fp32 restatement vs fp32 reference


def load(name):
    return torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)


def weights(fx):
    return synth.synth_state_dict({fx["key_prefix"] + k: s for k, s in fx["shapes"].items()}, fx["weight_seed"])


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


def test_schedule_and_scaling_known_answers():
    fx = load("schedule")
    for n, ref in fx["sigmas"].items():
        assert torch.allclose(O.edm_sigmas(n), ref, rtol=1e-6, atol=0)
    s5 = O.edm_sigmas(5)
    # closed-form KATs: endpoints of the rho-schedule and the appended zero
    assert abs(s5[0].item() - 700.0) < 1e-3 and abs(s5[4].item() - 0.002) < 1e-8 and s5[5].item() == 0.0
    s = fx["scaling_in"]
    c_skip, c_out, c_in, c_noise = fx["scaling_out"]
    assert torch.allclose(c_skip + c_out ** 2, torch.ones_like(s), atol=1e-6)      # identities of VScaling
    assert torch.allclose(c_in ** 2, c_skip, rtol=1e-6)
    assert torch.allclose(c_noise, 0.25 * s.log())


def test_synth_weights_are_reproducible():
    fx = load("unet_tiny_s1")
    sd = weights(fx)
    for k, v in fx["probe"].items():
        assert torch.equal(sd[fx["key_prefix"] + k].flatten()[:4], v), "torch CPU RNG differs from the fixture build"


@pytest.mark.parametrize("name", ["unet_tiny_s1", "unet_tiny_s2_ioi"])
def test_unet_oracle_matches_reference(name):
    fx = load(name)
    i = fx["inputs"]
    with torch.no_grad():
        out = O.video_unet(weights(fx), fx["cfg"], i["x"], i["timesteps"], i["context"], i["y"], fx["T"],
                           i["image_only_indicator"], prefix=fx["key_prefix"])
    assert rel(out, fx["output"]) < TOL


@pytest.mark.parametrize("name", ["sampler_tiny_s1", "sampler_tiny_s2"])
def test_sampler_oracle_matches_reference(name):
    fx = load(name)
    with torch.no_grad():
        out, traj = O.euler_edm_sample(weights(fx), fx["cfg"], fx["x0"], fx["c"], fx["uc"], fx["T"], fx["steps"],
                                       fx["max_scale"], prefix=fx["key_prefix"], return_all=True)
    for k in range(fx["steps"]):
        assert rel(traj[k], fx["traj"][k]) < TOL, f"step {k}"
    assert rel(out, fx["output"]) < TOL


@pytest.mark.parametrize("name", ["vae_tiny", "vae_full_lat8"])
def test_vae_decode_oracle_matches_reference(name):
    fx = load(name)
    with torch.no_grad():
        out = O.vae_decode(weights(fx), fx["ddconfig"], fx["z"], prefix=fx["key_prefix"])
    assert rel(out, fx["output"]) < TOL


def test_cross_attention_single_token_identity():
    """One context token => softmax == 1 => attn2(x) = to_out(to_v(ctx)) for every query
    (SURVEY 0.7).  The HIP path relies on this exact elimination."""
    g = torch.Generator().manual_seed(0)
    C, ctxd = 128, 1024
    sd = {"a.to_q.weight": torch.randn(C, C, generator=g), "a.to_k.weight": torch.randn(C, ctxd, generator=g),
          "a.to_v.weight": torch.randn(C, ctxd, generator=g) * 0.03, "a.to_out.0.weight": torch.randn(C, C, generator=g) * 0.1,
          "a.to_out.0.bias": torch.randn(C, generator=g)}
    x, ctx = torch.randn(3, 50, C, generator=g), torch.randn(3, 1, ctxd, generator=g)
    full = O._attn(sd, "a", x, ctx, heads=2)
    vec = torch.nn.functional.linear(torch.nn.functional.linear(ctx, sd["a.to_v.weight"]), sd["a.to_out.0.weight"],
                                     sd["a.to_out.0.bias"])
    assert torch.allclose(full, vec.expand_as(full), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", ["vae_enc_tiny", "vae_enc_full_64"])
def test_vae_encode_oracle_matches_reference(name):
    fx = load(name)
    sd = weights(fx)
    with torch.no_grad():
        mom = O.vae_encode_moments(sd, fx["ddconfig"], fx["x"], prefix=fx["key_prefix"])
        z = O.vae_encode(sd, fx["ddconfig"], fx["x"], noise=fx["sample_noise"], scale_factor=1.0, prefix=fx["key_prefix"])
    assert rel(mom, fx["moments"]) < TOL
    assert rel(z, fx["z_sampled"]) < TOL          # reference posterior.sample() with the same CPU draw


def test_unet_oracle_matches_reference_at_32_views_full_width():
    """The oracle at BASELINE config 4's frame count: full-width stage-2 UNet, T = 32, latent 16 x 16, against one forward of
    the reference VideoUNet (golden stored in fp16: 5e-4 of quantisation, hence the looser bound)."""
    fx = load("unet_s2_lat16_t32")
    inp = synth.synth_unet_inputs(fx["cfg"], fx["T"], fx["hw"], fx["input_seed"])
    assert torch.equal(inp["x"].flatten()[:16], fx["input_probe"]["head"])
    with torch.no_grad():
        out = O.video_unet(weights(fx), fx["cfg"], inp["x"], inp["timesteps"], inp["context"], inp["y"], fx["T"],
                           inp["image_only_indicator"], prefix=fx["key_prefix"])
    assert fx["T"] == 32 and rel(out, fx["output"].float()) < 1e-3


def test_v02_refine_oracle_matches_reference():
    """(the full-width 25-step golden v02_w320_25step is consumed by the GPU suite only: the oracle needs minutes for it)"""
    fx = load("v02_tiny")
    with torch.no_grad():
        out = O.v02_refine(weights(fx), fx["cfg"], fx["z_frames"], fx["init"], fx["c"], fx["uc"], fx["T"], fx["steps"],
                           fx["max_scale"], prefix=fx["key_prefix"])
    assert rel(out, fx["output"]) < TOL


@pytest.mark.parametrize("name", ["videodec_tiny", "videodec_full_lat8", "videodec_full_lat32", "videodec_tiny_k3", "videodec_full_lat16_k3"])
def test_video_decoder_oracle_matches_reference(name):
    fx = load(name)
    with torch.no_grad():
        out = O.video_decode(weights(fx), fx["ddconfig"], fx["z"], fx["T"], prefix=fx["key_prefix"])
    assert rel(out, fx["output"]) < TOL


@pytest.mark.parametrize("name", ["clip_vith_like", "clip_vitl_like"])
def test_clip_visual_oracle_matches_independent_implementation(name):
    """The CLIP vision tower restatement (original open_clip / OpenAI key names) vs HuggingFace's
    CLIPVisionModelWithProjection on the same weights (oracle/gen_golden_clip.py): head_dim 80 + GELU as ViT-H/14,
    head_dim 64 + QuickGELU as ViT-L/14.  open_clip / clip themselves are not installable here: unpinned against them."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle"))
    from gen_golden_clip import clip_shapes
    fx = load(name)
    c = fx["cfg"]
    sd = synth.synth_state_dict(clip_shapes(c["width"], c["layers"], c["patch"], c["grid"], c["out_dim"]), fx["seed"])
    out = O.clip_visual(sd, fx["img"], c["heads"], c["act"])
    assert out.shape == fx["out"].shape
    assert rel(out, fx["out"]) < TOL


def test_dpt_hybrid_oracle_matches_reference_midas():
    """The depth conditioner's DPT-hybrid restatement vs the reference's vendored MiDaS code run on the timm stand-in
    backbone (oracle/gen_golden_dpt.py); the fixture also records how far HuggingFace's independent DPT-hybrid was from
    that run on the same weights (pins the stand-in backbone: must be fp32 noise)."""
    fx = load("dpt_hybrid_64x96")
    P = fx["key_prefix"]
    sd = synth.damp_residual_tails(synth.synth_state_dict({P + k: s for k, s in fx["shapes"].items()}, fx["weight_seed"]), fx["damp"])
    out, layers = O.dpt_hybrid(sd, fx["x"], P, return_layers=True)
    for a, b in zip(layers, fx["layers"]):
        assert a.shape == b.shape and rel(a, b) < TOL
    assert out.shape == fx["output"].shape and rel(out, fx["output"]) < TOL
    assert fx["hf_maxdiff"] < 1e-3
    # DepthEmbedder.forward vs the reference's own class (vtdm/encoders.py:31-53 run by oracle/gen_golden_dpt.py with
    # MiDaSInference around the same network): resizes, per-frame min-max, 3x3 pixel-unshuffle
    xe = torch.rand(fx["embedder_input_shape"], generator=torch.Generator().manual_seed(fx["embedder_input_seed"])) * 2 - 1
    de = O.depth_embedder(sd, xe, prefix=P)
    assert de.shape == fx["embedder_output"].shape and (de - fx["embedder_output"]).abs().max() < 1e-5
    # ... and its shape, range and unshuffle order spelled out
    g = torch.Generator().manual_seed(5)
    x = torch.rand((2, 3, 192, 256), generator=g) * 2 - 1
    d = O.depth_embedder(sd, x, prefix=P)
    assert d.shape == (2, 9, 24, 32) and float(d.min()) == 0.0 and float(d.max()) == 1.0
    y = O.dpt_hybrid(sd, torch.nn.functional.interpolate(x, [64, 96], mode="bilinear"), P)[:, None]
    y = torch.nn.functional.interpolate(y, [72, 96], mode="bilinear")
    y = (y - y.amin(dim=(1, 2, 3), keepdim=True))
    y = y / y.amax(dim=(1, 2, 3), keepdim=True)
    assert torch.allclose(d[:, 4], y[:, 0, 1::3, 1::3], atol=1e-6)       # channel h0*3 + w0 = 4 <- offsets (1, 1)
