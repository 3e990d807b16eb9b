"""Parity AT THE BENCHMARKED SIZES (BASELINE config[2]: 16 views @ 1024x1024, CFG batch 32,
latent 128x128): the kernels use 32-bit buffer offsets, per-block descriptors and grid-size dependent
XCD tile maps, so the shapes bench.py times are checked here against fp32 CPU references / goldens of
the reference classes -- not only the small shapes of test_kernels_gpu.py.

Same tolerances as the small-shape tests (bf16 storage, fp32 accumulation):
  kernels  : max-abs error <= 1.2e-2 x max-abs reference (2e-2 for attention, P is bf16)
  UNet     : <= 2.5e-2, cosine >= 0.9998 at the full sizes (round 6: tightened from 4e-2 / 0.9995 -- measured 1.5 - 2.0e-2 /
             0.99989 on every full-size golden, so a kernel regression of 2x in error used to pass)
  25-step latents: cosine >= 0.999, <= 6e-2
  VAE image: <= 4e-2, PSNR >= 35 dB
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
BF16_TOL = 1.2e-2


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def bf(x):
    return x.to(torch.bfloat16)


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def cos(a, b):
    return F.cosine_similarity(a.float().cpu().flatten(), b.float().cpu().flatten(), dim=0).item()


def load(name):
    path = os.path.join(GOLD, name + ".pt")
    if not os.path.exists(path):
        pytest.skip(f"{name}.pt not generated (oracle/gen_golden.py --full)")
    return torch.load(path, weights_only=False)


# ------------------------------------------------------------------ kernels at size
@pytest.mark.parametrize("prescaled", [False, True])
def test_attention_d64_S16384(dev, prescaled):
    """The 128x128-latent spatial self-attention: 16384 tokens, 5 heads (attention.py:332-336)."""
    from hi3d_hip import ops
    B, H, S = 2, 5, 16384
    C = H * 64
    qkv = rnd((B * S, 3 * C), 1)
    if prescaled:
        qkv[:, :C] *= ops.Q_PRESCALE
    qkv = bf(qkv)
    q, k, v = [t.float().reshape(B, S, H, 64).transpose(1, 2) for t in qkv.split(C, dim=1)]
    if prescaled:      # base-2 softmax of q'.k  ==  softmax(q' k^T ln 2)
        ref = F.scaled_dot_product_attention(q, k, v, scale=math.log(2.0))
    else:
        ref = F.scaled_dot_product_attention(q, k, v)
    ref = ref.transpose(1, 2).reshape(B * S, C)
    out = ops.self_attention_fused_qkv(qkv.to(dev), B, S, H, q_prescaled=prescaled)
    # averaging 16384 values shrinks the output (|out| ~ 1e-2): compare against the output scale
    assert relerr(out, ref) < 2e-2 and cos(out, ref) > 0.9995


def test_conv3x3_320_at_128x128(dev):
    """ResBlock conv at the top level with its fused tail: [4,320,128,128] -> 320, + bias + emb + skip."""
    from hi3d_hip import ops
    from hi3d_hip.pack import pack_conv3x3
    Fr, H, W_, Cin, Cout = 4, 128, 128, 320, 320
    x = bf(rnd((Fr, Cin, H, W_), 1))
    w = bf(rnd((Cout, Cin, 3, 3), 2, (9 * Cin) ** -0.5)).float()
    b, emb = rnd((Cout,), 3), rnd((Fr, Cout), 4)
    R1 = bf(rnd((Fr * H * W_, Cout), 5))
    ref = F.conv2d(x.float(), w, b, padding=1) + emb[:, :, None, None]
    ref = ref.permute(0, 2, 3, 1).reshape(-1, Cout) + R1.float()
    xt = x.permute(0, 2, 3, 1).contiguous().reshape(-1, Cin)
    out = ops.gemm(xt.to(dev), pack_conv3x3(w, Cin).to(dev), M=Fr * H * W_, N=Cout, K=9 * Cin, bias=b.to(dev),
                   rowvec=emb.to(dev), rows_per_group=H * W_, R1=R1.to(dev),
                   conv3x3=dict(Hin=H, Win=W_, Cin=Cin, Hout=H, Wout=W_, stride=1, up2x=0))
    assert relerr(out, ref) < BF16_TOL


def test_gemm_dense_M524288(dev):
    """Row count of the stage-2 top level (32 x 16384 tokens): QKV projection, out = 1 GB."""
    from hi3d_hip import ops
    M, N, K = 32 * 16384, 960, 320
    A, W = bf(rnd((M, K), 1)), bf(rnd((N, K), 2, K ** -0.5))
    out = ops.gemm(A.to(dev), W.to(dev), M=M, N=N, K=K)
    ref = A.float() @ W.float().T
    assert relerr(out, ref) < BF16_TOL
    # the last rows / tiles specifically (XCD remap + M tail live there)
    assert relerr(out[-300:], ref[-300:]) < BF16_TOL


def test_ffn_fused_M524288(dev):
    from hi3d_hip import ops
    from hi3d_hip.pack import pack_geglu, pack_linear
    M, C = 32 * 16384, 320
    x = bf(rnd((M, C), 31))
    w1, b1 = bf(rnd((8 * C, C), 32, C ** -0.5)).float(), rnd((8 * C,), 33)
    w2, b2 = bf(rnd((C, 4 * C), 34, (4 * C) ** -0.5)).float(), rnd((C,), 35)
    R1 = bf(rnd((M, C), 36))
    w1p, b1p = pack_geglu(w1, b1)
    out = ops.ffn_geglu(x.to(dev), w1p.to(dev), b1p.to(dev), pack_linear(w2).to(dev), b2.to(dev), M=M, C=C, R1=R1.to(dev))
    sl = torch.cat([torch.arange(0, 4096), torch.arange(M // 2 - 1000, M // 2 + 1000), torch.arange(M - 4096, M)])
    h = x[sl].float() @ w1.T + b1
    hg = bf(h[:, :4 * C] * F.gelu(h[:, 4 * C:])).float()
    ref = hg @ w2.T + b2 + R1[sl].float()
    assert relerr(out[sl.to(dev)], ref) < BF16_TOL


def test_groupnorm_1Mpixel_C128(dev):
    """VAE decoder norm at the 1024x1024 level: one instance of 1 M pixels x 128 channels (model.py:21-24)."""
    from hi3d_hip import ops
    inst, P, C = 1, 1024 * 1024, 128
    x = bf(rnd((inst, P, C), 1) * 2.0 + 0.7)
    g, b = 1 + 0.1 * rnd((C,), 2), 0.1 * rnd((C,), 3)
    ref = F.silu(F.group_norm(x.float().permute(0, 2, 1), 32, g, b, 1e-6)).permute(0, 2, 1)
    out = ops.groupnorm_silu(x.reshape(-1, C).to(dev), g.to(dev), b.to(dev), inst, P, C, 1e-6, True)
    assert relerr(out.reshape(inst, P, C), ref) < BF16_TOL


def test_groupnorm_3d_16x128x128_C320(dev):
    """time_stack norm at the top level: statistics over (t,h,w) = 262144 positions per clip."""
    from hi3d_hip import ops
    inst, P, C = 2, 16 * 128 * 128, 320
    x = bf(rnd((inst, P, C), 1) * 1.5 - 0.4)
    g, b = 1 + 0.1 * rnd((C,), 2), 0.1 * rnd((C,), 3)
    ref = F.silu(F.group_norm(x.float().permute(0, 2, 1), 32, g, b, 1e-5)).permute(0, 2, 1)
    out = ops.groupnorm_silu(x.reshape(-1, C).to(dev), g.to(dev), b.to(dev), inst, P, C, 1e-5, True)
    assert relerr(out.reshape(inst, P, C), ref) < BF16_TOL


def test_layernorm_M524288(dev):
    from hi3d_hip import ops
    R, C = 32 * 16384, 320
    x = bf(rnd((R, C), 1) * 1.5 + 0.3)
    g, b = 1 + 0.1 * rnd((C,), 2), 0.1 * rnd((C,), 3)
    out = ops.layernorm(x.to(dev), g.to(dev), b.to(dev), R, C, 1e-5)
    assert relerr(out, F.layer_norm(x.float(), (C,), g, b, 1e-5)) < BF16_TOL


def test_attention_temporal_T32_H20(dev):
    """BASELINE config 4 family: 32 views, 1280 channels (20 heads), 32x32 latent level."""
    from hi3d_hip import ops
    B, T, S, H = 2, 32, 1024, 20
    C = H * 64
    qkv = bf(rnd((B * T * S, 3 * C), 1))
    q, k, v = [t.float().reshape(B, T, S, H, 64).permute(0, 2, 3, 1, 4).reshape(B * S, H, T, 64)
               for t in qkv.split(C, dim=1)]
    ref = F.scaled_dot_product_attention(q, k, v)
    ref = ref.reshape(B, S, H, T, 64).permute(0, 3, 1, 2, 4).reshape(B * T * S, C)
    out = ops.attention_temporal_fused_qkv(qkv.to(dev), B, T, S, H)
    assert relerr(out, ref) < 2e-2


def test_conv_temporal_T32_C1280(dev):
    from hi3d_hip import ops
    from hi3d_hip.pack import pack_convt3
    B, T, HW, C = 2, 32, 256, 1280
    x = bf(rnd((B, C, T, HW, 1), 1))
    w = bf(rnd((C, C, 3, 1, 1), 2, (3 * C) ** -0.5)).float()
    b = rnd((C,), 3)
    ref = F.conv3d(x.float(), w, b, padding=(1, 0, 0))
    xt = x.squeeze(-1).permute(0, 2, 3, 1).contiguous().reshape(-1, C)
    out = ops.gemm(xt.to(dev), pack_convt3(w).to(dev), M=B * T * HW, N=C, K=3 * C, bias=b.to(dev),
                   convt3=dict(T=T, HW=HW, Cin=C))
    got = out.float().cpu().reshape(B, T, HW, C).permute(0, 3, 1, 2).unsqueeze(-1)
    assert relerr(got, ref) < BF16_TOL


def test_softmax_rows_at_limit(dev):
    """VAE mid-block attention of a 1024x1024 frame: rows of 16384 scores (model.py:180-195)."""
    from hi3d_hip import ops
    s = rnd((64, 16384), 4) * 3
    p = ops.softmax_rows(s.to(dev), 64, 16384, 16384, 512 ** -0.5)
    assert relerr(p, torch.softmax(s * 512 ** -0.5, -1)) < 5e-3


# ------------------------------------------------------------------ networks at size
def _build_unet(fx, dev):
    from conftest import synth_unet          # seeded weights drawn once per session, a fresh module per call
    return synth_unet(fx, dev)


@pytest.mark.parametrize("attn", ["bf16", "fp8qk", "fp8"])
def test_unet_full_size_stage2_matches_reference_golden(dev, attn, monkeypatch):
    """THE benchmarked forward: B = 2x16 frames, 17 input channels, latent 128x128 (S = 16384 tokens,
    M = 524288-row GEMMs), against one forward of the reference VideoUNet (fp32 CPU).  bf16: the full-size tolerance
    (2.5e-2, cos >= 0.9998).  BASELINE config 5 at size, each reduced-precision attention under its own stated tolerance:
    fp8qk (score product on e4m3) 8e-2 / cos >= 0.998; fp8 (both products) 1.2e-1 / cos >= 0.995."""
    from hi3d_hip import synth
    monkeypatch.setenv("HI3D_ATTN_FP8QK", "1" if attn == "fp8qk" else "0")
    monkeypatch.setenv("HI3D_ATTN_FP8", "1" if attn == "fp8" else "0")
    fx = load("unet_s2_full")
    inp = synth.synth_unet_inputs(fx["cfg"], fx["T"], fx["hw"], fx["input_seed"])
    x = inp["x"]
    pr = fx["input_probe"]
    assert torch.equal(x.flatten()[:16], pr["head"]) and abs(float(x.double().sum()) - pr["sum"]) < 1e-6 * pr["abs_sum"], \
        "seeded inputs are not the ones the golden was generated from"
    m = _build_unet(fx, dev)
    rt = m.runtime(dev)
    assert rt.attn_fp8qk == (attn == "fp8qk") and rt.attn_fp8 == (attn == "fp8")
    out = m(x.to(dev), inp["timesteps"].to(dev), context=inp["context"].to(dev), y=inp["y"].to(dev),
            num_video_frames=fx["T"], image_only_indicator=inp["image_only_indicator"].to(dev))
    ref = fx["output"].float()
    rel, c = relerr(out, ref), cos(out, ref)
    print(f"unet_s2_full [{attn}]: rel {rel:.4f} cos {c:.6f}")
    assert tuple(out.shape) == tuple(ref.shape) == (32, 4, 128, 128)
    tol, cmin = {"bf16": (2.5e-2, 0.9998), "fp8qk": (8e-2, 0.998), "fp8": (1.2e-1, 0.995)}[attn]
    assert rel < tol and c > cmin


def test_unet_config4_full_size_matches_reference_golden(dev):
    """BASELINE config 4 at its REAL size against the REFERENCE (until round 5 this size had kernels vs fp32
    and the network vs itself only): the full-width stage-2 VideoUNet on the CFG pair of a 32-view clip -- 64 frames, latent
    128 x 128, 1,048,576 token rows, 16384-token spatial attention at B = 64, temporal attention / Conv3d / 3-D GroupNorm over 32
    frames -- in ONE forward on the GPU (the product's batch of 64), against the reference's own VideoUNet run on the CPU one
    clip per call (the two clips never mix inside the network; 90 GB in one call, 45 GB this way; oracle/gen_golden.py
    `unet_s2_full_t32`, ~40 min).  Same tolerance as the 16-view forward: 2.5e-2 of the output range, cosine >= 0.9998."""
    from hi3d_hip import synth
    if not os.path.exists(os.path.join(GOLD, "unet_s2_full_t32.pt")):
        pytest.skip("unet_s2_full_t32.pt not generated")
    fx = load("unet_s2_full_t32")
    T, hw = fx["T"], fx["hw"]
    assert (T, hw) == (32, 128) and fx.get("two_calls")
    inp = synth.synth_unet_inputs(fx["cfg"], T, hw, fx["input_seed"])
    x, pr = inp["x"], fx["input_probe"]
    assert torch.equal(x.flatten()[:16], pr["head"]) and abs(float(x.double().sum()) - pr["sum"]) < 1e-6 * pr["abs_sum"], \
        "seeded inputs are not the ones the golden was generated from"
    m = _build_unet(fx, dev)
    out = m(x.to(dev), inp["timesteps"].to(dev), context=inp["context"].to(dev), y=inp["y"].to(dev),
            num_video_frames=T, image_only_indicator=inp["image_only_indicator"].to(dev))
    ref = fx["output"].float()
    assert tuple(out.shape) == tuple(ref.shape) == (64, 4, 128, 128)
    rel, c = relerr(out, ref), cos(out, ref)
    relh = [relerr(out[b * T:(b + 1) * T], ref[b * T:(b + 1) * T]) for b in (0, 1)]
    print(f"unet_s2_full_t32 (config 4, 64 frames @ 128^2): rel {rel:.4f} cos {c:.6f}; per clip {relh[0]:.4f} / {relh[1]:.4f}")
    assert rel < 2.5e-2 and c > 0.9998


@pytest.mark.parametrize("name", ["unet_s2_lat16_t32", "unet_s2_lat64_t32"])
def test_unet_full_width_32_views_matches_reference_golden(dev, name):
    """BASELINE config 4 (32 views): the full-width stage-2 UNet (320 .. 1280 channels) at T = 32 -- CFG batch 64 -- on
    latent 16 x 16 and 64 x 64 (4096-token spatial attention, 262144-row GEMMs) against one forward of the REFERENCE
    VideoUNet at T = 32 (round 2 compared with the oracle, itself pinned at T = 4 .. 16 only): temporal attention over 32
    frames, Conv3d and 3-D GroupNorm over 32 frames at every width.  (The 128 x 128 latent of config 4 needs ~90 GB for the
    reference on the CPU: not generated.)"""
    from hi3d_hip import synth
    if not os.path.exists(os.path.join(GOLD, name + ".pt")):
        pytest.skip(f"{name}.pt not generated")
    fx = load(name)
    T, hw, cfg = fx["T"], fx["hw"], fx["cfg"]
    inp = synth.synth_unet_inputs(cfg, T, hw, fx["input_seed"])
    pr = fx["input_probe"]
    assert torch.equal(inp["x"].flatten()[:16], pr["head"]) and abs(float(inp["x"].double().sum()) - pr["sum"]) < 1e-6 * pr["abs_sum"]
    m = _build_unet(fx, dev)
    out = m(inp["x"].to(dev), inp["timesteps"].to(dev), context=inp["context"].to(dev), y=inp["y"].to(dev),
            num_video_frames=T, image_only_indicator=inp["image_only_indicator"].to(dev))
    ref = fx["output"].float()
    rel, c = relerr(out, ref), cos(out, ref)
    print(f"unet full width, 32 views, latent {hw} vs reference: rel {rel:.4f} cos {c:.6f}")
    assert T == 32 and tuple(out.shape) == (2 * T, 4, hw, hw) == tuple(ref.shape) and rel < 2.5e-2 and c > 0.9998


def test_sampler_25_steps_full_width_matches_reference_golden(dev):
    """bf16 error accumulation over the whole schedule: 25 Euler-EDM steps (sigma_max 700, CFG 1 -> 2.5)
    through the full-width (320..1280 channels, 1.52 B parameters) stage-1 UNet vs the reference's
    trajectory.  SURVEY 8d: 25-step latents cosine >= 0.999."""
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    fx = load("sampler_s1_w320_25step")
    T = fx["T"]
    model = OpenAIWrapper(_build_unet(fx, dev))
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
    worst = (0.0, 1.0)
    for i in sampler.get_sigma_gen(num_sigmas):
        x = sampler.step_call(denoiser, x, i, s_in, sigmas, num_sigmas, cond, ucond)
        rel, cs = relerr(x, fx["traj"][i]), cos(x, fx["traj"][i])
        worst = (max(worst[0], rel), min(worst[1], cs))
        assert rel < 6e-2 and cs > 0.999, f"step {i}: rel {rel:.4f} cos {cs:.6f}"
    print(f"25-step full-width trajectory: worst rel {worst[0]:.4f}, worst cos {worst[1]:.6f}; "
          f"final rel {relerr(x, fx['output']):.4f} cos {cos(x, fx['output']):.6f}")
    assert num_sigmas - 1 == 25 and cos(x, fx["output"]) > 0.999


@pytest.mark.parametrize("mode", ["fp8qk", "fp8"])
def test_sampler_25_steps_full_width_fp8_attention(dev, monkeypatch, mode):
    """BASELINE config 5 over the WHOLE schedule: the 25 Euler-EDM + CFG steps of the full-width stage-1
    UNet with every spatial attention on the fp8 matrix path -- score product only (fp8qk) or both products (fp8) -- against
    the reference's fp32 trajectory.  How the e4m3 rounding of q / k (/ P / v) accumulates over 25 steps was unknown; the
    separately stated bounds for these reduced-precision options (bf16 default: per-step cosine >= 0.999, <= 6e-2):
      fp8qk : per-step cosine >= 0.998, max-abs error <= 8e-2 x max-abs reference
      fp8   : per-step cosine >= 0.995, max-abs error <= 1.2e-1 x max-abs reference
    (the single-forward tolerances of these modes, tests/test_unet_gpu.py::test_unet_fp8_attention_paths: the error does not
    compound beyond them -- each Euler step re-anchors the latent on the network's denoised estimate)."""
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    monkeypatch.setenv("HI3D_ATTN_FP8QK", "1" if mode == "fp8qk" else "0")
    monkeypatch.setenv("HI3D_ATTN_FP8", "1" if mode == "fp8" else "0")
    fx = load("sampler_s1_w320_25step")
    T = fx["T"]
    unet = _build_unet(fx, dev)
    rt = unet.runtime(dev)
    assert rt.attn_fp8qk == (mode == "fp8qk") and rt.attn_fp8 == (mode == "fp8")
    model = OpenAIWrapper(unet)
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

    tol, cmin = {"fp8qk": (8e-2, 0.998), "fp8": (1.2e-1, 0.995)}[mode]
    x, s_in, sigmas, num_sigmas, cond, ucond = sampler.prepare_sampling_loop(fx["x0"].clone().to(dev), c, uc)
    worst = (0.0, 1.0)
    for i in sampler.get_sigma_gen(num_sigmas):
        x = sampler.step_call(denoiser, x, i, s_in, sigmas, num_sigmas, cond, ucond)
        rel, cs = relerr(x, fx["traj"][i]), cos(x, fx["traj"][i])
        worst = (max(worst[0], rel), min(worst[1], cs))
        assert rel < tol and cs > cmin, f"{mode} step {i}: rel {rel:.4f} cos {cs:.6f}"
    print(f"25-step full-width trajectory, {mode} attention: worst rel {worst[0]:.4f}, worst cos {worst[1]:.6f}; "
          f"final rel {relerr(x, fx['output']):.4f} cos {cos(x, fx['output']):.6f}")
    assert num_sigmas - 1 == 25


def test_stage2_refine_25_steps_full_width_matches_reference_golden(dev):
    """The north-star loop itself over its whole schedule: pipeline_i2v_eval_v02.py:103-135 -- re-noising blend towards the
    stage-1 latents, Euler-EDM, CFG 1 -> 2.0 -- for 25 steps through the full-width (1.52 B parameters, 17 input channels)
    stage-2 UNet, hi3d_hip.pipelines.stage2_refine (fused blend kernel + fused graph-replayed step) vs the reference's
    own loop run with the reference classes (oracle/gen_golden.py:gen_v02).  Final latents: cosine >= 0.999, max-abs
    error <= 6e-2 x max-abs reference (the sampler tolerance of DESIGN.md section 5)."""
    from types import SimpleNamespace
    from hi3d_hip.pipelines import stage2_refine
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    fx = load("v02_w320_25step")
    T = fx["T"]
    model = SimpleNamespace(
        device=dev, model=OpenAIWrapper(_build_unet(fx, dev)),
        denoiser=Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}),
        sampler=EulerEDMSampler(
            num_steps=fx["steps"], device=dev,
            discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
            guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                           "params": {"num_frames": T, "max_scale": fx["max_scale"], "min_scale": 1.0}}))
    c = {k: v.to(dev) for k, v in fx["c"].items()}
    uc = {k: v.to(dev) for k, v in fx["uc"].items()}
    out = stage2_refine(model, None, c, uc, init_noise=fx["init"], decode=False, z_frames=fx["z_frames"])
    rel, cs = relerr(out, fx["output"]), cos(out, fx["output"])
    print(f"stage-2 refine, 25 steps, full width: rel {rel:.4f} cos {cs:.6f}")
    assert fx["steps"] == 25 and fx["cfg"]["in_channels"] == 17 and fx["cfg"]["model_channels"] == 320
    assert rel < 6e-2 and cs > 0.999


def test_vae_encode_full_resolution_matches_reference_golden(dev):
    """encode_first_stage of one 1024 x 1024 frame through the full-width encoder (what stage 2 does 16 times per clip for
    the refine loop and 16 times for the conditioning latents): bottom/right-padded stride-2 convs from 1 M pixels down,
    16384-token d = 512 mid-block attention, quant_conv, posterior -- mode vs the reference's mean, sample (with the
    reference's own CPU noise draw) vs its posterior.sample()."""
    from hi3d_hip import synth
    from sgm.models.autoencoder import AutoencoderKL
    fx = load("vae_enc_full_1024")
    g = torch.Generator().manual_seed(fx["input_seed"])
    x = torch.rand(fx["x_shape"], generator=g) * 2 - 1
    assert torch.equal(x.flatten()[:16], fx["x_head"]), "input re-draw does not reproduce the generator's"
    ae = AutoencoderKL(embed_dim=4, ddconfig=fx["ddconfig"])
    synth.fill_module_(ae, fx["weight_seed"], prefix=fx["key_prefix"])
    ae = ae.to(dev)
    z = ae.encode(x.to(dev), noise=fx["sample_noise"]).float().cpu()
    zm = ae.encode(x.to(dev), noise=torch.zeros_like(fx["sample_noise"])).float().cpu()      # mean: sample with zero noise
    mean_ref = fx["moments"][:, :4]
    rel_s, rel_m = relerr(z, fx["z_sampled"]), relerr(zm, mean_ref)
    print(f"vae encode 1024^2: sample rel {rel_s:.4f} cos {cos(z, fx['z_sampled']):.6f}  mean rel {rel_m:.4f}")
    assert tuple(z.shape) == (1, 4, 128, 128) and rel_s < 4e-2 and rel_m < 4e-2 and cos(z, fx["z_sampled"]) > 0.9995


@pytest.mark.parametrize("name", ["vae_full_512", "vae_full_1024"])
def test_vae_decode_full_resolution_matches_reference_golden(dev, name):
    """decode_first_stage of one frame at 512x512 / 1024x1024 through the full-width decoder (16384-token
    d=512 mid-block attention, 1 M-pixel GroupNorm, 512 -> 1024 nearest-2x conv) vs the reference."""
    from hi3d_hip import synth
    from sgm.models.autoencoder import AutoencoderKL
    fx = load(name)
    ae = AutoencoderKL(embed_dim=4, ddconfig=fx["ddconfig"], lossconfig={"target": "torch.nn.Identity"})
    synth.fill_module_(ae, fx["weight_seed"], prefix=fx["key_prefix"])
    ae = ae.to(dev)
    g = torch.Generator().manual_seed(fx["input_seed"])
    z = torch.randn(fx["z_shape"], generator=g)
    assert torch.equal(z.flatten()[:16], fx["z_head"])
    out = ae.decode((z / 0.18215).to(dev)).float().cpu()
    ref = fx["output"].float()
    rel = relerr(out, ref)
    mse = ((out - ref) ** 2).mean().item()
    psnr = 10 * math.log10(ref.abs().max().item() ** 2 / mse)      # peak = max |ref| (as tests/test_vae_gpu.py)
    print(f"{name}: rel {rel:.4f} PSNR {psnr:.1f} dB")
    assert tuple(out.shape) == tuple(ref.shape)
    assert rel < 4e-2 and psnr > 35.0
