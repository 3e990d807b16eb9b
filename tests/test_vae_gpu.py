"""VAE decode parity vs golden outputs of the reference AutoencoderKL (CPU fp32).
Tolerance: bf16 activations through ~30 conv/norm layers -> max-abs rel <= 4e-2,
PSNR of the decoded image (peak = max |ref|) >= 35 dB."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["vae_tiny", "vae_full_lat8"])
def test_vae_decode_matches_reference_golden(dev, name):
    from hi3d_hip import synth
    from sgm.models.autoencoder import AutoencoderKL
    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    ae = AutoencoderKL(embed_dim=4, ddconfig=fx["ddconfig"], lossconfig={"target": "torch.nn.Identity"})
    synth.fill_module_(ae, fx["weight_seed"], prefix=fx["key_prefix"])
    ae = ae.to(dev)
    out = ae.decode((fx["z"] / 0.18215).to(dev)).float().cpu()
    ref = fx["output"]
    rel = ((out - ref).abs().max() / ref.abs().max()).item()
    psnr = 10 * math.log10(ref.abs().max().item() ** 2 / ((out - ref) ** 2).mean().item())
    print(f"{name}: rel {rel:.4f} psnr {psnr:.1f} dB")
    assert out.shape == ref.shape and rel < 4e-2 and psnr > 35.0


def test_engine_decode_first_stage_chunks(dev):
    """DiffusionEngine.decode_first_stage: 1/scale_factor, chunking by
    en_and_decode_n_samples_a_time, concatenation (models/diffusion.py:117-135)."""
    from hi3d_hip import synth
    from sgm.models.autoencoder import AutoencoderKL
    from sgm.models.diffusion import DiffusionEngine
    fx = torch.load(os.path.join(GOLD, "vae_tiny.pt"), weights_only=False)
    eng = DiffusionEngine.__new__(DiffusionEngine)
    torch.nn.Module.__init__(eng)
    eng.first_stage_model = AutoencoderKL(embed_dim=4, ddconfig=fx["ddconfig"])
    synth.fill_module_(eng.first_stage_model, fx["weight_seed"], prefix=fx["key_prefix"])
    eng.first_stage_model.to(dev)
    eng.scale_factor, eng.en_and_decode_n_samples_a_time = 0.18215, 1
    out = eng.decode_first_stage(fx["z"].to(dev)).float().cpu()
    rel = ((out - fx["output"]).abs().max() / fx["output"].abs().max()).item()
    assert rel < 4e-2


def test_chunks_on_two_streams_equal_one_stream(dev, monkeypatch):
    """Round 4: the chunk loop of decode_first_stage / encode_first_stage alternates chunks over two HIP streams
    (runtime_vae.run_chunks).  Same launches on the same data: bit-identical to the one-stream loop, repeatedly (a race on a
    shared scratch buffer or a missing join would show up as a difference), also for a full-width decoder with split-K
    launches and for an odd number of chunks."""
    from hi3d_hip import runtime_vae, synth
    from sgm.models.autoencoder import AutoencoderKL
    from sgm.models.diffusion import DiffusionEngine
    for name, n in (("vae_tiny", 5), ("vae_full_lat8", 6)):
        fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
        eng = DiffusionEngine.__new__(DiffusionEngine)
        torch.nn.Module.__init__(eng)
        eng.first_stage_model = AutoencoderKL(embed_dim=4, ddconfig=fx["ddconfig"])
        synth.fill_module_(eng.first_stage_model, fx["weight_seed"], prefix=fx["key_prefix"])
        eng.first_stage_model.to(dev)
        eng.scale_factor, eng.en_and_decode_n_samples_a_time = 0.18215, 1
        z0 = fx["z"].to(dev)
        z = torch.cat([z0[:1] * (1.0 + 0.1 * i) for i in range(n)])
        monkeypatch.setattr(runtime_vae, "VAE_STREAMS", 1)
        ref = eng.decode_first_stage(z)
        for ns in (2, 3, 2):
            monkeypatch.setattr(runtime_vae, "VAE_STREAMS", ns)
            for _ in range(2):
                out = eng.decode_first_stage(z)
                assert torch.equal(out, ref), (name, ns)
        # a model whose FIRST call goes through the two-stream loop: its runtime (weight re-layout) is built inside chunk 0, on the
        # caller's stream -- the side stream must not start before that is complete (round 4: NaN frames otherwise)
        for rep in range(3):
            junk = torch.full((64 << 20,), float("nan"), device=dev)        # (poison what the allocator hands out next)
            del junk
            fresh = AutoencoderKL(embed_dim=4, ddconfig=fx["ddconfig"])
            synth.fill_module_(fresh, fx["weight_seed"], prefix=fx["key_prefix"])
            eng.first_stage_model = fresh.to(dev)
            assert torch.equal(eng.decode_first_stage(z), ref), (name, "first call", rep)
        # ... and with a RAGGED last chunk (5 frames, 2 per call: 2 + 2 + 1) as the first call of a fresh model, the odd chunk on a
        # side stream when there are three: per-stream scratch grown for another chunk size, runtime built before the fork
        eng.en_and_decode_n_samples_a_time = 2
        monkeypatch.setattr(runtime_vae, "VAE_STREAMS", 1)
        ref2 = eng.decode_first_stage(z[:5])
        # (the decoder is per frame, but a two-frame call has other M -- other tile variants / split-K choices than a one-frame
        # call: equal up to accumulation order, not bit for bit)
        assert ((ref2.float() - ref[:5].float()).abs().max() / ref.float().abs().max()).item() < 2e-2
        for ns in (3, 2):
            monkeypatch.setattr(runtime_vae, "VAE_STREAMS", ns)
            junk = torch.full((64 << 20,), float("nan"), device=dev)
            del junk
            fresh = AutoencoderKL(embed_dim=4, ddconfig=fx["ddconfig"])
            synth.fill_module_(fresh, fx["weight_seed"], prefix=fx["key_prefix"])
            eng.first_stage_model = fresh.to(dev)
            assert torch.equal(eng.decode_first_stage(z[:5]), ref2), (name, "ragged first call", ns)
        eng.en_and_decode_n_samples_a_time = 1
        if name == "vae_tiny":                                  # the encoder's chunk loop (posterior .sample(): same noise order)
            x = ref.clamp(-1, 1)
            monkeypatch.setattr(runtime_vae, "VAE_STREAMS", 1)
            torch.manual_seed(3); a = eng.encode_first_stage(x)
            monkeypatch.setattr(runtime_vae, "VAE_STREAMS", 2)
            torch.manual_seed(3); b = eng.encode_first_stage(x)
            assert a.shape == z.shape and torch.equal(a, b)


@pytest.mark.parametrize("name", ["vae_enc_tiny", "vae_enc_full_64"])
def test_vae_encode_matches_reference_golden(dev, name):
    """Encoder + quant_conv + posterior: mode vs the reference mean, and sample with the
    reference's own CPU noise draw vs its posterior.sample()."""
    from hi3d_hip import synth
    from sgm.models.autoencoder import AutoencoderKL, AutoencoderKLModeOnly
    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    ae = AutoencoderKL(embed_dim=4, ddconfig=fx["ddconfig"])
    synth.fill_module_(ae, fx["weight_seed"], prefix=fx["key_prefix"])
    ae = ae.to(dev)
    mean_ref = fx["moments"][:, :4]
    z = ae.encode(fx["x"].to(dev), noise=fx["sample_noise"]).float().cpu()
    rel_s = ((z - fx["z_sampled"]).abs().max() / fx["z_sampled"].abs().max()).item()
    mo = AutoencoderKLModeOnly(embed_dim=4, ddconfig=fx["ddconfig"])
    synth.fill_module_(mo, fx["weight_seed"], prefix=fx["key_prefix"])
    zm = mo.to(dev).encode(fx["x"].to(dev)).float().cpu()
    rel_m = ((zm - mean_ref).abs().max() / mean_ref.abs().max()).item()
    print(f"{name}: sample rel {rel_s:.4f}  mode rel {rel_m:.4f}")
    assert rel_s < 4e-2 and rel_m < 4e-2
    torch.manual_seed(1234)                       # default path draws the posterior noise on the CPU generator, like the reference
    z2 = ae.encode(fx["x"].to(dev)).float().cpu()
    assert torch.equal(z2, z)


def test_cond_frame_embedder_matches_reference_encoder(dev):
    """VideoPredictionEmbedderWithEncoder (encoders/modules.py:951-1025): VAE mode of the conditioning
    frame, * scale_factor, `(b t) c h w -> b () (t c) h w`, repeated n_copies times."""
    from hi3d_hip import synth
    from sgm.modules.encoders.modules import VideoPredictionEmbedderWithEncoder
    fx = torch.load(os.path.join(GOLD, "vae_enc_tiny.pt"), weights_only=False)
    enc_cfg = {"target": "sgm.models.autoencoder.AutoencoderKLModeOnly",
               "params": {"embed_dim": 4, "ddconfig": fx["ddconfig"], "lossconfig": {"target": "torch.nn.Identity"}}}
    n = fx["x"].shape[0]
    emb = VideoPredictionEmbedderWithEncoder(n_cond_frames=1, n_copies=3, encoder_config=enc_cfg, is_ae=True,
                                             scale_factor=0.5, disable_encoder_autocast=True, en_and_decode_n_samples_a_time=1)
    synth.fill_module_(emb.encoder, fx["weight_seed"], prefix=fx["key_prefix"])
    out = emb.to(dev)(fx["x"].to(dev)).float().cpu()
    mean = fx["moments"][:, :4] * 0.5
    ref = mean[:, None].expand(n, 3, *mean.shape[1:]).reshape(n * 3, *mean.shape[1:])
    assert out.shape == ref.shape
    assert ((out - ref).abs().max() / ref.abs().max()).item() < 4e-2
    two = VideoPredictionEmbedderWithEncoder(n_cond_frames=2, n_copies=2, encoder_config=enc_cfg, is_ae=True)
    synth.fill_module_(two.encoder, fx["weight_seed"], prefix=fx["key_prefix"])
    if n % 2 == 0:                                   # two conditioning frames stack on the channel axis
        o2 = two.to(dev)(fx["x"].to(dev)).float().cpu()
        m2 = fx["moments"][:, :4].reshape(n // 2, 1, 8, *mean.shape[2:]).expand(n // 2, 2, 8, *mean.shape[2:]).reshape(n, 8, *mean.shape[2:])
        assert o2.shape == m2.shape and ((o2 - m2).abs().max() / m2.abs().max()).item() < 4e-2


def test_conv3x3_bottom_right_padding(dev):
    import torch.nn.functional as F
    from hi3d_hip import ops
    from hi3d_hip.pack import pack_conv3x3
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 64, 10, 12), generator=g).to(torch.bfloat16)
    w = (torch.randn((128, 64, 3, 3), generator=g) * (576 ** -0.5)).to(torch.bfloat16).float()
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w, stride=2)
    xt = x.permute(0, 2, 3, 1).contiguous().reshape(-1, 64)
    out = ops.gemm(xt.to(dev), pack_conv3x3(w).to(dev), M=2 * 5 * 6, N=128, K=576,
                   conv3x3=dict(Hin=10, Win=12, Cin=64, Hout=5, Wout=6, stride=2, up2x=0, pad_br_only=1))
    got = out.float().cpu().reshape(2, 5, 6, 128).permute(0, 3, 1, 2)
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 1.2e-2


@pytest.mark.parametrize("name", ["videodec_tiny", "videodec_full_lat8", "videodec_full_lat32", "videodec_tiny_k3", "videodec_full_lat16_k3"])
def test_video_decoder_matches_reference_golden(dev, name):
    """Temporal VAE decoder (north_star's 'AutoencoderKLTemporalDecoder' = VideoDecoder) through
    AutoencodingEngine + the DiffusionEngine.decode_first_stage `timesteps` hook.  The *_k3 goldens were made with
    video_kernel_size=3, the reference class's default (isotropic 3x3x3 time_stack / time_mix_conv); the others with the
    [3, 1, 1] that SVD / Hi3D configure."""
    from hi3d_hip import synth
    from sgm.models.autoencoder import AutoencodingEngine
    from sgm.models.diffusion import DiffusionEngine
    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    dd = fx["ddconfig"]
    ae = AutoencodingEngine(
        encoder_config={"target": "sgm.modules.diffusionmodules.model.Encoder", "params": dd},
        decoder_config={"target": "sgm.modules.autoencoding.temporal_ae.VideoDecoder",
                        "params": dict(dd, video_kernel_size=fx.get("video_kernel_size", [3, 1, 1]))},
        regularizer_config={"target": "sgm.modules.autoencoding.regularizers.DiagonalGaussianRegularizer"})
    assert {k: tuple(v.shape) for k, v in ae.state_dict().items()} == {k: tuple(v) for k, v in fx["shapes"].items()}
    synth.fill_module_(ae, fx["weight_seed"], prefix=fx["key_prefix"])
    ae = ae.to(dev)
    out = ae.decode(fx["z"].to(dev), timesteps=fx["T"]).float().cpu()
    ref = fx["output"]
    rel = ((out - ref).abs().max() / ref.abs().max()).item()
    psnr = 10 * math.log10(ref.abs().max().item() ** 2 / ((out - ref) ** 2).mean().item())
    print(f"{name}: rel {rel:.4f} psnr {psnr:.1f} dB")
    assert rel < 4e-2 and psnr > 35.0
    eng = DiffusionEngine.__new__(DiffusionEngine)
    torch.nn.Module.__init__(eng)
    eng.first_stage_model, eng.scale_factor, eng.en_and_decode_n_samples_a_time = ae, 0.5, fx["T"]
    out2 = eng.decode_first_stage(fx["z"].to(dev) * 0.5).float().cpu()          # chunk of T frames -> timesteps=T
    assert ((out2 - ref).abs().max() / ref.abs().max()).item() < 4e-2


def test_autoencoding_engine_encode_matches_reference_golden(dev):
    """AutoencodingEngine.encode (models/autoencoder.py:196-209; no quant convs): raw moments, the sampled posterior with the
    reference's own CPU noise draw, the default CPU-generator draw, and `sample: false` (the mode)."""
    from hi3d_hip import synth
    from sgm.models.autoencoder import AutoencodingEngine
    fx = torch.load(os.path.join(GOLD, "engine_enc_tiny.pt"), weights_only=False)
    dd = fx["ddconfig"]

    def build(reg_params=None):
        ae = AutoencodingEngine(
            encoder_config={"target": "sgm.modules.diffusionmodules.model.Encoder", "params": dd},
            decoder_config={"target": "sgm.modules.diffusionmodules.model.Decoder", "params": dd},
            regularizer_config={"target": "sgm.modules.autoencoding.regularizers.DiagonalGaussianRegularizer", "params": reg_params})
        assert {k: tuple(v.shape) for k, v in ae.state_dict().items()} == fx["shapes"]
        synth.fill_module_(ae, fx["weight_seed"], prefix=fx["key_prefix"])
        return ae.to(dev)
    ae = build()
    x = fx["x"].to(dev)
    mom, log = ae.encode(x, unregularized=True)
    assert log == {} and mom.shape == fx["moments"].shape
    r = lambda a, b: ((a.float().cpu() - b).abs().max() / b.abs().max()).item()
    z = ae.encode(x, noise=fx["sample_noise"])
    print(f"engine encode: moments rel {r(mom, fx['moments']):.4f}  sample rel {r(z, fx['z_sampled']):.4f}")
    assert r(mom, fx["moments"]) < 4e-2 and r(z, fx["z_sampled"]) < 4e-2
    torch.manual_seed(4321)                          # the seed the reference run drew its posterior noise with
    z2, log2 = ae.encode(x, return_reg_log=True)
    assert torch.equal(z2, z) and isinstance(log2, dict)
    zm = build({"sample": False}).encode(x)
    assert r(zm, fx["moments"][:, :dd["z_channels"]]) < 4e-2
