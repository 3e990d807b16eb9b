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
