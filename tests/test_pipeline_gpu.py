"""End-to-end: the reference's stage-1 call sequence (pipeline_i2v_eval_v01.py:62-98)
on the MI355X engine built from the shipped YAML (widths reduced so the CPU oracle of the
whole clip finishes in seconds), compared with the oracle's sampler + VAE decode."""
import os
import tempfile

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stage1_clip_create_model_sample_decode(dev):
    from hi3d_hip import synth
    from oracle import hi3d_oracle as O
    from vtdm.model import create_model
    from vtdm.util import tensor2vid

    from conftest import shrink_conditioner
    y = shrink_conditioner(yaml.safe_load(open(os.path.join(ROOT, "hi3d-official_amd", "configs", "inference-v01.yaml"))))
    P = y["model"]["params"]
    T, steps, hw = 4, 4, 8
    P["num_samples"] = T
    P["network_config"]["params"]["model_channels"] = 64
    P["first_stage_config"]["params"]["ddconfig"]["ch"] = 64
    P["sampler_config"]["params"]["num_steps"] = steps
    P["sampler_config"]["params"]["verbose"] = False
    P["sampler_config"]["params"]["guider_config"]["params"]["num_frames"] = T
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as fh:
        yaml.safe_dump(y, fh)
    model = create_model(fh.name)
    synth.fill_module_(model, seed=5)                      # keys: model.diffusion_model.* / first_stage_model.*
    sd = {k: v.clone().float() for k, v in model.state_dict().items()}
    model = model.to(dev)
    model.sampler.device = dev

    x0, c, uc = synth.synth_conditioning(T, hw, hw, stage=1, seed=9)
    cd = {k: v.to(dev) for k, v in c.items()}
    ucd = {k: v.to(dev) for k, v in uc.items()}
    extra = {"image_only_indicator": torch.zeros(2, T, device=dev), "num_video_frames": T}

    def denoiser(inp, sigma, cc):                          # pipeline_i2v_eval_v01.py:85-88
        return model.denoiser(model.model, inp, sigma, cc, **extra)

    samples = model.sampler(denoiser, x0.clone().to(dev), cond=cd, uc=ucd)       # :92
    images = model.decode_first_stage(samples)                                  # :94
    video = images.reshape(1, T, 3, 8 * hw, 8 * hw).permute(0, 2, 1, 3, 4)       # '(b t) c h w -> b c t h w'
    frames = tensor2vid(video)
    assert len(frames) == T and frames[0].shape == (8 * hw, 8 * hw, 3) and frames[0].dtype.name == "uint8"

    ucfg = dict(model.model.diffusion_model.cfg)
    with torch.no_grad():
        ref_lat = O.euler_edm_sample(sd, ucfg, x0, c, uc, T, steps, 2.5)
        ref_img = O.vae_decode(sd, P["first_stage_config"]["params"]["ddconfig"], ref_lat)
    lat_rel = ((samples.float().cpu() - ref_lat).abs().max() / ref_lat.abs().max()).item()
    img = images.float().cpu()
    img_rel = ((img - ref_img).abs().max() / ref_img.abs().max()).item()
    cos = torch.nn.functional.cosine_similarity(img.flatten(), ref_img.flatten(), dim=0).item()
    print(f"pipeline: latent rel {lat_rel:.4f}  image rel {img_rel:.4f} cos {cos:.6f}")
    assert lat_rel < 6e-2 and img_rel < 8e-2 and cos > 0.998
