#!/usr/bin/env python
"""Wall time of ONE clip through the whole drop-in, at the reference's full sizes, on one MI355X.

  stage 2 (pipeline_i2v_eval_v02.py:77-141): create_model(configs/inference-v02.yaml) -- the 1.52 B-parameter VideoUNet, the
      full AutoencoderKL, the OpenCLIP ViT-H/14 tower, the MiDaS DPT-hybrid depth network, all at the YAML's widths --
      add_custom_cond -> GeneralConditioner -> per-frame encode of 16 frames @ 1024^2 -> 25 re-noise + Euler-EDM + CFG steps
      -> decode_first_stage (16 frames @ 1024^2) -> tensor2vid;
  stage 1 (pipeline_i2v_eval_v01.py:62-98): create_model(configs/inference-v01.yaml), conditioner on one 512^2 image,
      25 steps at 16 views x 512^2, decode.

Weights are random (hi3d_hip.synth, drawn on the device: there are no checkpoints in the tree); the input is a synthetic
clip.  Nothing here is a parity check (tests/test_pipeline_gpu.py is, at reduced widths against the oracle chain): this tool
exercises every runtime at its production size in one process and reports where a clip's wall time goes.
usage: python tools/clip_e2e.py [s2|s1|both]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd"))
import torch  # noqa: E402

from hi3d_hip import pipelines, synth  # noqa: E402
from sgm.util import ParamTree  # noqa: E402
from vtdm.model import create_model  # noqa: E402
from vtdm.util import tensor2vid  # noqa: E402

dev = torch.device("cuda:0")


class Clock:
    def __init__(self):
        self.t, self.out = None, {}

    def lap(self, name):
        torch.cuda.synchronize()
        now = time.perf_counter()
        if self.t is not None:
            self.out[name] = round(now - self.t, 4)
        self.t = now


def build(yaml_name):
    t0 = time.perf_counter()
    ParamTree.skip_init = True
    try:
        with torch.device(dev):
            model = create_model(os.path.join(ROOT, "hi3d-official_amd", "configs", yaml_name))
    finally:
        ParamTree.skip_init = False
    model = model.to(dev)
    synth.fill_module_on_device_(model, seed=1)
    for e in model.conditioner.embedders:           # random 70-layer depth network: damp the residual tails as the fixtures do (synth.py)
        if type(e).__name__ == "DepthEmbedder":
            e.load_state_dict(synth.damp_residual_tails({k: v.clone() for k, v in e.state_dict().items()}, 0.25))
    model.sampler.device = dev
    torch.cuda.synchronize()
    n = sum(p.numel() for p in model.parameters())
    return model, n, time.perf_counter() - t0


def run_s2():
    model, n, tb = build("inference-v02.yaml")
    T, HW = 16, 1024
    g = torch.Generator(device=dev).manual_seed(5)
    video = torch.rand((1, 3, T, HW, HW), device=dev, generator=g) * 2 - 1          # the stage-1 clip, upsampled (v02.py:77-95)
    res = {}
    for it in ("first clip (kernel warm-up, weight re-layout, graph capture)", "second clip"):
        ck = Clock()
        ck.lap("start")
        with torch.no_grad():
            batch = model.add_custom_cond({"video": video, "elevation": torch.tensor([10.0], device=dev)}, infer=True)
            c, uc = model.conditioner.get_unconditional_conditioning(
                batch, force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"])
            ck.lap("conditioner (CLIP token, depth unshuffle, 16 conditioning latents)")
            zs = [model.encode_first_stage_with_noise(video[0, :, t].unsqueeze(0), None) for t in range(T)]
            z = torch.cat(zs, 0)
            ck.lap("per-frame VAE encode, 16 x 1024^2")
            lat = pipelines.stage2_refine(model, None, c, uc, decode=False, z_frames=z)
            ck.lap(f"{model.sampler.num_steps} refine steps (blend + Euler-EDM + CFG, 16 views x 1024^2)")
            img = model.decode_first_stage(lat)
            ck.lap("VAE decode, 16 x 1024^2")
            frames = tensor2vid(img.reshape(1, T, 3, HW, HW).permute(0, 2, 1, 3, 4))
            ck.lap("tensor2vid (to uint8 frames on the host)")
        assert torch.isfinite(img).all() and len(frames) == T
        ck.out["total_s"] = round(sum(ck.out.values()), 3)
        res[it] = ck.out
    return {"clip": "stage 2, 16 views @ 1024^2 (inference-v02.yaml, full widths)", "parameters": n, "build_s": round(tb, 1), **res}


def run_s1():
    model, n, tb = build("inference-v01.yaml")
    T, HW = 16, 512
    g = torch.Generator(device=dev).manual_seed(6)
    image = torch.rand((1, 3, HW, HW), device=dev, generator=g) * 2 - 1
    res = {}
    for it in ("first clip (kernel warm-up, weight re-layout, graph capture)", "second clip"):
        ck = Clock()
        ck.lap("start")
        with torch.no_grad():
            batch = model.add_custom_cond({"video": image.unsqueeze(2), "elevation": torch.tensor([10.0], device=dev)}, infer=True)   # (v01.py:66-73)
            c, uc = model.conditioner.get_unconditional_conditioning(
                batch, force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"])
            ck.lap("conditioner (CLIP token, conditioning latent)")
            lat = pipelines.stage1_denoise(model, c, uc, T, HW // 8, HW // 8, decode=False)
            ck.lap(f"{model.sampler.num_steps} Euler-EDM + CFG steps, 16 views x 512^2")
            img = model.decode_first_stage(lat)
            ck.lap("VAE decode, 16 x 512^2")
            frames = tensor2vid(img.reshape(1, T, 3, HW, HW).permute(0, 2, 1, 3, 4))
            ck.lap("tensor2vid (to uint8 frames on the host)")
        assert torch.isfinite(img).all() and len(frames) == T
        ck.out["total_s"] = round(sum(ck.out.values()), 3)
        res[it] = ck.out
    return {"clip": "stage 1, 16 views @ 512^2 (inference-v01.yaml, full widths)", "parameters": n, "build_s": round(tb, 1), **res}


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    out = []
    if which in ("s2", "both"):
        out.append(run_s2())
        torch.cuda.empty_cache()
    if which in ("s1", "both"):
        out.append(run_s1())
    print(json.dumps(out, indent=1))
