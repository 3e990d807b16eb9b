"""Conditioner CLIP vision towers on the gfx950 ViT runtime: vs the golden vectors of an independent
implementation (HuggingFace CLIPVisionModelWithProjection, oracle/gen_golden_clip.py), vs the CPU oracle at the real
ViT-H/14 and ViT-L/14 sizes, and the two embedder classes end to end.  bf16 storage, fp32 accumulate: tolerance 3e-2 of
the output range and cosine >= 0.999 through 24-32 layers."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle"))


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()


def cos(a, b):
    return F.cosine_similarity(a.float().cpu().flatten(), b.float().cpu().flatten(), dim=0).item()


@pytest.mark.parametrize("kind,n", [("gelu", 4096), ("quick_gelu", 40)])
def test_act_bf16(dev, kind, n):
    from hi3d_hip import ops
    x = (torch.randn(n, generator=torch.Generator().manual_seed(3)) * 3).to(torch.bfloat16)
    ref = F.gelu(x.float()) if kind == "gelu" else x.float() * torch.sigmoid(1.702 * x.float())
    out = ops.act_(x.to(dev).clone(), kind)
    assert rel(out, ref) < 6e-3                       # one bf16 rounding of the result
    with pytest.raises(ops._l.Hi3dError):
        ops.act_(torch.zeros(7, device=dev, dtype=torch.bfloat16), kind)


def test_l2_normalize_rows(dev):
    from hi3d_hip import ops
    x = torch.randn((5, 768), generator=torch.Generator().manual_seed(4))
    x[3] = 0
    out = ops.l2_normalize_rows_(x.to(dev).clone()).cpu()
    n = x.norm(dim=1, keepdim=True)
    assert torch.allclose(out, x / torch.where(n == 0, torch.ones_like(n), n), atol=1e-6)


@pytest.mark.parametrize("name", ["clip_vith_like", "clip_vitl_like"])
def test_vit_runtime_matches_independent_implementation(dev, name):
    from gen_golden_clip import clip_shapes
    from hi3d_hip import synth
    from hi3d_hip.runtime_vit import ViTRuntime
    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    c = fx["cfg"]
    sd = synth.synth_state_dict(clip_shapes(c["width"], c["layers"], c["patch"], c["grid"], c["out_dim"]), fx["seed"])
    rt = ViTRuntime(sd, "visual.", c["heads"], c["act"], dev)
    out = rt.forward(fx["img"].to(dev))
    assert out.shape == fx["out"].shape and out.dtype == torch.float32
    print(f"{name}: rel {rel(out, fx['out']):.4f} cos {cos(out, fx['out']):.6f}")
    assert rel(out, fx["out"]) < 3e-2 and cos(out, fx["out"]) > 0.999


@pytest.mark.parametrize("arch", ["ViT-H-14", "ViT-L-14"])
def test_vit_runtime_full_size_vs_oracle(dev, arch):
    """The real tower geometries (32 x 1280 / 16 heads of 80; 24 x 1024 / 16 heads of 64), synthetic weights, one
    224 x 224 image, against the fp32 CPU restatement."""
    from hi3d_hip import synth
    from hi3d_hip.runtime_vit import ViTRuntime
    from oracle import hi3d_oracle as O
    from sgm.modules.encoders.modules import CLIP_VISUAL_ARCHS, clip_visual_shapes
    a = CLIP_VISUAL_ARCHS[arch]
    sd = synth.synth_state_dict({"visual." + k: s for k, s in clip_visual_shapes(**a).items()}, 31)
    img = torch.randn((1, 3, 224, 224), generator=torch.Generator().manual_seed(32))
    ref = O.clip_visual(sd, img, a["heads"], a["act"])
    out = ViTRuntime(sd, "visual.", a["heads"], a["act"], dev).forward(img.to(dev))
    print(f"{arch}: rel {rel(out, ref):.4f} cos {cos(out, ref):.6f}")
    assert rel(out, ref) < 4e-2 and cos(out, ref) > 0.999


def test_openclip_prediction_embedder_end_to_end(dev):
    """FrozenOpenCLIPImagePredictionEmbedder: preprocess (resize 96 -> 224, CLIP normalisation), tower, n_copies repeat;
    parameters under the reference's names."""
    from hi3d_hip import synth
    from oracle import hi3d_oracle as O
    from sgm.modules.encoders.modules import CLIP_VISUAL_ARCHS, FrozenOpenCLIPImagePredictionEmbedder
    e = FrozenOpenCLIPImagePredictionEmbedder(
        open_clip_embedding_config={"target": "sgm.modules.encoders.modules.FrozenOpenCLIPImageEmbedder", "params": {"arch": "ViT-tiny-H"}},
        n_cond_frames=1, n_copies=3)
    synth.fill_module_(e, seed=6)
    sd = {k: v.clone().float() for k, v in e.state_dict().items()}
    assert "open_clip.model.visual.transformer.resblocks.1.mlp.c_fc.weight" in sd
    img = torch.rand((2, 3, 96, 96), generator=torch.Generator().manual_seed(7)) * 2 - 1
    ref = O.openclip_image_embedder(sd, img, CLIP_VISUAL_ARCHS["ViT-tiny-H"]["heads"], 1, 3)
    out = e.to(dev)(img.to(dev))
    assert out.shape == (6, 1, 1024) == ref.shape
    assert rel(out, ref) < 3e-2 and cos(out, ref) > 0.999
    assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[2])        # the n_copies repeat


def test_aes_embedder_end_to_end(dev):
    """AesEmbedder: middle frame -> 224 x 384 bilinear -> centre crop -> CLIP ViT-L tower -> L2 norm -> aesthetic MLP ->
    [score, sinusoid(100 score)].  The MLP runs fp32-accurate (split-precision GEMM); the score error is the bf16
    tower's, amplified 100x into the embedding's argument: the score is compared, the embedding must be exactly the
    embedding of the score that was produced."""
    from hi3d_hip import ops, synth
    from oracle import hi3d_oracle as O
    from vtdm.encoders import AesEmbedder
    e = AesEmbedder(arch="ViT-tiny-L")
    synth.fill_module_(e, seed=8)
    sd = {k: v.clone().float() for k, v in e.state_dict().items()}
    vid = torch.rand((2, 3, 5, 64, 96), generator=torch.Generator().manual_seed(9)) * 2 - 1
    ref = O.aes_embedder(sd, vid, heads=2)
    out = e.to(dev)(vid.to(dev))
    assert out.shape == (2, 256) == ref.shape
    assert (out[:, 0].cpu() - ref[:, 0]).abs().max() < 2e-2 * max(1.0, ref[:, 0].abs().max().item())
    assert torch.equal(out[:, 1:], ops.timestep_embedding(out[:, 0].contiguous() * 100, 255))
    # the MLP alone, on the oracle's features: fp32-accurate
    f = torch.randn((4, 768), generator=torch.Generator().manual_seed(10))
    f = f / f.norm(dim=1, keepdim=True)
    h, r = f.to(dev), f.clone()
    for (wp, b, o, op, kp), (i, _, _) in zip(e._packed_mlp(dev), e.MLP_DIMS):
        h = ops.gemm(e._split(h, kp), wp, M=4, N=op, K=3 * kp, bias=b, out_fp32=True)[:, :o].contiguous()
        r = F.linear(r, sd[f"aesthetic_mlp.layers.{i}.weight"], sd[f"aesthetic_mlp.layers.{i}.bias"])
    assert (h.cpu() - r).abs().max() < 2e-4 * max(1.0, r.abs().max().item())


@pytest.mark.parametrize("kind,hw", [("clip224", (1024, 1024)), ("clip224", (512, 576)), ("aes", (1024, 1024)), ("aes", (320, 512))])
def test_resample_image_matches_oracle(dev, kind, hw):
    """hi3d_resample_axis (two banded passes + fused CLIP affine) vs the oracle's restatement of the reference's resizes:
    kornia.geometry.resize(bicubic, align_corners, antialias) (modules.py:619-628) / F.interpolate(bilinear) + crop
    (vtdm/encoders.py:80-83).  fp32 throughout, the 70-odd taps of a pixel summed in another order than the reference's
    blur-then-interpolate: 5e-5 absolute on values up to 2.6."""
    from oracle import hi3d_oracle as O
    from hi3d_hip import ops
    img = torch.rand((2, 3) + hw, generator=torch.Generator().manual_seed(7)) * 2 - 1
    mean, std = torch.tensor(O.CLIP_MEAN), torch.tensor(O.CLIP_STD)
    # reference in float64: aten computes the source coordinate o * (in - 1) / (out - 1) in the tensor's precision, which in
    # fp32 at 1024 pixels is worth 6e-5 of a pixel -- the fp32 reference run disagrees with ITSELF in fp64 by that much; the
    # tap tables here are built in fp64, so the kernel is compared with the exact arithmetic (and loosely with the fp32 run)
    if kind == "clip224":
        ref = O.kornia_resize(img.double(), (224, 224))
        ref32 = O.kornia_resize(img, (224, 224))
    else:
        ref = F.interpolate(img.double(), [224, 384], mode="bilinear")[:, :, :, 80:304]
        ref32 = F.interpolate(img, [224, 384], mode="bilinear")[:, :, :, 80:304]
    norm = lambda t: ((t + 1) * 0.5 - mean.view(1, 3, 1, 1).to(t.dtype)) / std.view(1, 3, 1, 1).to(t.dtype)
    ref, ref32 = norm(ref).float(), norm(ref32)
    out = ops.resample_image(img.to(dev), kind, scale=(0.5 / std).to(dev), shift=((0.5 - mean) / std).to(dev))
    assert out.shape == ref.shape
    assert (out.cpu() - ref).abs().max() < 1e-5 and (out.cpu() - ref32).abs().max() < 3e-4
    plain = ops.resample_image(img.to(dev), kind)            # no affine
    assert (plain.cpu() * (0.5 / std).view(1, 3, 1, 1) + ((0.5 - mean) / std).view(1, 3, 1, 1) - ref).abs().max() < 1e-5
