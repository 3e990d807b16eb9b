"""Stage-2 depth conditioner (vtdm/encoders.py:15-53: MiDaS DPT-hybrid + min-max + 3x3 pixel-unshuffle) on the gfx950
kernels.  Small kernels vs torch; the whole network vs the golden produced by the reference's own MiDaS code
(oracle/gen_golden_dpt.py: every reassembled layer and the depth map); the DepthEmbedder class vs the CPU oracle.
bf16 storage / fp32 accumulate through ~70 convolutions and 12 transformer blocks: the stated tolerances are on the
depth map's range."""
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


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def bf(t):
    return t.to(torch.bfloat16)


def test_add_act(dev):
    from hi3d_hip import ops
    x, y = bf(rnd((3, 40), 1, 2.0)), bf(rnd((3, 40), 2, 2.0))
    assert torch.equal(ops.act_(x.to(dev).clone(), "relu").cpu(), F.relu(x))
    assert torch.equal(ops.add_act(x.to(dev), None, "relu").cpu(), F.relu(x))
    assert torch.equal(ops.add_act(x.to(dev), y.to(dev), "identity").cpu(), bf(x.float() + y.float()))
    assert torch.equal(ops.add_act(x.to(dev), y.to(dev), "relu").cpu(), bf(F.relu(x.float() + y.float())))
    with pytest.raises(ops._l.Hi3dError):
        ops.add_act(x.to(dev), y.to(dev)[:2], "relu")


@pytest.mark.parametrize("H,W", [(64, 96), (30, 22), (33, 17)])
def test_stem_conv_and_pools(dev, H, W):
    """TF-'SAME' 7x7 stride-2 convolution, 3x3 stride-2 max pool and the stride-2 pixel pick, even and odd sizes,
    against the padding rule timm uses (pad total = max((ceil(n/2) - 1) * 2 + k - n, 0), the odd unit after)."""
    from hi3d_hip import ops
    from timm_standin import pad_same
    x, w = rnd((2, 3, H, W), 5), rnd((64, 3, 7, 7), 6, 147 ** -0.5)
    ref = F.conv2d(pad_same(x, 7, 2), w, None, 2)
    out = ops.dpt_stem_conv(x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(2, 3, 1, 0).contiguous().to(dev))
    assert out.shape == (2, (H + 1) // 2, (W + 1) // 2, 64)
    assert rel(out.permute(0, 3, 1, 2), ref) < 6e-3
    f = bf(rnd((2, 16, H, W), 7))
    mp = ops.pool2(f.permute(0, 2, 3, 1).contiguous().to(dev), "max3").permute(0, 3, 1, 2).cpu()
    assert torch.equal(mp, F.max_pool2d(pad_same(f.float(), 3, 2, value=-float("inf")), 3, 2).to(torch.bfloat16))
    pk = ops.pool2(f.permute(0, 2, 3, 1).contiguous().to(dev), "pick").permute(0, 3, 1, 2).cpu()
    assert torch.equal(pk, f[:, :, ::2, ::2])


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("size", [((12, 20), (24, 40)), ((24, 24), (7, 13)), ((16, 16), (16, 16)), ((9, 5), (30, 11))])
def test_resize_bilinear(dev, align, size):
    from hi3d_hip import ops
    (Hi, Wi), (Ho, Wo) = size
    x = rnd((2, 8, Hi, Wi), 11)
    ref = F.interpolate(x, (Ho, Wo), mode="bilinear", align_corners=align)
    o32 = ops.resize_bilinear(x.permute(0, 2, 3, 1).contiguous().to(dev), Ho, Wo, align).permute(0, 3, 1, 2).cpu()
    assert torch.allclose(o32, ref, atol=2e-5)
    o16 = ops.resize_bilinear(bf(x).permute(0, 2, 3, 1).contiguous().to(dev), Ho, Wo, align).permute(0, 3, 1, 2)
    assert rel(o16, F.interpolate(bf(x).float(), (Ho, Wo), mode="bilinear", align_corners=align)) < 6e-3
    one = ops.resize_bilinear(x[:, :1].permute(0, 2, 3, 1).contiguous().to(dev), Ho, Wo, align).permute(0, 3, 1, 2).cpu()
    assert torch.allclose(one, ref[:, :1], atol=2e-5)


def test_head_out_and_normalize_unshuffle(dev):
    from hi3d_hip import ops
    x, w = bf(rnd((1000, 32), 21)), rnd((32,), 22, 0.3)
    ref = F.relu(F.relu(x.float()) @ w + 0.1)
    assert torch.allclose(ops.dpt_head_out(x.to(dev), w.to(dev), 0.1).cpu(), ref, atol=1e-4)
    d = rnd((3, 24, 36), 23).abs()
    d[2] = 0.7                                                    # a constant image: the 1e-6 floor, all zeros
    y = d - d.amin(dim=(1, 2), keepdim=True)
    y = y / y.amax(dim=(1, 2), keepdim=True).clamp_min(1e-6)
    ref = y.reshape(3, 1, 8, 3, 12, 3).permute(0, 1, 3, 5, 2, 4).reshape(3, 9, 8, 12)
    assert torch.allclose(ops.depth_normalize_unshuffle(d.to(dev), 3).cpu(), ref, atol=1e-6)


def _golden():
    from hi3d_hip import synth
    fx = torch.load(os.path.join(GOLD, "dpt_hybrid_64x96.pt"), weights_only=False)
    P = fx["key_prefix"]
    sd = synth.synth_state_dict({P + k: s for k, s in fx["shapes"].items()}, fx["weight_seed"])
    return fx, P, synth.damp_residual_tails(sd, fx["damp"])


def test_dpt_hybrid_matches_reference_midas(dev):
    from hi3d_hip.runtime_dpt import DPTHybridRuntime, dpt_hybrid_shapes
    fx, P, sd = _golden()
    assert dpt_hybrid_shapes() == fx["shapes"]
    rt = DPTHybridRuntime(sd, P, dev)
    d, layers = rt.forward_nhwc(fx["x"].permute(0, 2, 3, 1).contiguous().to(dev), return_layers=True)
    # Tolerances.  Every activation is stored in bf16 and the first decoder input sits behind ~25, the third behind ~50
    # convolution + GroupNorm layers plus 9 transformer blocks.  Rounding the SAME tensors to bf16 in the fp32 CPU
    # oracle costs rel 1.3e-2 / 1.6e-2 at layer_1 / layer_2 and 4.3e-2 (cos 0.9993) at the end of the backbone
    # (measured, round 2; the fixture damps the residual-branch tails, hi3d_hip.synth.RESIDUAL_TAILS -- undamped random
    # weights amplify a perturbation ~3x per stage and make the comparison meaningless).  What the bounds allow is the
    # storage format, not the kernels (each tested to <= 6e-3 above).
    stats = []
    for i, (a, b) in enumerate(zip(layers, fx["layers"])):
        a = a.permute(0, 3, 1, 2)
        assert a.shape == b.shape
        stats.append((f"layer_{i + 1}", rel(a, b), cos(a, b)))
    assert d.shape == fx["output"].shape and d.dtype == torch.float32
    stats.append(("depth", rel(d, fx["output"]), cos(d, fx["output"])))
    print(" | ".join(f"{n}: rel {r:.4f} cos {c:.5f}" for n, r, c in stats))
    for n, r, c in stats[:2]:
        assert r < 4e-2 and c > 0.999, (n, r, c)
    for n, r, c in stats[2:]:
        assert r < 5e-2 and c > 0.999, (n, r, c)                  # measured 2.5e-2 .. 3.0e-2, cos >= 0.9995


def test_depth_embedder_matches_oracle(dev):
    """The mirror class end to end (state_dict names of the reference tree, 4-D and 5-D inputs, use_3d) vs the CPU oracle
    of DepthEmbedder.forward; 16 frames of 256 x 320 -> MiDaS at 96 x 96... 96 x 96 / 96 x 96 (int(256 / 2.6666 / 32) * 32)."""
    import hi3d_oracle as O
    from vtdm.encoders import DepthEmbedder
    fx, P, sd = _golden()
    emb = DepthEmbedder()
    assert {P + k for k in fx["shapes"]} == set(emb.state_dict().keys())
    emb.load_state_dict(sd)
    x = torch.rand((16, 3, 256, 320), generator=torch.Generator().manual_seed(9)) * 2 - 1
    out = emb(x.to(dev)).cpu()
    ref = O.depth_embedder(sd, x, prefix=P)
    print(f"depth embedder: max |diff| {(out - ref).abs().max():.4f} mean |diff| {(out - ref).abs().mean():.5f} cos {cos(out, ref):.6f}")
    assert out.shape == ref.shape == (16, 9, 32, 40)
    # values live in [0, 1]; bf16 storage through the whole network (see above): a few per cent at the worst pixel
    assert (out - ref).abs().max() < 6e-2 and (out - ref).abs().mean() < 1e-2 and cos(out, ref) > 0.999   # measured 3.8e-2 / 5.0e-3 / 0.99979
    emb3 = DepthEmbedder(use_3d=True)
    emb3.load_state_dict(sd)
    out3 = emb3(x.reshape(1, 16, 3, 256, 320).permute(0, 2, 1, 3, 4).to(dev)).cpu()
    assert out3.shape == (1, 9, 16, 32, 40) and torch.equal(out3[0].permute(1, 0, 2, 3), out)
    with pytest.raises(ValueError):
        emb(x[:5].to(dev))


def test_v02_conditioner_end_to_end(dev):
    """The stage-2 conditioner exactly as pipeline_i2v_eval_v02.py:104-113 drives it: VideoLDM.add_custom_cond builds the
    batch from the clip, GeneralConditioner (instantiated from configs/inference-v02.yaml; CLIP tower reduced) turns it
    into c / uc with force_uc_zero_embeddings -- crossattn [1,1,1024] (CLIP image token), vector [1,512] (elevation ||
    cond_aug embeddings), concat [16, 9 + 4, h, w] (depth unshuffle || conditioning-frame latents); no embedder is a
    placeholder any more.  The depth channels are checked against the oracle on the batch's own (noised) frames."""
    import yaml
    import hi3d_oracle as O
    from conftest import shrink_conditioner
    from hi3d_hip import synth
    from sgm.modules.encoders.modules import _Unavailable
    from sgm.util import instantiate_from_config
    from vtdm.vtdm_gen_stage2_degradeImage import VideoLDM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    y = shrink_conditioner(yaml.safe_load(open(os.path.join(root, "hi3d-official_amd", "configs", "inference-v02.yaml"))))
    cond = instantiate_from_config(y["model"]["params"]["conditioner_config"])
    assert not any(isinstance(e, _Unavailable) for e in cond.embedders)
    synth.fill_module_(cond, seed=3)
    depth = [e for e in cond.embedders if type(e).__name__ == "DepthEmbedder"][0]
    dsd = synth.damp_residual_tails({k: v.clone().float() for k, v in depth.state_dict().items()}, 0.25)
    depth.load_state_dict(dsd)
    cond = cond.to(dev)
    T, H, W = 16, 256, 256
    video = (torch.rand((1, 3, T, H, W), generator=torch.Generator().manual_seed(12)) * 2 - 1).to(dev)
    stub = type("M", (), {"num_samples": T})()
    torch.manual_seed(5)                        # add_custom_cond draws the cond_aug noise from the global device generator
    batch = VideoLDM.add_custom_cond(stub, {"video": video, "elevation": torch.tensor([10], device=dev)}, infer=True)
    c, uc = cond.get_unconditional_conditioning(batch, force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"])
    assert c["crossattn"].shape == (1, 1, 1024) and c["vector"].shape == (1, 512)
    assert c["concat"].shape == (T, 13, H // 8, W // 8) == uc["concat"].shape
    assert float(uc["concat"].abs().max()) == 0.0 and float(uc["crossattn"].abs().max()) == 0.0
    assert torch.equal(uc["vector"], c["vector"])
    ref = O.depth_embedder(dsd, batch["cond_frames"].float().cpu(), prefix="model.model.")
    got = c["concat"][:, :9].float().cpu()
    print(f"v02 concat depth channels: max |diff| {(got - ref).abs().max():.4f} cos {cos(got, ref):.6f}")
    # (other weights than the fixture's, and the min-max normalisation divides the error by this clip's depth range; over
    # different noise draws the cosine was seen between 0.99897 and 0.99920 -- the draw used to depend on the tests that ran before)
    assert (got - ref).abs().max() < 1e-1 and (got - ref).abs().mean() < 1e-2 and cos(got, ref) > 0.998
    assert torch.isfinite(c["concat"]).all() and float(c["concat"][:, 9:].abs().max()) > 0


def test_depth_embedder_at_size(dev):
    """The shipped shape: 16 conditioning frames of 1024 x 1024 -> MiDaS at 384 x 384 (int(1024 / 2.6666 / 32) * 32) ->
    [16, 9, 128, 128].  The CPU oracle needs minutes at this size, so the checks are properties: range and per-frame
    min / max of the normalisation, frame independence (a frame's result does not depend on its batch neighbours),
    and agreement with the same frames run as a smaller batch."""
    import time
    from hi3d_hip import synth
    from vtdm.encoders import DepthEmbedder
    _, P, sd = _golden()
    emb = DepthEmbedder()
    emb.load_state_dict(sd)
    g = torch.Generator().manual_seed(31)
    base = F.interpolate(torch.rand((4, 3, 32, 32), generator=g), (1024, 1024), mode="bilinear") * 2 - 1   # smooth images
    x = base[[0, 1, 2, 3, 0, 1, 2, 3, 3, 2, 1, 0, 0, 0, 1, 1]].to(dev)
    out = emb(x)
    torch.cuda.synchronize()
    t0 = time.time()
    out = emb(x)
    torch.cuda.synchronize()
    print(f"DepthEmbedder 16 x 1024^2: {(time.time() - t0) * 1e3:.1f} ms")
    assert out.shape == (16, 9, 128, 128) and out.dtype == torch.float32 and torch.isfinite(out).all()
    assert float(out.min()) == 0.0 and float(out.max()) == 1.0
    assert all(float(out[i].min()) == 0.0 and float(out[i].max()) == 1.0 for i in range(16))
    for i, j in ((0, 4), (0, 11), (0, 12), (1, 5), (1, 15), (2, 9), (3, 8)):    # same frame, other batch position
        assert (out[i] - out[j]).abs().max() < 5e-2 and (out[i] - out[j]).abs().mean() < 2e-3
    # (bitwise on a healthy run; the tripwire for the attention kernel's ragged-tile incident of round 2 is
    #  test_attention_d64_ragged_repeatable, this test only bounds the effect)
    rt = emb.model.runtime(dev)
    four = rt.depth_embed(x[:4])              # (another batch size picks other GEMM tiles: same values to rounding, not bitwise)
    assert (four - out[:4]).abs().max() < 3e-2 and cos(four, out[:4]) > 0.9995
