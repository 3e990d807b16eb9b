"""TEST INFRASTRUCTURE -- golden vectors for the depth conditioner's MiDaS DPT-hybrid network.

Runs the REFERENCE's vendored MiDaS code (/root/reference/annotator/midas/{dpt_depth,blocks,vit}.py: forward_vit hooks,
readout projection, reassemble convs, fusion blocks, depth head) on CPU fp32.  The one piece that code imports from
outside the tree -- timm's `vit_base_resnet50_384` backbone -- is supplied by oracle/timm_standin.py (a restatement of
the published BiT-ResNetV2-50 + ViT-B/16 hybrid with timm's parameter names; timm is not installed here).  With
--check-hf the same synthetic weights are loaded, through the name mapping of transformers'
convert_dpt_hybrid_to_pytorch.py, into HuggingFace's independent DPTForDepthEstimation(is_hybrid=True) -- its own BiT
backbone, ViT, neck and head -- and the two depth maps are compared: that pins the stand-in backbone (and everything
else) against a second implementation of the architecture.  Weights are drawn per key by hi3d_hip.synth (never stored).

usage: python oracle/gen_golden_dpt.py [--check-hf]
"""
import argparse
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd", "hi3d_hip"))
import synth  # noqa: E402
import timm_standin  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
DAMP = 0.25
PREFIX = "model.model."          # DepthEmbedder.model (MiDaSInference) .model (DPTDepthModel): vtdm/encoders.py:18, api.py:159


def build_reference(seed):
    timm_standin.install()
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from annotator.midas.dpt_depth import DPTDepthModel
    m = DPTDepthModel(path=None, backbone="vitb_rn50_384", non_negative=True).eval()
    synth.fill_module_(m, seed, prefix=PREFIX)
    m.load_state_dict(synth.damp_residual_tails(m.state_dict(), DAMP))      # (see hi3d_hip.synth.RESIDUAL_TAILS)
    return m


def to_hf(sd):
    """MiDaS / timm names -> transformers.DPTForDepthEstimation (hybrid) names."""
    out = {}
    P = "pretrained.model."
    out["dpt.embeddings.cls_token"] = sd[P + "cls_token"]
    out["dpt.embeddings.position_embeddings"] = sd[P + "pos_embed"]
    B, H = P + "patch_embed.backbone.", "dpt.embeddings.backbone.bit."
    out[H + "embedder.convolution.weight"] = sd[B + "stem.conv.weight"]
    for n in ("weight", "bias"):
        out[H + "embedder.norm." + n] = sd[B + "stem.norm." + n]
        out["dpt.embeddings.projection." + n] = sd[P + "patch_embed.proj." + n]
        out["dpt.layernorm." + n] = sd[P + "norm." + n]
    for k, v in sd.items():
        if k.startswith(B + "stages."):
            out[H + "encoder." + k[len(B):].replace(".blocks.", ".layers.")] = v
    for i in range(12):
        b, h = P + f"blocks.{i}.", f"dpt.encoder.layer.{i}."
        for n in ("weight", "bias"):
            q, k_, v = sd[b + "attn.qkv." + n].chunk(3, dim=0)
            out[h + "attention.attention.query." + n] = q.contiguous()
            out[h + "attention.attention.key." + n] = k_.contiguous()
            out[h + "attention.attention.value." + n] = v.contiguous()
            out[h + "attention.output.dense." + n] = sd[b + "attn.proj." + n]
            out[h + "intermediate.dense." + n] = sd[b + "mlp.fc1." + n]
            out[h + "output.dense." + n] = sd[b + "mlp.fc2." + n]
            out[h + "layernorm_before." + n] = sd[b + "norm1." + n]
            out[h + "layernorm_after." + n] = sd[b + "norm2." + n]
    for n in ("weight", "bias"):
        for s in (3, 4):
            out[f"neck.reassemble_stage.readout_projects.{s - 1}.0.{n}"] = sd[f"pretrained.act_postprocess{s}.0.project.0.{n}"]
            out[f"neck.reassemble_stage.layers.{s - 1}.projection.{n}"] = sd[f"pretrained.act_postprocess{s}.3.{n}"]
        out[f"neck.reassemble_stage.layers.3.resize.{n}"] = sd[f"pretrained.act_postprocess4.4.{n}"]
    for s in range(1, 5):
        out[f"neck.convs.{s - 1}.weight"] = sd[f"scratch.layer{s}_rn.weight"]
        r, f = f"scratch.refinenet{s}.", f"neck.fusion_stage.layers.{4 - s}."
        for n in ("weight", "bias"):
            out[f + "projection." + n] = sd[r + "out_conv." + n]
            for u in (1, 2):
                for c in (1, 2):
                    out[f + f"residual_layer{u}.convolution{c}.{n}"] = sd[r + f"resConfUnit{u}.conv{c}.{n}"]
    for i in (0, 2, 4):
        for n in ("weight", "bias"):
            out[f"head.head.{i}.{n}"] = sd[f"scratch.output_conv.{i}.{n}"]
    return out


def check_hf(ref_model, x, want):
    from transformers import DPTConfig, DPTForDepthEstimation
    cfg = DPTConfig(is_hybrid=True, neck_hidden_sizes=[256, 512, 768, 768], layer_norm_eps=1e-6)
    hf = DPTForDepthEstimation(cfg).eval()
    missing, unexpected = hf.load_state_dict(to_hf(ref_model.state_dict()), strict=False)
    missing = [k for k in missing if "auxiliary" not in k]
    if missing or unexpected:
        raise SystemExit(f"HF mapping incomplete: missing {missing[:8]} unexpected {unexpected[:8]}")
    with torch.no_grad():
        hf.dpt.embeddings.image_size = tuple(x.shape[-2:])        # (its size check; the position grid is resized either way)
        got = hf(pixel_values=x).predicted_depth
    d = (got - want).abs().max().item()
    print(f"HF DPTForDepthEstimation vs reference-on-stand-in: max |diff| {d:.3e} (output absmax {want.abs().max():.4f})")
    return d


def reference_depth_embedder(dpt):
    """The reference's own vtdm.encoders.DepthEmbedder / annotator.midas.api.MiDaSInference classes around `dpt`.
    Their modules import packages that are absent here and unused by forward(): cv2 and torchvision (MiDaS's numpy
    image transforms), clip, raft, softsplat (other embedders of the same file) -- inert stand-ins; the constructors
    (which read ckpts/dpt_hybrid_384.pt and call .cuda()) are bypassed, forward() is the reference's."""
    import types
    import ref_import
    ref_import.install()

    def stub(name, **attrs):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            m.__path__ = []
            sys.modules[name] = m
        return sys.modules[name]
    stub("cv2", INTER_CUBIC=2, INTER_AREA=3)
    stub("torchvision")
    stub("torchvision.transforms", Compose=lambda ts: ts)
    stub("torchvision.models")
    stub("torchvision.models.optical_flow", raft_large=None)
    stub("clip")
    stub("tools.softmax_splatting")
    stub("tools.softmax_splatting.softsplat", softsplat=None)
    from annotator.midas.api import MiDaSInference
    from vtdm.encoders import DepthEmbedder
    inf = MiDaSInference.__new__(MiDaSInference)
    torch.nn.Module.__init__(inf)
    inf.model = dpt
    emb = DepthEmbedder.__new__(DepthEmbedder)
    torch.nn.Module.__init__(emb)
    emb.model, emb.use_3d, emb.shuffle_size, emb.scale_factor = inf, False, 3, 2.6666
    return emb.eval()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-hf", action="store_true")
    ap.add_argument("--seed", type=int, default=5)      # (a seed whose depth map is mostly positive: 12 % of the pixels at the final ReLU's zero)
    a = ap.parse_args()
    if a.check_hf:                     # (before the timm stand-in enters sys.modules: transformers probes for a real timm)
        from transformers import DPTConfig, DPTForDepthEstimation  # noqa: F401
    t0 = time.time()
    m = build_reference(a.seed)
    g = torch.Generator().manual_seed(0)
    x = torch.rand((2, 3, 64, 96), generator=g) * 2 - 1            # cond_frames live in [-1, 1]
    import annotator.midas.vit as rvit
    with torch.no_grad():
        out = m(x)
        layers = rvit.forward_vit(m.pretrained, x)
    fx = dict(kind="dpt_hybrid", weight_seed=a.seed, damp=DAMP, key_prefix=PREFIX, x=x, output=out,
              layers=[t.clone() for t in layers], shapes={k: tuple(v.shape) for k, v in m.state_dict().items()})
    print(f"dpt_hybrid: out {tuple(out.shape)} absmax {out.abs().max():.4f} mean {out.mean():.4f} zeros {(out == 0).float().mean():.3f} "
          f"({time.time() - t0:.1f}s)")
    if a.check_hf:                      # (HF's hybrid reassemble assumes a square token grid: its own input)
        xs = torch.rand((1, 3, 64, 64), generator=g) * 2 - 1
        with torch.no_grad():
            fx["hf_maxdiff"] = check_hf(m, xs, m(xs))
    # DepthEmbedder.forward itself (resizes, min-max, unshuffle) through the reference class: 16 frames (t = 16 is
    # hard-wired there) of 96 x 128 -> MiDaS at 32 x 32 -> [16, 9, 12, 16]; the input is re-drawn from its seed by the tests
    emb = reference_depth_embedder(m)
    xe = torch.rand((16, 3, 96, 128), generator=torch.Generator().manual_seed(77)) * 2 - 1
    with torch.no_grad():
        ye = emb(xe.clone())
    fx["embedder_input_seed"], fx["embedder_input_shape"], fx["embedder_output"] = 77, tuple(xe.shape), ye
    print(f"DepthEmbedder (reference class): out {tuple(ye.shape)} min {ye.min():.3f} max {ye.max():.3f}")
    torch.save(fx, os.path.join(GOLD, "dpt_hybrid_64x96.pt"))


if __name__ == "__main__":
    main()
