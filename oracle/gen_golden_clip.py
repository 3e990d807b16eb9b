"""TEST INFRASTRUCTURE -- golden vectors for the conditioner's CLIP vision towers.

The reference builds them from open_clip (ViT-H-14) and OpenAI clip (ViT-L/14); neither package is in the build
container.  HuggingFace `transformers` is, and its CLIPVisionModelWithProjection is an independent implementation of the
same published architecture (transformers' convert_clip_original_pytorch_to_hf.py maps the names used below).  This
script draws a state_dict in the ORIGINAL key names with hi3d_hip.synth, loads it into the HF model through that
mapping, runs HF on CPU fp32, and stores input + config + weight seed + output.  Reduced widths (the architecture is
width-agnostic); one fixture has head_dim 80 as ViT-H/14, one head_dim 64 + QuickGELU as ViT-L/14.

usage: python oracle/gen_golden_clip.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd", "hi3d_hip"))
import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def clip_shapes(width, layers, patch, grid, out_dim, prefix="visual."):
    s = {prefix + "conv1.weight": (width, 3, patch, patch), prefix + "class_embedding": (width,),
         prefix + "positional_embedding": (1 + grid * grid, width), prefix + "proj": (width, out_dim)}
    for n in ("ln_pre", "ln_post"):
        s[prefix + n + ".weight"] = (width,); s[prefix + n + ".bias"] = (width,)
    for i in range(layers):
        p = prefix + f"transformer.resblocks.{i}."
        for n in ("ln_1", "ln_2"):
            s[p + n + ".weight"] = (width,); s[p + n + ".bias"] = (width,)
        s[p + "attn.in_proj_weight"] = (3 * width, width); s[p + "attn.in_proj_bias"] = (3 * width,)
        s[p + "attn.out_proj.weight"] = (width, width); s[p + "attn.out_proj.bias"] = (width,)
        s[p + "mlp.c_fc.weight"] = (4 * width, width); s[p + "mlp.c_fc.bias"] = (4 * width,)
        s[p + "mlp.c_proj.weight"] = (width, 4 * width); s[p + "mlp.c_proj.bias"] = (width,)
    return s


def to_hf(sd, width, layers, prefix="visual."):
    """original (open_clip / OpenAI) names -> transformers.CLIPVisionModelWithProjection names"""
    g = lambda k: sd[prefix + k]
    out = {"vision_model.embeddings.class_embedding": g("class_embedding"),
           "vision_model.embeddings.patch_embedding.weight": g("conv1.weight"),
           "vision_model.embeddings.position_embedding.weight": g("positional_embedding"),
           "vision_model.pre_layrnorm.weight": g("ln_pre.weight"), "vision_model.pre_layrnorm.bias": g("ln_pre.bias"),
           "vision_model.post_layernorm.weight": g("ln_post.weight"), "vision_model.post_layernorm.bias": g("ln_post.bias"),
           "visual_projection.weight": g("proj").t().contiguous()}
    for i in range(layers):
        p, h = f"transformer.resblocks.{i}.", f"vision_model.encoder.layers.{i}."
        wq, wk, wv = g(p + "attn.in_proj_weight").chunk(3, dim=0)
        bq, bk, bv = g(p + "attn.in_proj_bias").chunk(3, dim=0)
        for n, wt, bs in (("q_proj", wq, bq), ("k_proj", wk, bk), ("v_proj", wv, bv)):
            out[h + f"self_attn.{n}.weight"] = wt.contiguous(); out[h + f"self_attn.{n}.bias"] = bs.contiguous()
        out[h + "self_attn.out_proj.weight"] = g(p + "attn.out_proj.weight"); out[h + "self_attn.out_proj.bias"] = g(p + "attn.out_proj.bias")
        out[h + "layer_norm1.weight"] = g(p + "ln_1.weight"); out[h + "layer_norm1.bias"] = g(p + "ln_1.bias")
        out[h + "layer_norm2.weight"] = g(p + "ln_2.weight"); out[h + "layer_norm2.bias"] = g(p + "ln_2.bias")
        out[h + "mlp.fc1.weight"] = g(p + "mlp.c_fc.weight"); out[h + "mlp.fc1.bias"] = g(p + "mlp.c_fc.bias")
        out[h + "mlp.fc2.weight"] = g(p + "mlp.c_proj.weight"); out[h + "mlp.fc2.bias"] = g(p + "mlp.c_proj.bias")
    return out


def make(name, width, heads, layers, patch, grid, out_dim, act, seed, batch=2):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    cfg = dict(width=width, heads=heads, layers=layers, patch=patch, grid=grid, out_dim=out_dim, act=act)
    shapes = clip_shapes(width, layers, patch, grid, out_dim)
    sd = synth.synth_state_dict(shapes, seed)
    # the embedding tables are not "weights" by synth's shape rule (it scales 2-D tensors by fan-in): fine, any values do
    hf_cfg = CLIPVisionConfig(hidden_size=width, intermediate_size=4 * width, num_hidden_layers=layers, num_attention_heads=heads,
                              image_size=patch * grid, patch_size=patch, projection_dim=out_dim, hidden_act=act,
                              layer_norm_eps=1e-5, attention_dropout=0.0)
    m = CLIPVisionModelWithProjection(hf_cfg).eval()
    missing, unexpected = m.load_state_dict(to_hf(sd, width, layers), strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(seed + 100)
    img = torch.randn((batch, 3, patch * grid, patch * grid), generator=g)
    with torch.no_grad():
        out = m(pixel_values=img).image_embeds
    torch.save(dict(cfg=cfg, seed=seed, img=img, out=out.float(), source="transformers.CLIPVisionModelWithProjection "
                    + __import__("transformers").__version__), os.path.join(GOLD, name + ".pt"))
    print(name, tuple(out.shape), float(out.abs().max()))


if __name__ == "__main__":
    make("clip_vith_like", width=320, heads=4, layers=2, patch=14, grid=4, out_dim=96, act="gelu", seed=21)        # head_dim 80
    make("clip_vitl_like", width=128, heads=2, layers=2, patch=14, grid=5, out_dim=48, act="quick_gelu", seed=22)  # head_dim 64
