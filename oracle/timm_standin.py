"""TEST INFRASTRUCTURE ONLY -- a stand-in for the one `timm` model the reference's MiDaS code asks for.

The reference's depth conditioner (vtdm/encoders.py:15-53) runs MiDaS DPT-hybrid, whose code is vendored under
/root/reference/annotator/midas -- except for the backbone, which it obtains with
`timm.create_model('vit_base_resnet50_384')` (annotator/midas/vit.py:499).  timm (pinned by the reference's
requirements as `timm==0.6.12`-era; the dpt_hybrid_384.pt checkpoint carries its parameter names) is not installed in the
build container and there is no network.  This module restates that model's PUBLISHED architecture ("vit_base_r50_s16_384":
Kolesnikov et al. BiT ResNetV2-50 stem + stages (3, 4, 9) with weight-standardised "SAME"-padded convolutions and
GroupNorm(32), a 1x1 patch projection to 768, ViT-B: 12 pre-LN blocks, 12 heads, LayerNorm eps 1e-6) with timm's
state_dict key names, so that the reference's own forward_vit / DPT code can run on top of it when generating goldens
(oracle/gen_golden_dpt.py).  Pinning: the reference code above this module is the reference's own; this restatement of
the backbone is pinned against HuggingFace's independent implementation (transformers.DPTForDepthEstimation, hybrid
mode) by oracle/gen_golden_dpt.py --check-hf, NOT against timm itself: "backbone parity unpinned against timm".
"""
import math
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


def pad_same(x, k, s, value=0.0):
    """TF 'SAME' padding for kernel k, stride s (dilation 1): total = max((ceil(n/s)-1)*s + k - n, 0), the odd unit after."""
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
    if ph or pw:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=value)
    return x


class StdConv2dSame(nn.Conv2d):
    """Weight-standardised conv (per output channel, biased variance, eps 1e-8), 'SAME' padding: symmetric when the
    stride is 1, TF-style dynamic otherwise."""

    def __init__(self, cin, cout, k, stride=1, eps=1e-8):
        self.dynamic = stride != 1
        super().__init__(cin, cout, k, stride=stride, padding=0 if self.dynamic else (k - 1) // 2, bias=False)
        self.eps = eps

    def forward(self, x):
        if self.dynamic:
            x = pad_same(x, self.kernel_size[0], self.stride[0])
        w = self.weight
        m = w.mean(dim=(1, 2, 3), keepdim=True)
        v = w.var(dim=(1, 2, 3), keepdim=True, unbiased=False)
        return F.conv2d(x, (w - m) / torch.sqrt(v + self.eps), None, self.stride, self.padding)


class GroupNormAct(nn.GroupNorm):
    def __init__(self, ch, apply_act=True):
        super().__init__(32, ch, eps=1e-5)
        self.apply_act = apply_act

    def forward(self, x):
        x = F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        return F.relu(x) if self.apply_act else x


class MaxPool2dSame(nn.Module):
    def forward(self, x):
        return F.max_pool2d(pad_same(x, 3, 2, value=-float("inf")), 3, 2)


class DownsampleConv(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv = StdConv2dSame(cin, cout, 1, stride=stride)
        self.norm = GroupNormAct(cout, apply_act=False)

    def forward(self, x):
        return self.norm(self.conv(x))


class Bottleneck(nn.Module):
    """Non-pre-activation BiT bottleneck: conv-norm(+ReLU) x2, conv-norm, add, ReLU."""

    def __init__(self, cin, cout, stride, proj):
        super().__init__()
        mid = cout // 4
        self.downsample = DownsampleConv(cin, cout, stride) if proj else None
        self.conv1 = StdConv2dSame(cin, mid, 1)
        self.norm1 = GroupNormAct(mid)
        self.conv2 = StdConv2dSame(mid, mid, 3, stride=stride)
        self.norm2 = GroupNormAct(mid)
        self.conv3 = StdConv2dSame(mid, cout, 1)
        self.norm3 = GroupNormAct(cout, apply_act=False)

    def forward(self, x):
        sc = x if self.downsample is None else self.downsample(x)
        x = self.norm1(self.conv1(x))
        x = self.norm2(self.conv2(x))
        x = self.norm3(self.conv3(x))
        return F.relu(x + sc)


class ResNetStage(nn.Module):
    def __init__(self, cin, cout, stride, depth):
        super().__init__()
        self.blocks = nn.Sequential(*[Bottleneck(cin if i == 0 else cout, cout, stride if i == 0 else 1, i == 0) for i in range(depth)])

    def forward(self, x):
        return self.blocks(x)


class ResNetV2(nn.Module):
    def __init__(self, layers=(3, 4, 9), channels=(256, 512, 1024)):
        super().__init__()
        self.stem = nn.Sequential()
        self.stem.add_module("conv", StdConv2dSame(3, 64, 7, stride=2))
        self.stem.add_module("norm", GroupNormAct(64))
        self.stem.add_module("pool", MaxPool2dSame())
        prev, stages = 64, []
        for i, (d, c) in enumerate(zip(layers, channels)):
            stages.append(ResNetStage(prev, c, 1 if i == 0 else 2, d))
            prev = c
        self.stages = nn.Sequential(*stages)
        self.norm = nn.Identity()
        self.head = nn.Identity()

    def forward(self, x):
        return self.stages(self.stem(x))


class HybridEmbed(nn.Module):
    def __init__(self, backbone, feature_dim, embed_dim):
        super().__init__()
        self.backbone = backbone
        self.proj = nn.Conv2d(feature_dim, embed_dim, 1)

    def forward(self, x):
        return self.proj(self.backbone(x)).flatten(2).transpose(1, 2)


class Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        q, k, v = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        a = (q @ k.transpose(-2, -1)) * (C // self.num_heads) ** -0.5
        return self.proj((a.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, N, C))


class Mlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1, self.act, self.fc2 = nn.Linear(dim, 4 * dim), nn.GELU(), nn.Linear(4 * dim, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class VisionTransformer(nn.Module):
    def __init__(self, embed_dim=768, depth=12, heads=12, grid=24):
        super().__init__()
        self.patch_embed = HybridEmbed(ResNetV2(), 1024, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + grid * grid, embed_dim))
        self.pos_drop = nn.Identity()
        self.blocks = nn.Sequential(*[Block(embed_dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Identity()            # (num_classes head of the ImageNet model: DPT never calls it)
        self.dist_token = None


def create_model(name, pretrained=False, **kw):
    if name != "vit_base_resnet50_384" or pretrained:
        raise NotImplementedError(f"timm stand-in: only vit_base_resnet50_384 without pretrained weights, not {name}")
    return VisionTransformer()


def install():
    if "timm" not in sys.modules:
        m = types.ModuleType("timm")
        m.create_model = create_model
        m.__hi3d_standin__ = True
        sys.modules["timm"] = m
