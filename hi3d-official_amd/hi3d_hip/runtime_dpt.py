"""MiDaS DPT-hybrid depth network on the gfx950 kernels: the stage-2 depth conditioner (once per clip).

Reference: `DepthEmbedder` (vtdm/encoders.py:15-53) runs `MiDaSInference("dpt_hybrid")` = `DPTDepthModel(backbone=
"vitb_rn50_384", non_negative=True)` (annotator/midas/api.py:92-101, dpt_depth.py:21-106): a BiT ResNetV2-50 stem + two
stages whose outputs feed the decoder directly, a third stage projected to 768-wide tokens, ViT-B/16 (12 pre-LN blocks,
hooks after blocks 8 and 11), readout projection + reassemble convolutions (vit.py:357-482), four RefineNet-style fusion
blocks (blocks.py:261-390) and a three-convolution head.  Parameters keep THEIR names (`pretrained.model.*` -- timm's
`vit_base_resnet50_384` names --, `pretrained.act_postprocess*`, `scratch.*`), so `ckpts/dpt_hybrid_384.pt` and the
`conditioner.embedders.*.model.model.*` entries of a Hi3D checkpoint load unchanged.

Channels-last bf16 throughout, everything through the C ABI: every 1x1 / 3x3 convolution, the ViT linears and the
readout are the MFMA GEMM (3x3: implicit GEMM, stride 2 and TF-'SAME' as `pad_br_only`), GroupNorm(32) is the UNet's
kernel, attention the d = 64 flash kernel; the weight standardisation of the BiT convolutions (timm StdConv2dSame, eps
1e-8) is a constant of the weights and is folded in at pack time in fp32.  The stem convolution, the max pool, the
stride-2 gathers, bilinear resampling and the one-channel output convolution are the small kernels of csrc/depth.hip.
"""

import torch

from . import ops
from .pack import _bf16, pack_conv1x1, pack_conv3x3

STAGES = ((3, 256), (4, 512), (9, 1024))      # BiT ResNetV2 (3, 4, 9): blocks, output channels per stage
WIDTH, HEADS, DEPTH, GRID0, FEAT = 768, 12, 12, 24, 256


def dpt_hybrid_shapes(prefix=""):
    """{state_dict key: shape} of DPTDepthModel(backbone='vitb_rn50_384') (366 entries)."""
    s = {}
    P = prefix + "pretrained.model."
    s[P + "cls_token"] = (1, 1, WIDTH)
    s[P + "pos_embed"] = (1, 1 + GRID0 * GRID0, WIDTH)
    B = P + "patch_embed.backbone."
    s[B + "stem.conv.weight"] = (64, 3, 7, 7)
    s[B + "stem.norm.weight"] = (64,); s[B + "stem.norm.bias"] = (64,)
    cin = 64
    for si, (depth, cout) in enumerate(STAGES):
        mid = cout // 4
        for b in range(depth):
            q = f"{B}stages.{si}.blocks.{b}."
            if b == 0:
                s[q + "downsample.conv.weight"] = (cout, cin, 1, 1)
                s[q + "downsample.norm.weight"] = (cout,); s[q + "downsample.norm.bias"] = (cout,)
            s[q + "conv1.weight"] = (mid, cin if b == 0 else cout, 1, 1)
            s[q + "conv2.weight"] = (mid, mid, 3, 3)
            s[q + "conv3.weight"] = (cout, mid, 1, 1)
            for n, c in (("norm1", mid), ("norm2", mid), ("norm3", cout)):
                s[q + n + ".weight"] = (c,); s[q + n + ".bias"] = (c,)
        cin = cout
    s[P + "patch_embed.proj.weight"] = (WIDTH, 1024, 1, 1); s[P + "patch_embed.proj.bias"] = (WIDTH,)
    for i in range(DEPTH):
        q = f"{P}blocks.{i}."
        for n in ("norm1", "norm2"):
            s[q + n + ".weight"] = (WIDTH,); s[q + n + ".bias"] = (WIDTH,)
        s[q + "attn.qkv.weight"] = (3 * WIDTH, WIDTH); s[q + "attn.qkv.bias"] = (3 * WIDTH,)
        s[q + "attn.proj.weight"] = (WIDTH, WIDTH); s[q + "attn.proj.bias"] = (WIDTH,)
        s[q + "mlp.fc1.weight"] = (4 * WIDTH, WIDTH); s[q + "mlp.fc1.bias"] = (4 * WIDTH,)
        s[q + "mlp.fc2.weight"] = (WIDTH, 4 * WIDTH); s[q + "mlp.fc2.bias"] = (WIDTH,)
    s[P + "norm.weight"] = (WIDTH,); s[P + "norm.bias"] = (WIDTH,)
    A = prefix + "pretrained.act_postprocess"
    for k in ("3", "4"):
        s[f"{A}{k}.0.project.0.weight"] = (WIDTH, 2 * WIDTH); s[f"{A}{k}.0.project.0.bias"] = (WIDTH,)
        s[f"{A}{k}.3.weight"] = (WIDTH, WIDTH, 1, 1); s[f"{A}{k}.3.bias"] = (WIDTH,)
    s[A + "4.4.weight"] = (WIDTH, WIDTH, 3, 3); s[A + "4.4.bias"] = (WIDTH,)
    S = prefix + "scratch."
    for i, c in enumerate((256, 512, WIDTH, WIDTH)):
        s[f"{S}layer{i + 1}_rn.weight"] = (FEAT, c, 3, 3)
    for i in range(1, 5):
        q = f"{S}refinenet{i}."
        s[q + "out_conv.weight"] = (FEAT, FEAT, 1, 1); s[q + "out_conv.bias"] = (FEAT,)
        for u in (1, 2):
            for c in (1, 2):
                s[f"{q}resConfUnit{u}.conv{c}.weight"] = (FEAT, FEAT, 3, 3); s[f"{q}resConfUnit{u}.conv{c}.bias"] = (FEAT,)
    s[S + "output_conv.0.weight"] = (FEAT // 2, FEAT, 3, 3); s[S + "output_conv.0.bias"] = (FEAT // 2,)
    s[S + "output_conv.2.weight"] = (32, FEAT // 2, 3, 3); s[S + "output_conv.2.bias"] = (32,)
    s[S + "output_conv.4.weight"] = (1, 32, 1, 1); s[S + "output_conv.4.bias"] = (1,)
    return s


def _standardise(w):
    """timm StdConv2dSame: (w - mean) / sqrt(var + 1e-8) per output channel, biased variance, in fp32."""
    m = w.mean(dim=(1, 2, 3), keepdim=True)
    v = w.var(dim=(1, 2, 3), keepdim=True, unbiased=False)
    return (w - m) / torch.sqrt(v + 1e-8)


class DPTHybridRuntime:
    def __init__(self, sd, prefix, device):
        self.dev = torch.device(device)
        g = lambda k: sd[prefix + k].detach().to(device=self.dev, dtype=torch.float32)
        f = lambda k: g(k).contiguous()
        W = {}
        B = "pretrained.model.patch_embed.backbone."
        W["stem.w"] = _standardise(g(B + "stem.conv.weight")).permute(2, 3, 1, 0).contiguous()       # [7, 7, 3, 64] fp32
        W["stem.g"], W["stem.b"] = f(B + "stem.norm.weight"), f(B + "stem.norm.bias")
        for si, (depth, cout) in enumerate(STAGES):
            for b in range(depth):
                q, r = f"{B}stages.{si}.blocks.{b}.", f"s{si}.{b}."
                if b == 0:
                    W[r + "ds.w"] = pack_conv1x1(_standardise(g(q + "downsample.conv.weight")))
                    W[r + "ds.g"], W[r + "ds.b"] = f(q + "downsample.norm.weight"), f(q + "downsample.norm.bias")
                W[r + "c1.w"] = pack_conv1x1(_standardise(g(q + "conv1.weight")))
                W[r + "c2.w"] = pack_conv3x3(_standardise(g(q + "conv2.weight")))
                W[r + "c3.w"] = pack_conv1x1(_standardise(g(q + "conv3.weight")))
                for n in ("1", "2", "3"):
                    W[r + f"n{n}.g"], W[r + f"n{n}.b"] = f(q + f"norm{n}.weight"), f(q + f"norm{n}.bias")
        P = "pretrained.model."
        W["proj.w"], W["proj.b"] = pack_conv1x1(g(P + "patch_embed.proj.weight")), f(P + "patch_embed.proj.bias")
        self.cls, self.pos = g(P + "cls_token").reshape(WIDTH), g(P + "pos_embed")[0]
        self._pos_cache = {}
        for i in range(DEPTH):
            q, r = f"{P}blocks.{i}.", f"b{i}."
            for n in ("1", "2"):
                W[r + f"ln{n}.g"], W[r + f"ln{n}.b"] = f(q + f"norm{n}.weight"), f(q + f"norm{n}.bias")
            W[r + "qkv.w"], W[r + "qkv.b"] = _bf16(g(q + "attn.qkv.weight")), f(q + "attn.qkv.bias")
            W[r + "o.w"], W[r + "o.b"] = _bf16(g(q + "attn.proj.weight")), f(q + "attn.proj.bias")
            W[r + "fc1.w"], W[r + "fc1.b"] = _bf16(g(q + "mlp.fc1.weight")), f(q + "mlp.fc1.bias")
            W[r + "fc2.w"], W[r + "fc2.b"] = _bf16(g(q + "mlp.fc2.weight")), f(q + "mlp.fc2.bias")
        for k in ("3", "4"):
            q = f"pretrained.act_postprocess{k}."
            w = g(q + "0.project.0.weight")                                  # [768, 1536]: columns [token | class token]
            W[f"ro{k}.x.w"], W[f"ro{k}.c.w"], W[f"ro{k}.b"] = _bf16(w[:, :WIDTH]), _bf16(w[:, WIDTH:]), f(q + "0.project.0.bias")
            W[f"ro{k}.p.w"], W[f"ro{k}.p.b"] = pack_conv1x1(g(q + "3.weight")), f(q + "3.bias")
        W["ro4.d.w"], W["ro4.d.b"] = pack_conv3x3(g("pretrained.act_postprocess4.4.weight")), f("pretrained.act_postprocess4.4.bias")
        for i in range(1, 5):
            W[f"rn{i}.w"] = pack_conv3x3(g(f"scratch.layer{i}_rn.weight"))
            q = f"scratch.refinenet{i}."
            W[f"rf{i}.o.w"], W[f"rf{i}.o.b"] = pack_conv1x1(g(q + "out_conv.weight")), f(q + "out_conv.bias")
            for u in (1, 2):
                for c in (1, 2):
                    W[f"rf{i}.u{u}.c{c}.w"] = pack_conv3x3(g(f"{q}resConfUnit{u}.conv{c}.weight"))
                    W[f"rf{i}.u{u}.c{c}.b"] = f(f"{q}resConfUnit{u}.conv{c}.bias")
        W["h0.w"], W["h0.b"] = pack_conv3x3(g("scratch.output_conv.0.weight")), f("scratch.output_conv.0.bias")
        W["h2.w"], W["h2.b"] = pack_conv3x3(g("scratch.output_conv.2.weight")), f("scratch.output_conv.2.bias")
        W["h4.w"] = g("scratch.output_conv.4.weight").reshape(32).contiguous()
        self.h4_b = float(g("scratch.output_conv.4.bias").item())
        self.W = W

    # ------------------------------------------------------------------
    def _pos_embed(self, gh, gw):
        """vit.py:107-123 `_resize_pos_embed`: the 24 x 24 grid of position embeddings, bilinearly resampled to the
        input's token grid (a constant of the input size: computed once per size, on the device, through the same kernel)."""
        key = (gh, gw)
        if key not in self._pos_cache:
            grid = self.pos[1:].reshape(1, GRID0, GRID0, WIDTH).contiguous()
            grid = grid if (gh, gw) == (GRID0, GRID0) else ops.resize_bilinear(grid, gh, gw, align_corners=False)
            self._pos_cache[key] = (_bf16(self.cls + self.pos[0]), _bf16(grid.reshape(gh * gw, WIDTH)))
        return self._pos_cache[key]

    def _gn(self, x, key, N, P, C, relu):
        y = ops.groupnorm_silu(x.reshape(N * P, C), self.W[key + ".g"], self.W[key + ".b"], N, P, C, 1e-5, silu=False)
        return ops.act_(y, "relu") if relu else y

    def _conv3(self, x, key, N, H, Wd, Cin, Cout, stride=1, same=False, bias=True, R1=None):
        """3x3 convolution of [N*H*Wd, Cin]; stride 2 with `same`: TF 'SAME' (pad after only), else padding 1."""
        Ho, Wo = (H, Wd) if stride == 1 else ((H // 2, Wd // 2) if same else ((H - 1) // 2 + 1, (Wd - 1) // 2 + 1))
        return ops.gemm(x, self.W[key + ".w"], M=N * Ho * Wo, N=Cout, K=9 * Cin, bias=self.W[key + ".b"] if bias else None, R1=R1,
                        conv3x3=dict(Hin=H, Win=Wd, Cin=Cin, Hout=Ho, Wout=Wo, stride=stride, up2x=0,
                                     pad_br_only=1 if (same and stride == 2) else 0))

    def _stage(self, si, x, N, H, Wd, cin):
        depth, cout = STAGES[si]
        mid, stride = cout // 4, (1 if si == 0 else 2)
        for b in range(depth):
            r, st = f"s{si}.{b}.", (stride if b == 0 else 1)
            Ho, Wo = H // st, Wd // st
            if b == 0:
                xs = x if st == 1 else ops.pool2(x.reshape(N, H, Wd, cin), "pick")
                sc = ops.gemm(xs.reshape(N * Ho * Wo, cin), self.W[r + "ds.w"], M=N * Ho * Wo, N=cout, K=cin)
                sc = self._gn(sc, r + "ds", N, Ho * Wo, cout, False)
            else:
                sc = x
            h = ops.gemm(x.reshape(N * H * Wd, cin), self.W[r + "c1.w"], M=N * H * Wd, N=mid, K=cin)
            h = self._gn(h, r + "n1", N, H * Wd, mid, True)
            h = self._conv3(h, r + "c2", N, H, Wd, mid, mid, stride=st, same=True, bias=False)
            h = self._gn(h, r + "n2", N, Ho * Wo, mid, True)
            h = ops.gemm(h, self.W[r + "c3.w"], M=N * Ho * Wo, N=cout, K=mid)
            h = self._gn(h, r + "n3", N, Ho * Wo, cout, False)
            x = ops.add_act(h, sc.reshape(h.shape), "relu")
            H, Wd, cin = Ho, Wo, cout
        return x, H, Wd

    def _readout(self, t, k, N, S, gh, gw):
        """ProjectReadout + Transpose / Unflatten + the 1x1 convolution (vit.py:33-45,446-457): channels-last, the token
        order IS the pixel order, so nothing is transposed; the class-token half of the projection is a per-image bias."""
        W, T = self.W, gh * gw
        cls = t.reshape(N, S, WIDTH)[:, 0].contiguous()
        cvec = ops.gemm(cls, W[f"ro{k}.c.w"], M=N, N=WIDTH, K=WIDTH, bias=W[f"ro{k}.b"], out_fp32=True)
        f = torch.empty((N * T, WIDTH), device=self.dev, dtype=torch.bfloat16)
        for b in range(N):
            ops.gemm(t[b * S + 1:(b + 1) * S], W[f"ro{k}.x.w"], M=T, N=WIDTH, K=WIDTH, bias=cvec[b], out=f[b * T:(b + 1) * T])
        ops.act_(f, "gelu")
        return ops.gemm(f, W[f"ro{k}.p.w"], M=N * T, N=WIDTH, K=WIDTH, bias=W[f"ro{k}.p.b"])

    def _rcu(self, key, u, N, H, Wd):
        """ResidualConvUnit_custom (blocks.py:300-323): relu -> conv -> relu -> conv, + input."""
        o = self._conv3(ops.add_act(u, None, "relu"), key + ".c1", N, H, Wd, FEAT, FEAT)
        return self._conv3(ops.act_(o, "relu"), key + ".c2", N, H, Wd, FEAT, FEAT, R1=u)

    def _fusion(self, i, a, b, N, H, Wd):
        """FeatureFusionBlock_custom (blocks.py:368-390) at H x Wd -> [N * 2H * 2Wd, 256]."""
        o = a if b is None else ops.add_act(a, self._rcu(f"rf{i}.u1", b, N, H, Wd), "identity")
        o = self._rcu(f"rf{i}.u2", o, N, H, Wd)
        o = ops.resize_bilinear(o.reshape(N, H, Wd, FEAT), 2 * H, 2 * Wd, align_corners=True)
        return ops.gemm(o.reshape(N * 4 * H * Wd, FEAT), self.W[f"rf{i}.o.w"], M=N * 4 * H * Wd, N=FEAT, K=FEAT, bias=self.W[f"rf{i}.o.b"])

    @torch.no_grad()
    def forward_nhwc(self, x, return_layers=False):
        """x: fp32 [N, H, W, 3] channels-last on the device, H and W multiples of 32 -> inverse depth fp32 [N, H, W]."""
        W = self.W
        N, H, Wd, _ = x.shape
        if H % 32 or Wd % 32 or x.shape[-1] != 3:
            raise ops._l.Hi3dError(f"DPT-hybrid input must be [N, H, W, 3] with H, W multiples of 32, got {tuple(x.shape)}")
        with torch.cuda.device(self.dev):
            h = ops.dpt_stem_conv(x.contiguous(), W["stem.w"])                       # [N, H/2, W/2, 64]
            h = self._gn(h, "stem", N, (H // 2) * (Wd // 2), 64, True)
            h = ops.pool2(h.reshape(N, H // 2, Wd // 2, 64), "max3")                 # [N, H/4, W/4, 64]
            l1, H1, W1 = self._stage(0, h, N, H // 4, Wd // 4, 64)                    # 256 ch, H/4
            l2, H2, W2 = self._stage(1, l1, N, H1, W1, 256)                           # 512 ch, H/8
            l3, gh, gw = self._stage(2, l2, N, H2, W2, 512)                           # 1024 ch, H/16
            # ---- tokens (vit.py:126-160 forward_flex)
            T, S = gh * gw, gh * gw + 1
            cls_pos, pos = self._pos_embed(gh, gw)
            t = torch.empty((N * S, WIDTH), device=self.dev, dtype=torch.bfloat16)
            for b in range(N):
                t[b * S] = cls_pos
                ops.gemm(l3[b * T:(b + 1) * T], W["proj.w"], M=T, N=WIDTH, K=1024, bias=W["proj.b"], R1=pos, out=t[b * S + 1:(b + 1) * S])
            hooks = {}
            for i in range(DEPTH):
                r = f"b{i}."
                n = ops.layernorm(t, W[r + "ln1.g"], W[r + "ln1.b"], N * S, WIDTH, eps=1e-6)
                qkv = ops.gemm(n, W[r + "qkv.w"], M=N * S, N=3 * WIDTH, K=WIDTH, bias=W[r + "qkv.b"])
                a = ops.self_attention_fused_qkv(qkv, N, S, HEADS, scale=(WIDTH // HEADS) ** -0.5)
                t = ops.gemm(a, W[r + "o.w"], M=N * S, N=WIDTH, K=WIDTH, bias=W[r + "o.b"], R1=t)
                n = ops.layernorm(t, W[r + "ln2.g"], W[r + "ln2.b"], N * S, WIDTH, eps=1e-6)
                m = ops.gemm(n, W[r + "fc1.w"], M=N * S, N=4 * WIDTH, K=WIDTH, bias=W[r + "fc1.b"])
                ops.act_(m, "gelu")
                t = ops.gemm(m, W[r + "fc2.w"], M=N * S, N=WIDTH, K=4 * WIDTH, bias=W[r + "fc2.b"], R1=t)
                if i in (8, 11):
                    hooks[i] = t
            # ---- reassemble
            r3 = self._readout(hooks[8], "3", N, S, gh, gw)                           # [N*gh*gw, 768]
            r4 = self._readout(hooks[11], "4", N, S, gh, gw)
            r4 = self._conv3(r4, "ro4.d", N, gh, gw, WIDTH, WIDTH, stride=2)          # padding 1: ceil(g / 2)
            H4, W4 = (gh - 1) // 2 + 1, (gw - 1) // 2 + 1
            layers = [(l1, H1, W1, 256), (l2, H2, W2, 512), (r3, gh, gw, WIDTH), (r4, H4, W4, WIDTH)]
            rn = [self._conv3(x_, f"rn{i + 1}", N, h_, w_, c_, FEAT, bias=False) for i, (x_, h_, w_, c_) in enumerate(layers)]
            # ---- fusion decoder (dpt_depth.py:71-80)
            p = self._fusion(4, rn[3], None, N, H4, W4)
            if (2 * H4, 2 * W4) != (gh, gw):
                raise ops._l.Hi3dError("DPT-hybrid: token grid must be even in both directions (H, W multiples of 32)")
            p = self._fusion(3, p, rn[2], N, gh, gw)
            p = self._fusion(2, p, rn[1], N, H2, W2)
            p = self._fusion(1, p, rn[0], N, H1, W1)                                  # [N * (H/2) * (W/2), 256]
            # ---- head (dpt_depth.py:88-104)
            o = self._conv3(p, "h0", N, H // 2, Wd // 2, FEAT, FEAT // 2)
            o = ops.resize_bilinear(o.reshape(N, H // 2, Wd // 2, FEAT // 2), H, Wd, align_corners=True)
            o = self._conv3(o.reshape(N * H * Wd, FEAT // 2), "h2", N, H, Wd, FEAT // 2, 32)
            d = ops.dpt_head_out(o, W["h4.w"], self.h4_b).reshape(N, H, Wd)
        if return_layers:
            return d, [(x_.reshape(N, h_, w_, c_)) for x_, h_, w_, c_ in layers]
        return d

    @torch.no_grad()
    def depth_embed(self, x, shuffle_size=3, scale_factor=2.6666):
        """DepthEmbedder.forward for one batch of frames (vtdm/encoders.py:36-50): x fp32 [N, 3, H, W] in [-1, 1] ->
        fp32 [N, shuffle_size^2, H/8, W/8]."""
        N, _, H, Wd = x.shape
        sH, sW = int(H / scale_factor / 32) * 32, int(Wd / scale_factor / 32) * 32
        if sH < 32 or sW < 32:
            raise ops._l.Hi3dError(f"DepthEmbedder: frames of {H} x {Wd} are too small for MiDaS at scale 1/{scale_factor}")
        with torch.cuda.device(self.dev):
            xh = x.to(self.dev, torch.float32).permute(0, 2, 3, 1).contiguous()       # channels-last (a re-layout)
            y = ops.resize_bilinear(xh, sH, sW, align_corners=False)
            d = self.forward_nhwc(y)
            Hs, Ws = H // 8 * shuffle_size, Wd // 8 * shuffle_size
            d = ops.resize_bilinear(d.reshape(N, sH, sW, 1), Hs, Ws, align_corners=False).reshape(N, Hs, Ws)
            return ops.depth_normalize_unshuffle(d, shuffle_size)
