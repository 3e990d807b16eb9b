"""TEST INFRASTRUCTURE -- CPU fp32 oracle for the Hi3D denoising hot path.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module; the product path (hi3d-official_amd/) never does and has no CPU fallback.

A functional restatement (plain torch fp32 on CPU, NCHW like the reference) of what the
reference's nn.Module tree computes on this path.  Weights come in as a flat
`state_dict`-style mapping using the REFERENCE'S key names, so the same checkpoint /
synthetic tensors drive the reference, this oracle and the HIP path.

Pinned (tests/test_oracle_cpu.py) against golden vectors produced by running the
reference's own classes in the build container (oracle/gen_golden.py ->
tests/golden/*.pt).  The reference itself has no golden vectors or tests for this path
(SURVEY.md section 4), so those generated fixtures are the pin.

All citations are relative to /root/reference.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# leaf pieces
# --------------------------------------------------------------------------------------
def sinusoid(t, dim, max_period=10000.0):
    """cos || sin embedding -- sgm/modules/diffusionmodules/util.py:207-231."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t.float()[:, None] * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _gn(sd, p, x, eps):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _mlp2(sd, p, x):
    """Linear-SiLU-Linear stored as p.0 / p.2 (video_model.py:151-182; video_attention.py:220-224)."""
    return _lin(sd, p + ".2", F.silu(_lin(sd, p + ".0", x)))


def _mha(q, k, v, heads):
    """softmax(q k^T / sqrt(d)) v with heads packed as '(h d)' -- attention.py:281-344."""
    b, n, c = q.shape
    d = c // heads
    split = lambda t: t.reshape(b, t.shape[1], heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(split(q), split(k), split(v))
    return o.transpose(1, 2).reshape(b, n, c)


def _attn(sd, p, x, ctx, heads):
    """CrossAttention.forward (attention.py:281-344): context=None means self-attention."""
    src = x if ctx is None else ctx
    o = _mha(F.linear(x, sd[p + ".to_q.weight"]), F.linear(src, sd[p + ".to_k.weight"]),
             F.linear(src, sd[p + ".to_v.weight"]), heads)
    return _lin(sd, p + ".to_out.0", o)


def _geglu_ff(sd, p, x):
    """FeedForward with GEGLU (attention.py:87-113): net.0.proj -> a * gelu(g) -> net.2."""
    a, g = _lin(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", a * F.gelu(g))


def _alpha(sd, p, image_only_indicator):
    """AlphaBlender 'learned_with_images' (util.py:341-357): 1 where the frame is
    image-only, sigmoid(mix_factor) elsewhere.  Returns [b, t]."""
    a = torch.sigmoid(sd[p + ".mix_factor"]).reshape(1, 1)
    return torch.where(image_only_indicator.bool(), torch.ones(1, 1), a)


# --------------------------------------------------------------------------------------
# UNet blocks
# --------------------------------------------------------------------------------------
def resblock(sd, p, x, emb, dims):
    """ResBlock._forward, non-updown / non-scale-shift branch (openaimodel.py:328-354).
    dims=2: x [N,C,H,W], emb [N,E].  dims=3 (time_stack, exchange_temb_dims):
    x [b,C,t,H,W], emb [b,t,E] and the embedding lands on [b,C,t,1,1]."""
    conv = F.conv2d if dims == 2 else F.conv3d
    pad = 1 if dims == 2 else (1, 0, 0)
    h = conv(F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5)), sd[p + ".in_layers.2.weight"],
             sd[p + ".in_layers.2.bias"], padding=pad)
    e = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    if dims == 2:
        e = e[:, :, None, None]
    else:
        e = e.transpose(1, 2)[:, :, :, None, None]          # b t c -> b c t 1 1
    h = h + e
    h = conv(F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)), sd[p + ".out_layers.3.weight"],
             sd[p + ".out_layers.3.bias"], padding=pad)
    if (p + ".skip_connection.weight") in sd:                # 1x1 conv when channels change (:314)
        x = conv(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def video_resblock(sd, p, x, emb, T, ioi):
    """VideoResBlock.forward (video_model.py:62-81): spatial ResBlock, then the temporal
    ResBlock on 'b c t h w', blended per frame by AlphaBlender('b t -> b 1 t 1 1')."""
    x = resblock(sd, p, x, emb, 2)
    n, c, hh, ww = x.shape
    b = n // T
    x5 = x.reshape(b, T, c, hh, ww).permute(0, 2, 1, 3, 4)
    xt = resblock(sd, p + ".time_stack", x5, emb.reshape(b, T, -1), 3)
    a = _alpha(sd, p + ".time_mixer", ioi)[:, None, :, None, None]
    out = a * x5 + (1.0 - a) * xt
    return out.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


def spatial_video_transformer(sd, p, x, ctx, T, ioi, heads, depth=1):
    """SpatialVideoTransformer.forward (video_attention.py:230-301) with
    use_linear=True, use_spatial_context=True, ff_in=True (configs/inference-v01.yaml:39-46)."""
    n, c, hh, ww = x.shape
    b, s = n // T, hh * ww
    x_in = x
    # time context = context of each clip's first frame, one copy per pixel (:249-253)
    tctx = ctx[::T].repeat_interleave(s, dim=0)
    h = _gn(sd, p + ".norm", x, 1e-6).reshape(n, c, s).transpose(1, 2)      # b (h w) c
    h = _lin(sd, p + ".proj_in", h)
    # frame-position embedding (:266-276) -- depends on T only
    pos = _mlp2(sd, p + ".time_pos_embed", sinusoid(torch.arange(T).repeat(b), c))[:, None, :]
    a = _alpha(sd, p + ".time_mixer", ioi).reshape(n, 1, 1)                  # 'b t -> (b t) 1 1'
    for d in range(depth):
        sp, tp = f"{p}.transformer_blocks.{d}", f"{p}.time_stack.{d}"
        # BasicTransformerBlock._forward (attention.py:551-572)
        h = h + _attn(sd, sp + ".attn1", _ln(sd, sp + ".norm1", h), None, heads)
        h = h + _attn(sd, sp + ".attn2", _ln(sd, sp + ".norm2", h), ctx, heads)
        h = h + _geglu_ff(sd, sp + ".ff", _ln(sd, sp + ".norm3", h))
        # VideoTransformerBlock._forward (video_attention.py:109-140) on '(b s) t c'
        m = (h + pos).reshape(b, T, s, c).permute(0, 2, 1, 3).reshape(b * s, T, c)
        m = m + _geglu_ff(sd, tp + ".ff_in", _ln(sd, tp + ".norm_in", m))
        m = m + _attn(sd, tp + ".attn1", _ln(sd, tp + ".norm1", m), None, heads)
        m = m + _attn(sd, tp + ".attn2", _ln(sd, tp + ".norm2", m), tctx, heads)
        m = m + _geglu_ff(sd, tp + ".ff", _ln(sd, tp + ".norm3", m))
        m = m.reshape(b, s, T, c).permute(0, 2, 1, 3).reshape(n, s, c)
        h = a * h + (1.0 - a) * m                                            # (:290-294)
    h = _lin(sd, p + ".proj_out", h)
    return h.transpose(1, 2).reshape(n, c, hh, ww) + x_in


def unet_layout(cfg):
    """Block plan of VideoUNet.__init__ (video_model.py:186-440) for resblock_updown=False:
    returns (input_blocks, middle, output_blocks) as lists of layer tags per block."""
    mc, mult, nres = cfg["model_channels"], list(cfg["channel_mult"]), cfg["num_res_blocks"]
    att = set(cfg["attention_resolutions"])
    inp, chans, ch, ds = [[("conv_in",)]], [mc], mc, 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            layers = [("res", ch, m * mc)]
            ch = m * mc
            if ds in att:
                layers.append(("attn", ch))
            inp.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    mid = [("res", ch, ch), ("attn", ch), ("res", ch, ch)]
    outp = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nres + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * m)]
            ch = mc * m
            if ds in att:
                layers.append(("attn", ch))
            if level and i == nres:
                layers.append(("up", ch))
                ds //= 2
            outp.append(layers)
    return inp, mid, outp


def video_unet(sd, cfg, x, timesteps, context, y, T, ioi, prefix=""):
    """VideoUNet.forward (video_model.py:442-501).  x [b*T, Cin, H, W]; timesteps [b*T];
    context [b or b*T, 1, ctx]; y [b or b*T, adm]; ioi [b, T]."""
    P = prefix
    nheads = lambda ch: ch // cfg["num_head_channels"]
    if y.shape[0] != x.shape[0]:                     # "fast implementation" repeat (:459-465)
        y = y.repeat_interleave(T, dim=0)
    if context.shape[0] != x.shape[0]:
        context = context.repeat_interleave(T, dim=0)
    emb = _mlp2(sd, P + "time_embed", sinusoid(timesteps, cfg["model_channels"]))
    emb = emb + _mlp2(sd, P + "label_emb.0", y)
    inp, mid, outp = unet_layout(cfg)

    def run(h, layers, base):
        for j, L in enumerate(layers):
            p = f"{base}.{j}"
            if L[0] == "conv_in":
                h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
            elif L[0] == "res":
                h = video_resblock(sd, p, h, emb, T, ioi)
            elif L[0] == "attn":
                h = spatial_video_transformer(sd, p, h, context, T, ioi, nheads(L[1]))
            elif L[0] == "down":                     # Downsample, stride-2 conv (openaimodel.py:192-199)
                h = F.conv2d(h, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=2, padding=1)
            elif L[0] == "up":                       # Upsample: nearest 2x + conv (openaimodel.py:154-156)
                h = F.conv2d(F.interpolate(h, scale_factor=2, mode="nearest"),
                             sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
        return h

    h, hs = x, []
    for i, layers in enumerate(inp):
        h = run(h, layers, f"{P}input_blocks.{i}")
        hs.append(h)
    h = run(h, mid, P + "middle_block")
    for i, layers in enumerate(outp):
        h = run(torch.cat([h, hs.pop()], dim=1), layers, f"{P}output_blocks.{i}")
    h = F.silu(_gn(sd, P + "out.0", h, 1e-5))
    return F.conv2d(h, sd[P + "out.2.weight"], sd[P + "out.2.bias"], padding=1)


# --------------------------------------------------------------------------------------
# sampling control
# --------------------------------------------------------------------------------------
def edm_sigmas(n, sigma_min=0.002, sigma_max=700.0, rho=7.0):
    """EDMDiscretization.get_sigmas + append_zero (discretizer.py:17-39)."""
    ramp = torch.linspace(0, 1, n)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return torch.cat([(hi + ramp * (lo - hi)) ** rho, torch.zeros(1)])


def denoise_cfg(sd, cfg, x, sigma, c, uc, T, scale, prefix="model.diffusion_model."):
    """One guided denoiser evaluation for a scalar sigma:
    guiders.py:88-99 (batch doubling uc||c) -> denoiser.py:23-39 with
    VScalingWithEDMcNoise (denoiser_scaling.py:51-59) -> wrappers.py:23-34 (cat concat)
    -> VideoUNet -> guiders.py:78-86 (per-frame linear CFG scale)."""
    xx = torch.cat([x, x])
    s = torch.full((xx.shape[0],), float(sigma))
    c_skip, c_out = 1.0 / (s ** 2 + 1.0), -s / (s ** 2 + 1.0) ** 0.5
    c_in, c_noise = 1.0 / (s ** 2 + 1.0) ** 0.5, 0.25 * s.log()
    bc = lambda v: v[:, None, None, None]
    net_in = torch.cat([xx * bc(c_in), torch.cat([uc["concat"], c["concat"]])], dim=1)
    ioi = torch.zeros(2, T)
    out = video_unet(sd, cfg, net_in, c_noise, torch.cat([uc["crossattn"], c["crossattn"]]),
                     torch.cat([uc["vector"], c["vector"]]), T, ioi, prefix)
    den = out * bc(c_out) + xx * bc(c_skip)
    d_u, d_c = den.chunk(2)
    return d_u + scale.reshape(-1, 1, 1, 1) * (d_c - d_u)


def euler_edm_sample(sd, cfg, x, c, uc, T, num_steps, max_scale, min_scale=1.0, sigma_max=700.0,
                     prefix="model.diffusion_model.", return_all=False):
    """EulerEDMSampler.__call__ with s_churn=0 (sampling.py:41-52,93-147,228-232;
    sampling_utils.py:34-35): x *= sqrt(1+s0^2); x += (s_{i+1}-s_i) * (x - D)/s_i."""
    sig = edm_sigmas(num_steps, sigma_max=sigma_max)
    scale = torch.linspace(min_scale, max_scale, T)
    x = x * torch.sqrt(1.0 + sig[0] ** 2)
    traj = []
    for i in range(num_steps):
        d = denoise_cfg(sd, cfg, x, sig[i], c, uc, T, scale, prefix)
        x = x + (sig[i + 1] - sig[i]) * (x - d) / sig[i]
        traj.append(x)
    return (x, traj) if return_all else x


# --------------------------------------------------------------------------------------
# first stage (2-D AutoencoderKL decoder, the one the shipped configs wire)
# --------------------------------------------------------------------------------------
def _vae_resnet(sd, p, x):
    """ResnetBlock.forward with temb=None (model.py:131-151); GroupNorm eps 1e-6 (:52-55)."""
    h = F.conv2d(F.silu(_gn(sd, p + ".norm1", x, 1e-6)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(sd, p + ".norm2", h, 1e-6)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".nin_shortcut.weight"], sd[p + ".nin_shortcut.bias"])
    return x + h


def _vae_attn(sd, p, x):
    """AttnBlock (model.py:161-200): single head, d = C, 1x1-conv q/k/v/proj_out."""
    b, c, hh, ww = x.shape
    n = _gn(sd, p + ".norm", x, 1e-6)
    q, k, v = [F.conv2d(n, sd[f"{p}.{t}.weight"], sd[f"{p}.{t}.bias"]).reshape(b, c, -1).transpose(1, 2)
               for t in ("q", "k", "v")]
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = o.transpose(1, 2).reshape(b, c, hh, ww)
    return x + F.conv2d(o, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])


def vae_decode(sd, dd, z, scale_factor=0.18215, prefix="first_stage_model."):
    """DiffusionEngine.decode_first_stage (models/diffusion.py:117-135) ->
    AutoencodingEngineLegacy.decode (models/autoencoder.py:490-505: post_quant_conv) ->
    Decoder.forward (model.py:715-748).  dd = ddconfig dict."""
    P = prefix
    z = z / scale_factor
    z = F.conv2d(z, sd[P + "post_quant_conv.weight"], sd[P + "post_quant_conv.bias"])
    D = P + "decoder."
    h = F.conv2d(z, sd[D + "conv_in.weight"], sd[D + "conv_in.bias"], padding=1)
    h = _vae_resnet(sd, D + "mid.block_1", h)
    h = _vae_attn(sd, D + "mid.attn_1", h)
    h = _vae_resnet(sd, D + "mid.block_2", h)
    nlev = len(dd["ch_mult"])
    for lvl in reversed(range(nlev)):
        for blk in range(dd["num_res_blocks"] + 1):
            h = _vae_resnet(sd, f"{D}up.{lvl}.block.{blk}", h)
        if lvl != 0:                                  # Upsample (model.py:58-71)
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"),
                         sd[f"{D}up.{lvl}.upsample.conv.weight"], sd[f"{D}up.{lvl}.upsample.conv.bias"], padding=1)
    h = F.silu(_gn(sd, D + "norm_out", h, 1e-6))
    return F.conv2d(h, sd[D + "conv_out.weight"], sd[D + "conv_out.bias"], padding=1)


def vae_encode_moments(sd, dd, x, prefix="first_stage_model."):
    """Encoder.forward (model.py:576-601) + quant_conv (models/autoencoder.py:468-476):
    returns the posterior moments [N, 2*Cz, h, w] (mean || logvar)."""
    P = prefix
    E = P + "encoder."
    h = F.conv2d(x, sd[E + "conv_in.weight"], sd[E + "conv_in.bias"], padding=1)
    nlev = len(dd["ch_mult"])
    for lvl in range(nlev):
        for blk in range(dd["num_res_blocks"]):
            h = _vae_resnet(sd, f"{E}down.{lvl}.block.{blk}", h)
        if lvl != nlev - 1:                          # Downsample: pad (0,1,0,1) then stride-2 conv (model.py:76-90)
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[f"{E}down.{lvl}.downsample.conv.weight"],
                         sd[f"{E}down.{lvl}.downsample.conv.bias"], stride=2)
    h = _vae_resnet(sd, E + "mid.block_1", h)
    h = _vae_attn(sd, E + "mid.attn_1", h)
    h = _vae_resnet(sd, E + "mid.block_2", h)
    h = F.silu(_gn(sd, E + "norm_out", h, 1e-6))
    h = F.conv2d(h, sd[E + "conv_out.weight"], sd[E + "conv_out.bias"], padding=1)
    return F.conv2d(h, sd[P + "quant_conv.weight"], sd[P + "quant_conv.bias"])


def vae_encode(sd, dd, x, noise=None, scale_factor=0.18215, prefix="first_stage_model."):
    """encode_first_stage (models/diffusion.py:137-150): posterior sample (noise given) or mode,
    times scale_factor.  DiagonalGaussianDistribution: distributions.py:24-41."""
    mean, logvar = vae_encode_moments(sd, dd, x, prefix).chunk(2, dim=1)
    z = mean if noise is None else mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise
    return scale_factor * z


def v02_refine(sd, cfg, z_frames, init_noise, c, uc, T, num_steps, max_scale, alpha_pow=40.0,
               prefix="model.diffusion_model."):
    """The stage-2 loop of pipeline_i2v_eval_v02.py:103-135: before every Euler step the latents are
    pulled back towards (noise*sigma_i + z) with alpha_i = (0.5(1+cos(i/num_steps)))^40."""
    sig = edm_sigmas(num_steps)
    scale = torch.linspace(1.0, max_scale, T)
    lat = init_noise * torch.sqrt(1.0 + sig[0] ** 2)
    for i in range(num_steps):
        a = (0.5 * (1.0 + math.cos(i * 1.0 / num_steps))) ** alpha_pow
        lat = lat * (1.0 - a) + (init_noise * sig[i] + z_frames) * a
        d = denoise_cfg(sd, cfg, lat, sig[i], c, uc, T, scale, prefix)
        lat = lat + (sig[i + 1] - sig[i]) * (lat - d) / sig[i]
    return lat


def video_decode(sd, dd, z, T, prefix="first_stage_model."):
    """VideoDecoder.forward, time_mode 'conv-only' (temporal_ae.py:18-107,293-349 on top of Decoder.forward
    model.py:715-748); video_kernel_size is read off the weights: [3,1,1] (SVD / Hi3D) or the class default 3 = isotropic
    3x3x3, padding k // 2 either way (temporal_ae.py:87-98, openaimodel.py:257-261 with dims=3); z already divided by
    scale_factor; no post_quant_conv (AutoencodingEngine)."""
    D = prefix + "decoder."
    n = z.shape[0]
    b = n // T
    pad = lambda w: tuple(k // 2 for k in w.shape[2:])

    def vres(p, x):
        x = _vae_resnet(sd, p, x)
        c, hh, ww = x.shape[1:]
        x5 = x.reshape(b, T, c, hh, ww).permute(0, 2, 1, 3, 4)
        q = p + ".time_stack"          # ResBlock(dims=3, skip_t_emb=True): openaimodel.py:328-354 without emb
        h = F.conv3d(F.silu(_gn(sd, q + ".in_layers.0", x5, 1e-5)), sd[q + ".in_layers.2.weight"], sd[q + ".in_layers.2.bias"], padding=pad(sd[q + ".in_layers.2.weight"]))
        h = F.conv3d(F.silu(_gn(sd, q + ".out_layers.0", h, 1e-5)), sd[q + ".out_layers.3.weight"], sd[q + ".out_layers.3.bias"], padding=pad(sd[q + ".out_layers.3.weight"]))
        a = torch.sigmoid(sd[p + ".mix_factor"])
        out = a * (x5 + h) + (1.0 - a) * x5
        return out.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)

    h = F.conv2d(z, sd[D + "conv_in.weight"], sd[D + "conv_in.bias"], padding=1)
    h = vres(D + "mid.block_1", h)
    h = _vae_attn(sd, D + "mid.attn_1", h)
    h = vres(D + "mid.block_2", h)
    for lvl in reversed(range(len(dd["ch_mult"]))):
        for blk in range(dd["num_res_blocks"] + 1):
            h = vres(f"{D}up.{lvl}.block.{blk}", h)
        if lvl != 0:
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"),
                         sd[f"{D}up.{lvl}.upsample.conv.weight"], sd[f"{D}up.{lvl}.upsample.conv.bias"], padding=1)
    h = F.silu(_gn(sd, D + "norm_out", h, 1e-6))
    h = F.conv2d(h, sd[D + "conv_out.weight"], sd[D + "conv_out.bias"], padding=1)          # AE3DConv: 2-D conv ...
    c, hh, ww = h.shape[1:]
    h5 = h.reshape(b, T, c, hh, ww).permute(0, 2, 1, 3, 4)                                   # ... then time_mix_conv
    h5 = F.conv3d(h5, sd[D + "conv_out.time_mix_conv.weight"], sd[D + "conv_out.time_mix_conv.bias"], padding=pad(sd[D + "conv_out.time_mix_conv.weight"]))
    return h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


# --------------------------------------------------------------------------------------
# conditioner: CLIP vision towers (once per clip)
# --------------------------------------------------------------------------------------
# PARITY PIN: the towers live in third-party packages that are absent from /root/reference and from this
# container -- open_clip (environments.yaml:151 `open-clip-torch==2.24.0`, ViT-H-14 / laion2b_s32b_b79k, built at
# sgm/modules/encoders/modules.py:592-596 and called at :700-704) and OpenAI `clip` (environments.yaml:53
# `clip==1.0`, ViT-L/14, vtdm/encoders.py:59,86).  This restates their published VisionTransformer forward
# (open_clip/transformer.py `VisionTransformer.forward` with pool_type 'tok', no patch dropout; clip/model.py
# `VisionTransformer.forward`), keyed by THEIR state_dict names, and is pinned against an independent
# implementation of the same architecture that IS installed here: HuggingFace `transformers.CLIPVisionModelWithProjection`
# (its conversion script maps exactly these names) -- oracle/gen_golden_clip.py -> tests/golden/clip_*.pt.
# Pinned against the reference's own packages it is not: "parity unpinned" for open_clip / clip proper.
def clip_visual(sd, img, heads, act="gelu", prefix="visual."):
    """img fp32 [B,3,H,W] (already CLIP-normalised) -> pooled, projected embedding [B, out_dim].

    sd keys (open_clip / OpenAI clip): conv1.weight [W,3,p,p]; class_embedding [W]; positional_embedding [1+g*g, W];
    ln_pre / ln_post; transformer.resblocks.{i}.{ln_1, attn.in_proj_weight [3W,W], attn.in_proj_bias,
    attn.out_proj, ln_2, mlp.c_fc, mlp.c_proj}; proj [W, out]."""
    g = lambda k: sd[prefix + k]
    w = g("conv1.weight")
    x = F.conv2d(img, w, None, stride=w.shape[-1])                       # [B, W, gh, gw]
    B, Wd = x.shape[0], x.shape[1]
    x = x.reshape(B, Wd, -1).permute(0, 2, 1)                            # [B, g*g, W]
    x = torch.cat([g("class_embedding").reshape(1, 1, Wd).expand(B, 1, Wd), x], dim=1) + g("positional_embedding")[None]
    x = F.layer_norm(x, (Wd,), g("ln_pre.weight"), g("ln_pre.bias"), 1e-5)
    i = 0
    while (prefix + f"transformer.resblocks.{i}.ln_1.weight") in sd:
        p = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (Wd,), g(p + "ln_1.weight"), g(p + "ln_1.bias"), 1e-5)
        q, k, v = F.linear(h, g(p + "attn.in_proj_weight"), g(p + "attn.in_proj_bias")).chunk(3, dim=-1)
        S, d = q.shape[1], Wd // heads
        q, k, v = [t.reshape(B, S, heads, d).transpose(1, 2) for t in (q, k, v)]
        a = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1) @ v
        x = x + F.linear(a.transpose(1, 2).reshape(B, S, Wd), g(p + "attn.out_proj.weight"), g(p + "attn.out_proj.bias"))
        h = F.layer_norm(x, (Wd,), g(p + "ln_2.weight"), g(p + "ln_2.bias"), 1e-5)
        h = F.linear(h, g(p + "mlp.c_fc.weight"), g(p + "mlp.c_fc.bias"))
        h = F.gelu(h) if act == "gelu" else h * torch.sigmoid(1.702 * h)  # nn.GELU (ViT-H-14) / QuickGELU (OpenAI ViT-L/14)
        x = x + F.linear(h, g(p + "mlp.c_proj.weight"), g(p + "mlp.c_proj.bias"))
        i += 1
    pooled = F.layer_norm(x[:, 0], (Wd,), g("ln_post.weight"), g("ln_post.bias"), 1e-5)
    return pooled @ g("proj")


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def kornia_resize(img, size, interpolation="bicubic", align_corners=True, antialias=True):
    """kornia.geometry.resize as the reference calls it (sgm/modules/encoders/modules.py:619-625) -- kornia 0.6.9
    (environments.yaml:107), a third-party package ABSENT from this image: its published algorithm
    (kornia/geometry/transform/affwarp.py:resize, kornia/filters/gaussian.py, kornia/filters/kernels.py:gaussian) restated;
    parity unpinned against the package itself.  A down-scale is pre-blurred with a separable Gaussian
    (sigma = (factor - 1) / 2 per axis, kernel size int(max(4 sigma, 3)) made odd, 'reflect' border), then
    F.interpolate(mode, align_corners); an input already at `size` is returned untouched."""
    h, w = img.shape[-2:]
    if (h, w) == tuple(size):
        return img
    factors = (h / size[0], w / size[1])
    if antialias and max(factors) > 1:
        sigmas = (max((factors[0] - 1.0) / 2.0, 0.001), max((factors[1] - 1.0) / 2.0, 0.001))
        ks = [int(max(2.0 * 2 * sigmas[0], 3)), int(max(2.0 * 2 * sigmas[1], 3))]
        ks = [k + 1 if k % 2 == 0 else k for k in ks]

        def gauss(k, sg):
            xs = torch.arange(k, dtype=img.dtype) - k // 2
            g = torch.exp(-xs * xs / (2.0 * sg * sg))
            return g / g.sum()
        ky, kx = gauss(ks[0], sigmas[0]), gauss(ks[1], sigmas[1])
        C = img.shape[1]
        img = F.pad(img, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
        img = F.conv2d(img, kx.view(1, 1, 1, -1).expand(C, 1, 1, -1), groups=C)       # filter2d_separable: x, then y
        img = F.conv2d(img, ky.view(1, 1, -1, 1).expand(C, 1, -1, 1), groups=C)
    return F.interpolate(img, size=tuple(size), mode=interpolation, align_corners=align_corners)


def openclip_image_embedder(sd, img, heads, n_cond_frames=1, n_copies=1, prefix="open_clip.model.visual.", size=224):
    """FrozenOpenCLIPImagePredictionEmbedder (sgm/modules/encoders/modules.py:1028-1046) around
    FrozenOpenCLIPImageEmbedder.forward (:641-692, ucg_rate 0 path): kornia resize to 224 (bicubic, align_corners,
    antialias -- `kornia_resize` above), [-1,1] -> [0,1], CLIP mean / std, vision tower, then "(b t) d -> b t d" and the
    n_copies repeat."""
    img = kornia_resize(img, (size, size))
    x = (img + 1.0) / 2.0
    x = (x - torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)) / torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    z = clip_visual(sd, x, heads, "gelu", prefix)
    z = z.reshape(-1, n_cond_frames, z.shape[-1])
    return z.repeat_interleave(n_copies, dim=0)


def aes_embedder(sd, video, heads=16, clip_prefix="aesthetic_model.visual.", mlp_prefix="aesthetic_mlp.layers."):
    """AesEmbedder.forward (vtdm/encoders.py:73-91): middle frame, bilinear resize to 224 x 384, centre crop 224,
    CLIP normalisation, OpenAI CLIP ViT-L/14 image features, L2 normalisation (tools/aes_score.py:56-61), the
    5-layer aesthetic MLP (tools/aes_score.py:21-30, dropouts inert in eval), then [score, timestep_embedding(100 score, 255)]."""
    B, C, T, H, W = video.shape
    y = video[:, :, T // 2]
    y = F.interpolate(y, [224, 384], mode="bilinear")[:, :, :, 80:304]
    y = (y + 1) * 0.5
    y = (y - torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)) / torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    f = clip_visual(sd, y, heads, "quick_gelu", clip_prefix)
    n = f.norm(dim=-1, keepdim=True)
    f = f / torch.where(n == 0, torch.ones_like(n), n)
    for i in (0, 2, 4, 6, 7):
        f = F.linear(f, sd[mlp_prefix + f"{i}.weight"], sd[mlp_prefix + f"{i}.bias"])
    return torch.cat([f, sinusoid(f[:, 0] * 100, 255)], dim=1)


# --------------------------------------------------------------------------------------
# depth conditioner (stage-2 / v02): MiDaS DPT-hybrid, then min-max normalisation and a 3x3 pixel-unshuffle
# --------------------------------------------------------------------------------------
# PARITY PIN: the DPT side (hooks, readout projection, reassemble, fusion, head) restates the reference's vendored
# annotator/midas/{vit,blocks,dpt_depth}.py and is pinned against THAT code (tests/golden/dpt_hybrid_64x96.pt,
# oracle/gen_golden_dpt.py); depth_embedder() is pinned against the reference's own vtdm.encoders.DepthEmbedder.forward
# (same fixture: the class run with its constructor bypassed, bit-exact).  The backbone is timm's `vit_base_resnet50_384` (annotator/midas/vit.py:499; timm is a
# pip dependency, absent from /root/reference and from this container): restated from the published BiT-ResNetV2 /
# ViT architecture, and pinned against HuggingFace's independent DPTForDepthEstimation(is_hybrid=True) with its own
# BiT backbone (8.8e-5 on shared weights, gen_golden_dpt.py --check-hf) -- "parity unpinned" against timm proper.
def _tf_same_pad(x, k, s, value=0.0):
    ih, iw = x.shape[-2:]
    ph = max((-(-ih // s) - 1) * s + k - ih, 0)
    pw = max((-(-iw // s) - 1) * s + k - iw, 0)
    return F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=value) if (ph or pw) else x


def _std_conv(sd, p, x, stride=1):
    """timm StdConv2dSame(eps=1e-8): per-output-channel weight standardisation (biased variance); symmetric padding at
    stride 1, TensorFlow 'SAME' (the odd unit after) otherwise."""
    w = sd[p + ".weight"]
    k = w.shape[-1]
    w = (w - w.mean(dim=(1, 2, 3), keepdim=True)) / torch.sqrt(w.var(dim=(1, 2, 3), keepdim=True, unbiased=False) + 1e-8)
    if stride != 1:
        return F.conv2d(_tf_same_pad(x, k, stride), w, None, stride)
    return F.conv2d(x, w, None, 1, (k - 1) // 2)


def _gn32(sd, p, x, relu):
    x = F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-5)
    return F.relu(x) if relu else x


def _bit_stage(sd, p, x, stride, depth):
    """BiT bottlenecks, non-pre-activation: conv1x1-GN-ReLU, conv3x3(stride)-GN-ReLU, conv1x1-GN, + shortcut
    (first block: conv1x1(stride)-GN), ReLU."""
    for b in range(depth):
        q = f"{p}.blocks.{b}"
        st = stride if b == 0 else 1
        sc = _gn32(sd, q + ".downsample.norm", _std_conv(sd, q + ".downsample.conv", x, st), False) if b == 0 else x
        h = _gn32(sd, q + ".norm1", _std_conv(sd, q + ".conv1", x), True)
        h = _gn32(sd, q + ".norm2", _std_conv(sd, q + ".conv2", h, st), True)
        h = _gn32(sd, q + ".norm3", _std_conv(sd, q + ".conv3", h), False)
        x = F.relu(h + sc)
    return x


def dpt_hybrid(sd, x, prefix="model.model.", heads=12, return_layers=False):
    """DPTDepthModel(backbone='vitb_rn50_384', non_negative=True).forward (annotator/midas/dpt_depth.py:60-106):
    x fp32 [B,3,H,W], H and W multiples of 32 -> inverse depth [B,H,W]."""
    g = lambda k: sd[prefix + k]
    sub = lambda p: {k[len(prefix + p):]: v for k, v in sd.items() if k.startswith(prefix + p)}
    B, _, H, W = x.shape
    # ---- backbone (timm VisionTransformer with HybridEmbed; forward_flex, annotator/midas/vit.py:126-160)
    bb = sub("pretrained.model.patch_embed.backbone.")
    h = _gn32(bb, "stem.norm", _std_conv(bb, "stem.conv", x, 2), True)
    h = F.max_pool2d(_tf_same_pad(h, 3, 2, -float("inf")), 3, 2)
    l1 = _bit_stage(bb, "stages.0", h, 1, 3)                         # hook '1': 256 ch, H/4
    l2 = _bit_stage(bb, "stages.1", l1, 2, 4)                        # hook '2': 512 ch, H/8
    l3 = _bit_stage(bb, "stages.2", l2, 2, 9)                        # 1024 ch, H/16
    P = "pretrained.model."
    t = F.conv2d(l3, g(P + "patch_embed.proj.weight"), g(P + "patch_embed.proj.bias")).flatten(2).transpose(1, 2)
    gh, gw = H // 16, W // 16
    pe = g(P + "pos_embed")
    g0 = int(math.sqrt(pe.shape[1] - 1))
    grid = F.interpolate(pe[0, 1:].reshape(1, g0, g0, -1).permute(0, 3, 1, 2), size=(gh, gw), mode="bilinear")
    pe = torch.cat([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)], dim=1)
    t = torch.cat([g(P + "cls_token").expand(B, -1, -1), t], dim=1) + pe
    Wd = t.shape[-1]
    hooks = {}
    for i in range(12):
        p = P + f"blocks.{i}."
        n = F.layer_norm(t, (Wd,), g(p + "norm1.weight"), g(p + "norm1.bias"), 1e-6)
        q, k, v = F.linear(n, g(p + "attn.qkv.weight"), g(p + "attn.qkv.bias")).chunk(3, dim=-1)
        S, d = q.shape[1], Wd // heads
        q, k, v = [u.reshape(B, S, heads, d).transpose(1, 2) for u in (q, k, v)]
        a = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1) @ v
        t = t + F.linear(a.transpose(1, 2).reshape(B, S, Wd), g(p + "attn.proj.weight"), g(p + "attn.proj.bias"))
        n = F.layer_norm(t, (Wd,), g(p + "norm2.weight"), g(p + "norm2.bias"), 1e-6)
        t = t + F.linear(F.gelu(F.linear(n, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias"))), g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
        if i in (8, 11):
            hooks[i] = t
    # ---- reassemble (vit.py:446-476): ProjectReadout -> [B, 768, gh, gw] -> conv1x1 (-> conv3x3 stride 2)
    def readout(tok, p):
        cls = tok[:, :1].expand(-1, tok.shape[1] - 1, -1)
        f = F.gelu(F.linear(torch.cat([tok[:, 1:], cls], dim=-1), g(p + ".0.project.0.weight"), g(p + ".0.project.0.bias")))
        f = f.transpose(1, 2).reshape(B, Wd, gh, gw)
        return F.conv2d(f, g(p + ".3.weight"), g(p + ".3.bias"))
    r3 = readout(hooks[8], "pretrained.act_postprocess3")
    r4 = readout(hooks[11], "pretrained.act_postprocess4")
    r4 = F.conv2d(r4, g("pretrained.act_postprocess4.4.weight"), g("pretrained.act_postprocess4.4.bias"), 2, 1)
    layers = [l1, l2, r3, r4]
    # ---- fusion decoder (dpt_depth.py:71-82; blocks.py:261-390): features 256, ReLU, no BN, align_corners=True
    rn = [F.conv2d(layers[i], g(f"scratch.layer{i + 1}_rn.weight"), None, 1, 1) for i in range(4)]

    def rcu(p, u):
        o = F.conv2d(F.relu(u), g(p + ".conv1.weight"), g(p + ".conv1.bias"), 1, 1)
        return F.conv2d(F.relu(o), g(p + ".conv2.weight"), g(p + ".conv2.bias"), 1, 1) + u

    def fusion(p, a, b=None):
        o = a if b is None else a + rcu(p + ".resConfUnit1", b)
        o = F.interpolate(rcu(p + ".resConfUnit2", o), scale_factor=2, mode="bilinear", align_corners=True)
        return F.conv2d(o, g(p + ".out_conv.weight"), g(p + ".out_conv.bias"))
    path = fusion("scratch.refinenet4", rn[3])
    path = fusion("scratch.refinenet3", path, rn[2])
    path = fusion("scratch.refinenet2", path, rn[1])
    path = fusion("scratch.refinenet1", path, rn[0])
    # ---- head (dpt_depth.py:88-104)
    o = F.conv2d(path, g("scratch.output_conv.0.weight"), g("scratch.output_conv.0.bias"), 1, 1)
    o = F.interpolate(o, scale_factor=2, mode="bilinear", align_corners=True)
    o = F.relu(F.conv2d(o, g("scratch.output_conv.2.weight"), g("scratch.output_conv.2.bias"), 1, 1))
    o = F.relu(F.conv2d(o, g("scratch.output_conv.4.weight"), g("scratch.output_conv.4.bias")))
    return (o.squeeze(1), layers) if return_layers else o.squeeze(1)


def depth_embedder(sd, x, T=16, shuffle_size=3, scale_factor=2.6666, prefix="model.model."):
    """DepthEmbedder.forward (vtdm/encoders.py:31-53; use_3d False): x [(b t),3,H,W] in [-1,1] -> bilinear resize to
    the multiple of 32 below H / 2.6666 -> DPT-hybrid -> bilinear resize to (H/8*3, W/8*3) -> per-image
    (y - min) / max(max, 1e-6) -> 'b c (h h0) (w w0) -> b (c h0 w0) h w' -> [(b t), 9, H/8, W/8]."""
    H, W = x.shape[-2:]
    sH, sW = int(H / scale_factor / 32) * 32, int(W / scale_factor / 32) * 32
    y = dpt_hybrid(sd, F.interpolate(x, [sH, sW], mode="bilinear"), prefix)[:, None]
    y = F.interpolate(y, [H // 8 * shuffle_size, W // 8 * shuffle_size], mode="bilinear")
    y = y - y.amin(dim=(1, 2, 3), keepdim=True)
    y = y / y.amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-6)
    B, _, Hh, Ww = y.shape
    s = shuffle_size
    return y.reshape(B, 1, Hh // s, s, Ww // s, s).permute(0, 1, 3, 5, 2, 4).reshape(B, s * s, Hh // s, Ww // s)
