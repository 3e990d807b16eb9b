"""TEST INFRASTRUCTURE -- the CPU fp32 oracle UNet (oracle/hi3d_oracle.py) executed under the
frame <-> space re-sharding of hi3d_hip.parallel.FrameSpaceGroup, one process per rank.

Purpose: prove, with the reference's arithmetic and a real process group (gloo, world_size 2 in the
CPU tests), that the multi-GPU mapping of ONE clip is exact:

  spatial sub-blocks  (2-D ResBlock, GroupNorm per frame, spatial transformer block, skip 1x1,
                       Down/Upsample, conv_in / out)            run on this rank's FRAMES,
  temporal sub-blocks (3-D ResBlock: Conv3d (3,1,1) video_model.py:42-55 + GroupNorm over (t,h,w)
                       util.py:274-276; temporal transformer block video_attention.py:109-140)
                                                                 run on this rank's PIXELS, all frames,
  with an all-to-all between them and an all-reduce of the [b, 32, 2] GroupNorm partial sums.

The same `FrameSpaceGroup` object drives the HIP runtime (hi3d_hip/runtime_unet.py, `sp=`): this module
checks the communication plan, tests/test_parallel_gpu.py checks the kernels under it.
Only tests/ import this module.
"""
import torch
import torch.nn.functional as F

from . import hi3d_oracle as O


def _to_tokens(x):                       # [n, c, h, w] -> [n*h*w, c]
    n, c, hh, ww = x.shape
    return x.reshape(n, c, hh * ww).transpose(1, 2).reshape(n * hh * ww, c)


def _to_nchw(tok, n, hh, ww):
    c = tok.shape[-1]
    return tok.reshape(n, hh * ww, c).transpose(1, 2).reshape(n, c, hh, ww)


def _gn3d_sharded(sd, p, x5, eps, comm):
    """GroupNorm(32) over (c/32, t, h, w) of 'b c t sl 1' when every rank holds only sl = S/w pixels:
    fp64 partial (sum, sum of squares) per (b, group), summed over the group of ranks."""
    b, C = x5.shape[:2]
    xg = x5.reshape(b, 32, -1).double()
    sums = torch.stack([xg.sum(-1), (xg * xg).sum(-1)], dim=-1)          # [b, 32, 2]
    comm.allreduce_sum_(sums)
    n = xg.shape[-1] * comm.world
    mean = sums[..., 0] / n
    var = (sums[..., 1] / n - mean * mean).clamp_min(0.0)
    xn = ((xg - mean[..., None]) * torch.rsqrt(var + eps)[..., None]).float().reshape(x5.shape)
    shape = (1, C) + (1,) * (x5.dim() - 2)
    return xn * sd[p + ".weight"].reshape(shape) + sd[p + ".bias"].reshape(shape)


def _resblock3d_sharded(sd, p, x5, emb_bt, comm):
    """oracle.resblock(dims=3) with the sharded norm; x5 'b c t sl 1', emb_bt [b, t, E]"""
    pad = (1, 0, 0)
    h = F.conv3d(F.silu(_gn3d_sharded(sd, p + ".in_layers.0", x5, 1e-5, comm)), sd[p + ".in_layers.2.weight"],
                 sd[p + ".in_layers.2.bias"], padding=pad)
    e = O._lin(sd, p + ".emb_layers.1", F.silu(emb_bt)).transpose(1, 2)[:, :, :, None, None]
    h = h + e
    h = F.conv3d(F.silu(_gn3d_sharded(sd, p + ".out_layers.0", h, 1e-5, comm)), sd[p + ".out_layers.3.weight"],
                 sd[p + ".out_layers.3.bias"], padding=pad)
    return x5 + h


def video_resblock_sharded(sd, p, x, emb_local, emb_full, T, ioi, comm):
    x = O.resblock(sd, p, x, emb_local, 2)                                   # this rank's frames
    n_l, c, hh, ww = x.shape
    B, S = n_l // comm.Tl, hh * ww
    Sl = S // comm.world
    sp = comm.frames_to_space(_to_tokens(x), B, S)                           # rows (b t sl)
    x5 = sp.reshape(B, T, Sl, c).permute(0, 3, 1, 2)[..., None]              # b c t sl 1
    xt = _resblock3d_sharded(sd, p + ".time_stack", x5, emb_full.reshape(B, T, -1), comm)
    a = O._alpha(sd, p + ".time_mixer", ioi)[:, None, :, None, None]
    out = a * x5 + (1.0 - a) * xt
    tok = out[..., 0].permute(0, 2, 3, 1).reshape(B * T * Sl, c)
    return _to_nchw(comm.space_to_frames(tok, B, S), n_l, hh, ww)


def spatial_video_transformer_sharded(sd, p, x, ctx_b, T, ioi, heads, comm):
    """oracle.spatial_video_transformer with the temporal block on this rank's pixels.
    ctx_b: [B, 1, ctx] one context per clip."""
    n_l, c, hh, ww = x.shape
    Tl = comm.Tl
    B, S = n_l // Tl, hh * ww
    Sl = S // comm.world
    ctx = ctx_b.repeat_interleave(Tl, dim=0)                                 # per local frame
    tctx = ctx_b.repeat_interleave(Sl, dim=0)                                # per local pixel (clip's first frame)
    h = O._gn(sd, p + ".norm", x, 1e-6).reshape(n_l, c, S).transpose(1, 2)
    h = O._lin(sd, p + ".proj_in", h)
    pos = O._mlp2(sd, p + ".time_pos_embed", O.sinusoid(torch.arange(T).repeat(B), c)).reshape(B, T, 1, c)
    a = O._alpha(sd, p + ".time_mixer", ioi).reshape(B, T, 1, 1)
    sp_, tp = f"{p}.transformer_blocks.0", f"{p}.time_stack.0"
    h = h + O._attn(sd, sp_ + ".attn1", O._ln(sd, sp_ + ".norm1", h), None, heads)
    h = h + O._attn(sd, sp_ + ".attn2", O._ln(sd, sp_ + ".norm2", h), ctx, heads)
    h = h + O._geglu_ff(sd, sp_ + ".ff", O._ln(sd, sp_ + ".norm3", h))
    hs = comm.frames_to_space(h.reshape(n_l * S, c), B, S).reshape(B, T, Sl, c)           # (b t sl)
    m = (hs + pos).permute(0, 2, 1, 3).reshape(B * Sl, T, c)
    m = m + O._geglu_ff(sd, tp + ".ff_in", O._ln(sd, tp + ".norm_in", m))
    m = m + O._attn(sd, tp + ".attn1", O._ln(sd, tp + ".norm1", m), None, heads)
    m = m + O._attn(sd, tp + ".attn2", O._ln(sd, tp + ".norm2", m), tctx, heads)
    m = m + O._geglu_ff(sd, tp + ".ff", O._ln(sd, tp + ".norm3", m))
    m = m.reshape(B, Sl, T, c).permute(0, 2, 1, 3)
    hb = a * hs + (1.0 - a) * m                                                            # blend in the space layout
    h = comm.space_to_frames(hb.reshape(B * T * Sl, c), B, S).reshape(n_l, S, c)
    h = O._lin(sd, p + ".proj_out", h)
    return h.transpose(1, 2).reshape(n_l, c, hh, ww) + x


def video_unet_sharded(sd, cfg, x_local, timestep, context_b, y_b, T, ioi, comm, prefix=""):
    """VideoUNet.forward on this rank's frames.  x_local [B*Tl, Cin, H, W] (frames t_lo..t_lo+Tl of every
    clip, clip-major); timestep: scalar c_noise (equal for all frames, as in the sampler); context_b
    [B,1,ctx]; y_b [B, adm]; ioi [B, T].  Returns [B*Tl, out, H, W]."""
    P = prefix
    B = x_local.shape[0] // comm.Tl
    nheads = lambda ch: ch // cfg["num_head_channels"]
    t_full = torch.full((B * T,), float(timestep))
    emb_full = O._mlp2(sd, P + "time_embed", O.sinusoid(t_full, cfg["model_channels"])) + \
        O._mlp2(sd, P + "label_emb.0", y_b.repeat_interleave(T, dim=0))
    emb_local = emb_full[comm.local_frames(B)]
    inp, mid, outp = O.unet_layout(cfg)

    def run(h, layers, base):
        for j, L in enumerate(layers):
            p = f"{base}.{j}"
            if L[0] == "conv_in":
                h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
            elif L[0] == "res":
                h = video_resblock_sharded(sd, p, h, emb_local, emb_full, T, ioi, comm)
            elif L[0] == "attn":
                h = spatial_video_transformer_sharded(sd, p, h, context_b, T, ioi, nheads(L[1]), comm)
            elif L[0] == "down":
                h = F.conv2d(h, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=2, padding=1)
            elif L[0] == "up":
                h = F.conv2d(F.interpolate(h, scale_factor=2, mode="nearest"), sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
        return h

    h, hs = x_local, []
    for i, layers in enumerate(inp):
        h = run(h, layers, f"{P}input_blocks.{i}")
        hs.append(h)
    h = run(h, mid, P + "middle_block")
    for i, layers in enumerate(outp):
        h = run(torch.cat([h, hs.pop()], dim=1), layers, f"{P}output_blocks.{i}")
    h = F.silu(O._gn(sd, P + "out.0", h, 1e-5))
    return F.conv2d(h, sd[P + "out.2.weight"], sd[P + "out.2.bias"], padding=1)
