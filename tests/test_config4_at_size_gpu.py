"""Parity AT BASELINE CONFIG 4's REAL SIZE: stage 2, 32 views @ 1024x1024 -- T = 32, latent 128 x 128, CFG batch 64 frames, i.e.
M = 64 * 16384 = 1,048,576 token rows at the top level.  The kernels address global memory with
32-bit per-lane offsets on per-block buffer descriptors; at this size a [M, 1280] bf16 tensor is 2.7 GB and a [M, 960] one 2.0 GB,
so the last rows lie beyond 2^31 bytes from the tensor base.  Every launch family is run at M = 1,048,576 on device-generated
inputs and its first / middle / LAST rows are compared with fp32 CPU arithmetic on the same (bf16-rounded) inputs; the spatial
attention at B = 64, H = 5, S = 16384; and the whole VideoUNet forward at that size through a size-independent property (the
two CFG halves of a batch whose halves are identical must come out bit-identical, and equal to the 32-frame half run alone).

Tolerances as in tests/test_at_size_gpu.py: max-abs error <= 1.2e-2 x max-abs reference (2e-2 for attention: P is bf16)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16_TOL = 1.2e-2
M1 = 64 * 16384


def drand(shape, seed, dev, scale=1.0, shift=0.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(shape, device=dev, generator=g) * scale + shift).to(torch.bfloat16)


def crand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def rows_sample(M, dev):
    """first 300, 300 around the middle (both sides of the 2^31-byte line of the wide tensors), last 300"""
    return torch.cat([torch.arange(0, 300), torch.arange(M // 2 - 150, M // 2 + 150), torch.arange(M - 300, M)]).to(dev)


def test_dense_qkv_and_proj_M1048576(dev):
    from hi3d_hip import ops
    M, C = M1, 320
    x = drand((M, C), 1, dev)
    wq = crand((3 * C, C), 2, C ** -0.5).to(torch.bfloat16)
    sl = rows_sample(M, dev)
    qkv = ops.gemm(x, wq.to(dev), M=M, N=3 * C, K=C)                           # [M, 960]: 2.0 GB
    assert relerr(qkv[sl], x[sl].float().cpu() @ wq.float().T) < BF16_TOL
    wo, bo = crand((C, C), 3, C ** -0.5).to(torch.bfloat16), crand((C,), 4)
    rv = crand((64, C + 8), 5)
    R1 = drand((M, C), 6, dev)
    out = ops.gemm(x, wo.to(dev), M=M, N=C, K=C, bias=bo.to(dev), rowvec=rv.to(dev), ldrv=C + 8, rows_per_group=16384, R1=R1)
    grp = (sl // 16384).cpu()
    ref = x[sl].float().cpu() @ wo.float().T + bo + rv[grp, :C] + R1[sl].float().cpu()
    assert relerr(out[sl], ref) < BF16_TOL


def test_geglu_gemm_M1048576(dev):
    """GEGLU GEMM of the 320-channel feed-forward (the two-GEMM path, HI3D_FUSED_FFN=0): output [M, 1280] = 2.7 GB"""
    from hi3d_hip import ops
    from hi3d_hip.pack import pack_geglu
    M, C = M1, 320
    x = drand((M, C), 11, dev)
    w, b = crand((8 * C, C), 12, C ** -0.5).to(torch.bfloat16).float(), crand((8 * C,), 13)
    wp, bp = pack_geglu(w, b)
    out = ops.gemm(x, wp.to(dev), M=M, N=8 * C, K=C, bias=bp.to(dev), geglu=True)
    sl = rows_sample(M, dev)
    h = x[sl].float().cpu() @ w.T + b
    assert relerr(out[sl], h[:, :4 * C] * F.gelu(h[:, 4 * C:])) < BF16_TOL
    del out
    # and the second GEMM of that path: K = 1280 over the 2.7 GB operand
    hg = drand((M, 4 * C), 14, dev)
    w2, b2 = crand((C, 4 * C), 15, (4 * C) ** -0.5).to(torch.bfloat16), crand((C,), 16)
    R1 = drand((M, C), 17, dev)
    o2 = ops.gemm(hg, w2.to(dev), M=M, N=C, K=4 * C, bias=b2.to(dev), R1=R1)
    assert relerr(o2[sl], hg[sl].float().cpu() @ w2.float().T + b2 + R1[sl].float().cpu()) < BF16_TOL


def test_ffn_fused_M1048576(dev):
    from hi3d_hip import ops
    from hi3d_hip.pack import pack_geglu, pack_linear
    M, C = M1, 320
    x = drand((M, C), 21, dev)
    w1, b1 = crand((8 * C, C), 22, C ** -0.5).to(torch.bfloat16).float(), crand((8 * C,), 23)
    w2, b2 = crand((C, 4 * C), 24, (4 * C) ** -0.5).to(torch.bfloat16).float(), crand((C,), 25)
    R1, R2 = drand((M, C), 26, dev), drand((M, C), 27, dev)
    a1, a2 = crand((64,), 28).abs() + 0.5, crand((64,), 29)
    w1p, b1p = pack_geglu(w1, b1)
    out = ops.ffn_geglu(x, w1p.to(dev), b1p.to(dev), pack_linear(w2).to(dev), b2.to(dev), M=M, C=C, R1=R1, R2=R2,
                        a1=a1.to(dev), a2=a2.to(dev), rows_per_group=16384)
    sl = rows_sample(M, dev)
    h = x[sl].float().cpu() @ w1.T + b1
    hg = (h[:, :4 * C] * F.gelu(h[:, 4 * C:])).to(torch.bfloat16).float()
    grp = (sl // 16384).cpu()
    ref = (hg @ w2.T + b2 + R1[sl].float().cpu()) * a1[grp, None] + a2[grp, None] * R2[sl].float().cpu()
    assert relerr(out[sl], ref) < BF16_TOL


def test_layernorm_M1048576(dev):
    from hi3d_hip import ops
    M, C = M1, 320
    x = drand((M, C), 31, dev, 1.5, 0.3)
    g, b = 1 + 0.1 * crand((C,), 32), 0.1 * crand((C,), 33)
    pos = crand((64, C), 34)
    xm = torch.empty_like(x)
    out = ops.layernorm(x, g.to(dev), b.to(dev), M, C, addvec=pos.to(dev), rows_per_group=16384, sum_out=xm)
    sl = rows_sample(M, dev)
    s = x[sl].float().cpu() + pos[(sl // 16384).cpu()]
    assert relerr(xm[sl], s) < BF16_TOL
    assert relerr(out[sl], F.layer_norm(s, (C,), g, b, 1e-5)) < BF16_TOL


@pytest.mark.parametrize("three_d", [False, True])
def test_groupnorm_M1048576(dev, three_d):
    """2-D: 64 frames of 16384 pixels; 3-D time_stack norm: 2 clips of 32 x 16384 = 524288 positions each"""
    from hi3d_hip import ops
    C = 320
    inst, P = (2, 32 * 16384) if three_d else (64, 16384)
    x = drand((inst * P, C), 41, dev, 1.5, -0.4)
    g, b = 1 + 0.1 * crand((C,), 42), 0.1 * crand((C,), 43)
    out = ops.groupnorm_silu(x, g.to(dev), b.to(dev), inst, P, C, 1e-5, True)
    # statistics in fp64 on the device (plain torch reductions: independent of the kernel), the affine + SiLU on sampled rows in fp32
    xf = x.float().reshape(inst, P, 32, C // 32)
    mean = xf.double().mean(dim=(1, 3))
    var = (xf.double() ** 2).mean(dim=(1, 3)) - mean ** 2
    del xf
    sl = rows_sample(inst * P, dev)
    ii = (sl // P)
    xs = x[sl].float().reshape(-1, 32, C // 32)
    y = (xs - mean[ii].float()[:, :, None]) * torch.rsqrt(var[ii].float() + 1e-5)[:, :, None]
    ref = F.silu(y.reshape(-1, C) * g.to(dev) + b.to(dev))
    assert relerr(out[sl], ref) < BF16_TOL


def test_conv3x3_M1048576(dev):
    """ResBlock conv at the top level: 64 frames x 128 x 128, 320 -> 320, bias + row vector + residual"""
    from hi3d_hip import ops
    from hi3d_hip.pack import pack_conv3x3
    Fr, H, C = 64, 128, 320
    M = Fr * H * H
    x = drand((M, C), 51, dev)
    w, bias = crand((C, C, 3, 3), 52, (9 * C) ** -0.5).to(torch.bfloat16).float(), crand((C,), 53)
    rv = crand((Fr, C), 54)
    R1 = drand((M, C), 55, dev)
    out = ops.gemm(x, pack_conv3x3(w).to(dev), M=M, N=C, K=9 * C, bias=bias.to(dev), rowvec=rv.to(dev), ldrv=C, rows_per_group=H * H,
                   R1=R1, conv3x3=dict(Hin=H, Win=H, Cin=C, Hout=H, Wout=H, stride=1, up2x=0))
    # frames 0, 31 | 32 (either side of the 2^31-byte line of the input) and 63: whole frames through F.conv2d on the CPU, rows 0..2 / 126..127
    for f in (0, 31, 32, 63):
        xf = x[f * H * H:(f + 1) * H * H].float().cpu().reshape(H, H, C).permute(2, 0, 1)[None]
        ref = F.conv2d(xf, w, bias, padding=1)[0].permute(1, 2, 0) + rv[f] + R1[f * H * H:(f + 1) * H * H].float().cpu().reshape(H, H, C)
        got = out[f * H * H:(f + 1) * H * H].reshape(H, H, C)
        assert relerr(got, ref) < BF16_TOL, f"frame {f}"


def test_convt3_T32_M1048576(dev):
    """time_stack Conv3d (3,1,1) at the top level: 2 clips x 32 frames x 16384 pixels, 320 -> 320, AlphaBlender tail"""
    from hi3d_hip import ops
    from hi3d_hip.pack import pack_convt3
    B, T, HW, C = 2, 32, 16384, 320
    M = B * T * HW
    x = drand((M, C), 61, dev)
    w, bias = crand((C, C, 3, 1, 1), 62, (3 * C) ** -0.5).to(torch.bfloat16).float(), crand((C,), 63)
    a1 = crand((B * T,), 64).abs() + 0.25
    R2 = drand((M, C), 65, dev)
    out = ops.gemm(x, pack_convt3(w).to(dev), M=M, N=C, K=3 * C, bias=bias.to(dev), a1=a1.to(dev), R2=R2, rows_per_group=HW,
                   convt3=dict(T=T, HW=HW, Cin=C))
    px = torch.cat([torch.arange(0, 40), torch.arange(HW - 40, HW)])
    for b in range(B):
        xs = x.reshape(B, T, HW, C)[b][:, px.to(dev)].float().cpu()                       # [T, npx, C]
        ref = F.conv3d(xs.permute(2, 0, 1)[None, :, :, :, None], w, bias, padding=(1, 0, 0))[0, :, :, :, 0].permute(1, 2, 0)   # [T, npx, C]
        r2 = R2.reshape(B, T, HW, C)[b][:, px.to(dev)].float().cpu()
        ref = ref * a1[b * T:(b + 1) * T, None, None] + r2
        got = out.reshape(B, T, HW, C)[b][:, px.to(dev)]
        assert relerr(got, ref) < BF16_TOL, f"clip {b}"


def test_attention_B64_H5_S16384(dev):
    """spatial self-attention of config 4's top level: 64 frames x 5 heads x 16384 tokens (fused qkv tensor: 2.0 GB)"""
    from hi3d_hip import ops
    B, H, S = 64, 5, 16384
    C = H * 64
    qkv = drand((B * S, 3 * C), 71, dev)
    out = ops.self_attention_fused_qkv(qkv, B, S, H)
    rows = torch.cat([torch.arange(0, 64), torch.arange(S // 2, S // 2 + 64), torch.arange(S - 64, S)])
    for b, h in ((0, 0), (31, 4), (32, 0), (63, 4), (63, 2)):
        blk = qkv[b * S:(b + 1) * S].float().cpu()
        q, k, v = blk[rows, h * 64:(h + 1) * 64], blk[:, C + h * 64:C + (h + 1) * 64], blk[:, 2 * C + h * 64:2 * C + (h + 1) * 64]
        ref = torch.softmax(q @ k.T * 0.125, dim=-1) @ v
        got = out[b * S:(b + 1) * S][rows.to(dev), h * 64:(h + 1) * 64]
        assert relerr(got, ref) < 2e-2, f"(b, h) = ({b}, {h})"


def test_attention_temporal_T32_S16384(dev):
    from hi3d_hip import ops
    B, T, S, H = 2, 32, 16384, 5
    C = H * 64
    qkv = drand((B * T * S, 3 * C), 81, dev)
    out = ops.attention_temporal_fused_qkv(qkv, B, T, S, H)
    px = torch.cat([torch.arange(0, 32), torch.arange(S - 32, S)]).to(dev)
    for b in range(B):
        blk = qkv.reshape(B, T, S, 3 * C)[b][:, px].float().cpu()                            # [T, npx, 3C]
        q, k, v = [blk[:, :, i * C:(i + 1) * C].reshape(T, -1, H, 64).permute(1, 2, 0, 3) for i in range(3)]   # [npx, H, T, 64]
        ref = F.scaled_dot_product_attention(q, k, v).permute(2, 0, 1, 3).reshape(T, -1, C)
        got = out.reshape(B, T, S, C)[b][:, px]
        assert relerr(got, ref) < 2e-2, f"clip {b}"


def test_unet_config4_full_size_cfg_halves(dev):
    """The WHOLE full-width stage-2 VideoUNet at config 4's size -- 64 frames (CFG pair of a 32-view clip), latent 128 x 128 --
    through a size-independent property: with the conditional and unconditional halves fed IDENTICAL inputs the two halves of
    the output are bit-identical (every kernel treats a frame / clip by the same arithmetic wherever it sits in the batch: rows
    0 .. 524287 against rows 524288 .. 1048575, the latter beyond 2^31 bytes in every wide tensor), finite, and agree with the
    32-frame half run alone (another batch size: other grids, rasters and GroupNorm partial-sum blockings) within the UNet
    tolerance (<= 4e-2, cosine >= 0.9995)."""
    from hi3d_hip import synth
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from sgm.util import ParamTree
    from test_parallel_gpu import FULL_CFG
    T, lat = 32, 128
    ParamTree.skip_init = True
    try:
        with torch.device(dev):
            unet = VideoUNet(**FULL_CFG)
    finally:
        ParamTree.skip_init = False
    synth.fill_module_on_device_(unet, seed=1, prefix="model.diffusion_model.")
    g = torch.Generator(device=dev).manual_seed(9)
    x1 = torch.randn((T, 17, lat, lat), device=dev, generator=g)
    ctx1 = torch.randn((1, 1, 1024), device=dev, generator=g)
    y1 = torch.randn((1, 512), device=dev, generator=g)
    ts = torch.full((2 * T,), 0.25 * 1.5, device=dev)
    ioi = torch.zeros(2, T, device=dev)
    with torch.no_grad():
        out = unet(torch.cat([x1, x1]), ts, context=ctx1.expand(2, 1, 1024).contiguous(), y=y1.expand(2, 512).contiguous(),
                   num_video_frames=T, image_only_indicator=ioi)
        assert out.shape == (2 * T, 4, lat, lat) and torch.isfinite(out).all()
        assert torch.equal(out[:T], out[T:])
        half = unet(x1, ts[:T], context=ctx1, y=y1, num_video_frames=T, image_only_indicator=ioi[:1])
    rel = ((half - out[:T]).abs().max() / out[:T].abs().max()).item()
    cosv = F.cosine_similarity(half.float().flatten(), out[:T].float().flatten(), dim=0).item()
    print(f"config-4 UNet: 64-frame batch halves identical; vs the 32-frame run rel {rel:.2e} cos {cosv:.6f}")
    # (another batch size re-blocks the GroupNorm partial sums and re-tiles the small levels: fp32 summation order differs, and
    # ~100 random-weight layers amplify it -- measured 1.5e-2; the bound is the UNet tolerance of tests/test_at_size_gpu.py)
    assert rel < 4e-2 and cosv > 0.9995
