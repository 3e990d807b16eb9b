"""torch.ops.hi3d.* (the TORCH_LIBRARY shim, csrc/torch_ops.cpp) against plain fp32 torch references of the reference's
modules at the plug-in points INTEGRATION.md (B) names -- and bit-identical to the ctypes path, since both reach the same
C-ABI entry points."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()


@pytest.fixture(scope="module")
def ns():
    from hi3d_hip import torch_ops
    return torch_ops.load()


@pytest.mark.parametrize("B,S,H", [(2, 1024, 5), (3, 577, 12), (1, 48, 2)])
def test_self_attention_op(dev, ns, B, S, H):
    """CrossAttention.forward's softmax(q k^T / sqrt(64)) v (attention.py:332-336) on a fused qkv projection."""
    from hi3d_hip import ops
    C = H * 64
    qkv = (torch.randn((B * S, 3 * C), generator=torch.Generator().manual_seed(S)) * 0.7).to(torch.bfloat16)
    out = ns.self_attention(qkv.to(dev), B, S, H, 0.125)
    q, k, v = (t.float().reshape(B, S, H, 64).transpose(1, 2) for t in qkv.split(C, dim=1))
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B * S, C)
    assert rel(out, ref) < 2e-2
    assert torch.equal(out, ops.self_attention_fused_qkv(qkv.to(dev), B, S, H, scale=0.125))      # same kernels through ctypes
    q2, k2, v2 = (t.contiguous().to(dev) for t in qkv.split(C, dim=1))
    assert torch.equal(ns.attn_d64(q2, k2, v2, B, H, S, S, 0.125), out)
    with pytest.raises(RuntimeError):
        ns.self_attention(qkv.to(dev).float(), B, S, H, 0.125)                                     # dtype check before any launch
    with pytest.raises(RuntimeError):
        ns.self_attention(qkv.to(dev), B, S + 1, H, 0.125)


def test_norm_linear_conv_ffn_ops(dev, ns):
    """GroupNorm32 + SiLU, LayerNorm, nn.Linear (+ residual), Conv2d 3x3 (stride 1 / 2 / nearest-2x) and FeedForward(glu)
    through torch.ops.hi3d vs fp32 torch on the same bf16-rounded operands."""
    from hi3d_hip import pack
    g = torch.Generator().manual_seed(11)
    N, H, W, C = 2, 16, 16, 320
    x = torch.randn((N * H * W, C), generator=g).to(torch.bfloat16)
    gam, bet = torch.randn(C, generator=g) * 0.2 + 1, torch.randn(C, generator=g) * 0.2
    y = ns.groupnorm_silu(x.to(dev), gam.to(dev), bet.to(dev), N, H * W, C, 1e-5, True)
    xr = x.float().reshape(N, H * W, C).transpose(1, 2)
    ref = F.silu(F.group_norm(xr, 32, gam, bet, 1e-5)).transpose(1, 2).reshape(-1, C)
    assert rel(y, ref) < 1.2e-2
    y = ns.layernorm(x.to(dev), gam.to(dev), bet.to(dev), 1e-5)
    assert rel(y, F.layer_norm(x.float(), (C,), gam, bet, 1e-5)) < 1.2e-2
    w = (torch.randn((640, C), generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(640, generator=g) * 0.1
    r = torch.randn((N * H * W, 640), generator=g).to(torch.bfloat16)
    y = ns.linear(x.to(dev), w.to(dev), b.to(dev), r.to(dev))
    assert rel(y, F.linear(x.float(), w.float(), b) + r.float()) < 1.2e-2
    assert rel(ns.linear(x.to(dev), w.to(dev), None, None), F.linear(x.float(), w.float())) < 1.2e-2
    cw = torch.randn((C, C, 3, 3), generator=g) * 0.02
    cb = torch.randn(C, generator=g) * 0.1
    wp = pack.pack_conv3x3(cw.to(dev))
    xi = x.float().reshape(N, H, W, C).permute(0, 3, 1, 2)
    cwb = cw.to(torch.bfloat16).float()
    for stride, up in ((1, False), (2, False), (1, True)):
        y = ns.conv3x3(x.to(dev), wp, cb.to(dev), N, H, W, stride, up, None)
        src = F.interpolate(xi, scale_factor=2, mode="nearest") if up else xi
        ref = F.conv2d(src, cwb, cb, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, C)
        assert y.shape == ref.shape and rel(y, ref) < 1.2e-2
    for Cf in (320, 640):                       # fused kernel at 320, two GEMMs at 640
        xf = torch.randn((512, Cf), generator=g).to(torch.bfloat16)
        w1 = torch.randn((8 * Cf, Cf), generator=g) * 0.04
        b1 = torch.randn(8 * Cf, generator=g) * 0.1
        w2 = (torch.randn((Cf, 4 * Cf), generator=g) * 0.03).to(torch.bfloat16)
        b2 = torch.randn(Cf, generator=g) * 0.1
        w1p, b1p = pack.pack_geglu(w1.to(dev), b1.to(dev))
        y = ns.ffn_geglu(xf.to(dev), w1p, b1p, w2.to(dev), b2.to(dev), xf.to(dev))
        h = F.linear(xf.float(), w1.to(torch.bfloat16).float(), b1)
        a, gt = h.chunk(2, dim=-1)
        ref = F.linear((a * F.gelu(gt)).to(torch.bfloat16).float(), w2.float(), b2) + xf.float()
        assert rel(y, ref) < 1.5e-2
