"""One-time weight re-layout: reference state_dict tensors (OIHW / OIDHW / [out,in],
fp32 or fp16) -> the K-major bf16 images the gfx950 GEMM consumes.

All functions are pure tensor reshuffles on whatever device the input lives on.
"""
import torch


def _bf16(w):
    return w.detach().to(torch.bfloat16).contiguous()


def pack_linear(w):
    """nn.Linear weight [N, K] is already K-major."""
    return _bf16(w)


def pack_conv1x1(w):
    """[O, I, 1, 1] -> [O, I]"""
    return _bf16(w.reshape(w.shape[0], w.shape[1]))


def pack_conv3x3(w, cin_pad=None, cout_pad=None):
    """OIHW [O, I, 3, 3] -> [O_pad, (ky, kx, I_pad)]: k = (ky*3+kx)*I_pad + ci."""
    O, I = w.shape[:2]
    Ip = I if cin_pad is None else cin_pad
    Op = O if cout_pad is None else cout_pad
    out = torch.zeros((Op, 3, 3, Ip), dtype=torch.bfloat16, device=w.device)
    out[:O, :, :, :I] = w.detach().permute(0, 2, 3, 1).to(torch.bfloat16)
    return out.reshape(Op, 9 * Ip).contiguous()


def pack_conv3x3_up_phases(w):
    """Upsample(2x nearest) + conv3x3 pad 1 (openaimodel.py:107-146) as four 2x2 convolutions on the LOW-resolution image, one
    per output phase (a, b) = (y & 1, x & 1): output pixel (2i + a, 2j + b) reads up-sampled rows 2i + a + ky - 1, i.e. source
    rows i - 1 (ky = 0), i, i (a = 0) or i, i, i + 1 (a = 1) -- two source rows, the weights of the coinciding taps summed
    (fp32, then bf16); columns alike.  In terms of the 3x3 pad-1 gather on the source image the phase reads taps
    (a + dy, b + dx), dy, dx in {0, 1}.  Returns [(W_ab [O, 4 * I] bf16 laid out [O][dy*2+dx][I], taps (ky*3+kx) x 4)] for
    (a, b) = (0,0), (0,1), (1,0), (1,1): 4/9 of the multiply-adds of the 3x3 conv on the up-sampled image."""
    O, I = w.shape[:2]
    return [(m.reshape(O, 4 * I).to(torch.bfloat16).contiguous(), taps) for m, taps in up_phase_filters(w)]


def up_phase_filters(w):
    """The arithmetic of pack_conv3x3_up_phases in the weights' own precision: [(m [O, 2, 2, I] (dy, dx), taps)] per phase
    (a, b) = (0,0), (0,1), (1,0), (1,1) (tests/test_host_cpu.py checks it against conv2d on the up-sampled image)."""
    O, I = w.shape[:2]
    wf = w.detach() if w.dtype == torch.float64 else w.detach().float()
    sets = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}          # phase -> original taps merged into (d = 0, d = 1)
    out = []
    for a in (0, 1):
        for b in (0, 1):
            m = torch.zeros((O, 2, 2, I), dtype=wf.dtype, device=w.device)
            for dy in (0, 1):
                for dx in (0, 1):
                    for ky in sets[a][dy]:
                        for kx in sets[b][dx]:
                            m[:, dy, dx, :] += wf[:, :, ky, kx]
            out.append((m, tuple((a + dy) * 3 + (b + dx) for dy in (0, 1) for dx in (0, 1))))
    return out


def pack_convt3(w):
    """Conv3d (3,1,1) weight [O, I, 3, 1, 1] -> [O, (kt, I)]."""
    O, I = w.shape[:2]
    return _bf16(w.reshape(O, I, 3).permute(0, 2, 1).reshape(O, 3 * I))


def pack_geglu(w, b):
    """GEGLU proj [2*inner, K] (first half = value, second half = gate,
    sgm/modules/attention.py:92-94) -> rows ordered [x0, x1, g0, g1] per 4 rows so the
    MFMA epilogue finds value and gate of the same output column in one lane."""
    two_inner, K = w.shape
    inner = two_inner // 2
    assert inner % 2 == 0
    wx, wg = w[:inner].reshape(inner // 2, 2, K), w[inner:].reshape(inner // 2, 2, K)
    wp = torch.cat([wx, wg], dim=1).reshape(two_inner, K)
    bx, bg = b[:inner].reshape(inner // 2, 2), b[inner:].reshape(inner // 2, 2)
    bp = torch.cat([bx, bg], dim=1).reshape(two_inner)
    return _bf16(wp), bp.detach().to(torch.float32).contiguous()


def pack_qkv(wq, wk, wv, q_scale=None):
    """to_q / to_k / to_v (bias-free, attention.py:272-274) -> one [3C, C] matrix.
    q_scale: factor folded into the to_q rows in fp32 before the (single) bf16 rounding -- the
    spatial attention kernel then takes q as exp2-ready (softmax scale * log2 e)."""
    if q_scale is not None:
        wq = wq.detach().to(torch.float32) * q_scale
    return _bf16(torch.cat([wq.to(torch.float32), wk.detach().to(torch.float32), wv.detach().to(torch.float32)], dim=0))


def f32(t):
    return t.detach().to(torch.float32).contiguous()


def pad_vec(b, n):
    out = torch.zeros((n,), dtype=torch.float32, device=b.device)
    out[: b.numel()] = b.detach().to(torch.float32)
    return out
