"""CPU study (round 2, for the next one): can P.V of the flash attention run on the fp8 matrix path inside the attention
parity tolerance (2e-2 of the output range)?  Softmax over 4096 keys, d = 64; P quantised to e4m3 with a power-of-two
scale per (row, 32-key block) -- what an MX e8m0 block scale gives --, V to e4m3 with a scale per (32-key block, d);
fp32 accumulate, row sums from the fp32 P as the kernel keeps them.  A single MFMA cannot mix fp8 P with bf16 V, so the
'fp8 P, bf16 V' rows only separate the two error sources.   usage: python tools/studies/fp8_pv_error.py"""
import torch

torch.manual_seed(0)
F8 = torch.float8_e4m3fn


def q_e4m3(x, block_dim):
    shp = x.shape
    xb = x.unflatten(block_dim, (-1, 32))
    amax = xb.abs().amax(dim=block_dim + 1, keepdim=True).clamp_min(1e-30)
    s = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
    return ((xb / s).to(F8).float() * s).reshape(shp)


def attn(q, k, v, mode):
    d = q.shape[1]
    sc = (q @ k.T) * d ** -0.5
    p = torch.exp(sc - sc.amax(dim=1, keepdim=True))
    if mode == "bf16":
        pq, vq = p.bfloat16().float(), v.bfloat16().float()
    elif mode == "fp8 P and V (MX block scales)":
        pq, vq = q_e4m3(p, 1), q_e4m3(v, 0)
    else:
        pq, vq = q_e4m3(p, 1), v.bfloat16().float()
    return (pq @ vq) / p.sum(dim=1, keepdim=True)


for name, temp in (("diffuse attention (logit std 1)", 1.0), ("typical (logit std 3)", 3.0), ("peaked (logit std 8)", 8.0)):
    S, d = 4096, 64
    q, k, v = torch.randn(S, d) * temp, torch.randn(S, d), torch.randn(S, d)
    ref = torch.softmax((q.double() @ k.double().T) * d ** -0.5, dim=1) @ v.double()
    for mode in ("bf16", "fp8 P and V (MX block scales)", "fp8 P (MX), bf16 V"):
        o = attn(q, k, v, mode).double()
        rel = ((o - ref).abs().max() / ref.abs().max()).item()
        rms = ((o - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        print(f"{name:34s} {mode:32s} max-rel {rel:.4f}  rms-rel {rms:.4f}")
