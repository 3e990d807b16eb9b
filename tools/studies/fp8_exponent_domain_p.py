"""CPU study: fp8 attention P built in the exponent domain (byte = trunc(8 (s + 7) + c)) vs RNE e4m3 of exp2(s), with the row sum
taken (a) exactly (today's kernel) or (b) from the quantised P itself (ones-row MFMA).  Model of the kernel: one reference point per
row = the row max (the kernel's m_ref is a running stale max: P = exp2(s - m + 3)), V quantised e4m3 per (d row, 64-key tile) with a
power-of-two scale, fp32 accumulate."""
import torch
torch.manual_seed(0)
F8 = torch.float8_e4m3fn

def q_v(v):                       # V^T e4m3 per (64-key tile, d): shared exponent floor(log2 amax) - 7
    S, d = v.shape
    vb = v.reshape(S // 64, 64, d)
    amax = vb.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    e = torch.floor(torch.log2(amax)) - 7
    s = torch.exp2(e)
    return ((vb / s).to(F8).float() * s).reshape(S, d)

def decode_byte(b):               # e4m3 value of byte b in [0, 126]
    E, M = (b // 8).float(), (b % 8).float()
    return torch.where(b < 8, M / 8 * 2.0 ** -6, torch.exp2(E - 7) * (1 + M / 8))

def run(temp, S=8192, d=64, rows=512):
    q, k, v = torch.randn(rows, d) * temp, torch.randn(S, d), torch.randn(S, d)
    sc = ((q.double() @ k.double().T) * d ** -0.5 * 1.4426950408889634).float()     # log2 domain
    ref = (torch.softmax(sc.double() * 0.6931471805599453, dim=1) @ v.double())
    m = sc.amax(dim=1, keepdim=True)
    # stale reference point: the kernel's m_ref lags the true max by up to a few units; model with m - delta, delta in [0, 2]
    out = {}
    vq = q_v(v)
    for delta in (0.0, 1.5):
        s = sc - (m - delta) + 3.0                    # P = exp2(s): max P = 8 * 2^delta
        p = torch.exp2(s)
        p_rne = p.to(F8).float()
        out[f"RNE, exact sum         (delta {delta})"] = (p_rne @ vq) / p.sum(1, keepdim=True)
        out[f"RNE, sum of quantised  (delta {delta})"] = (p_rne @ vq) / p_rne.sum(1, keepdim=True)
        for c in (0.04, 0.5):
            b = torch.clamp(torch.trunc(8 * s + 56 + c), 0, 126).long()
            pe = decode_byte(b)
            out[f"exp-domain c={c:4.2f}, sum of quantised (delta {delta})"] = (pe @ vq) / pe.sum(1, keepdim=True)
    for name, o in out.items():
        o = o.double()
        rms = ((o - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        cos = torch.nn.functional.cosine_similarity(o.flatten(), ref.flatten(), dim=0).item()
        print(f"  logit std {temp:3.1f}  {name:52s} rms {rms:.4f}  cos {cos:.5f}")

for temp in (1.0, 3.0, 8.0):
    run(temp)
