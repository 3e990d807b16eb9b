"""CLIP vision tower on the gfx950 kernels: the conditioner's image encoders (once per clip).

Reference: `FrozenOpenCLIPImageEmbedder` builds open_clip's ViT-H-14 visual tower
(sgm/modules/encoders/modules.py:592-596) and calls `self.model.visual(img)` (:700-704); `AesEmbedder` calls OpenAI
CLIP ViT-L/14 `encode_image` (vtdm/encoders.py:59,86).  Both are the same pre-LN VisionTransformer (patch conv,
class token, learned positions, ln_pre, N x [ln_1, MHA, ln_2, MLP], ln_post on the class token, projection); they
differ in width / heads / depth and the MLP activation (GELU vs QuickGELU).  Parameters keep THEIR state_dict
names (`visual.conv1.weight`, `visual.transformer.resblocks.{i}.attn.in_proj_weight`, ...), so an open_clip /
OpenAI checkpoint loads unchanged.

Everything runs through the C ABI: the patch convolution is a GEMM over unfolded patches (stride = kernel: no
overlap, the unfold is a pure re-layout), LayerNorm, fused-QKV GEMM, attention, projections with the residual in the
GEMM epilogue, `hi3d_act_bf16` for the MLP activation.  Head dim 64 (ViT-L/14) uses the flash kernel; head dim 80
(ViT-H/14) pads every head to 128 columns at pack time (zero weight rows / columns: exact) and runs
scores GEMM -> row softmax -> P V GEMM per (image, head), as the VAE mid-block attention does -- 257 tokens, once per clip.
"""
import torch

from . import ops
from .pack import _bf16


class ViTRuntime:
    def __init__(self, sd, prefix, heads, act, device):
        """sd: {name: tensor} holding `prefix + <open_clip visual key>`; act: 'gelu' | 'quick_gelu'."""
        g = lambda k: sd[prefix + k].detach().to(device=device, dtype=torch.float32)
        self.dev, self.heads, self.act = torch.device(device), heads, act
        w = g("conv1.weight")
        self.width, self.patch = w.shape[0], w.shape[-1]
        Wd, H = self.width, heads
        if Wd % H or Wd % 8:
            raise ops._l.Hi3dError("ViT width must be a multiple of heads and of 8")
        self.d = Wd // H
        self.dp = (self.d + 63) // 64 * 64                  # head dim padded to the GEMM's K granule
        self.k0 = 3 * self.patch * self.patch
        self.k0p = (self.k0 + 63) // 64 * 64
        wc = torch.zeros((Wd, self.k0p), device=device)
        wc[:, :self.k0] = w.reshape(Wd, self.k0)
        self.W = {"conv": _bf16(wc)}
        pos = g("positional_embedding")
        self.tokens = pos.shape[0]
        self.grid = int(round((self.tokens - 1) ** 0.5))
        self.W["cls_pos"] = _bf16(g("class_embedding") + pos[0])            # the class row never changes
        self.W["pos"] = _bf16(pos[1:])
        for n in ("ln_pre", "ln_post"):
            self.W[n + ".g"], self.W[n + ".b"] = g(n + ".weight").contiguous(), g(n + ".bias").contiguous()
        self.out_dim = g("proj").shape[1]
        self.W["proj"] = _bf16(g("proj").t())                                # [out, W], K-major
        self.layers = 0
        while (prefix + f"transformer.resblocks.{self.layers}.ln_1.weight") in sd:
            i = self.layers
            p = f"transformer.resblocks.{i}."
            for n in ("ln_1", "ln_2"):
                self.W[f"{i}.{n}.g"], self.W[f"{i}.{n}.b"] = g(p + n + ".weight").contiguous(), g(p + n + ".bias").contiguous()
            wi, bi = g(p + "attn.in_proj_weight"), g(p + "attn.in_proj_bias")
            wo = g(p + "attn.out_proj.weight")
            if self.dp == self.d:
                self.W[f"{i}.qkv.w"], self.W[f"{i}.qkv.b"] = _bf16(wi), bi.contiguous()
                self.W[f"{i}.o.w"] = _bf16(wo)
            else:                                                              # heads padded d -> dp with zeros
                wq = torch.zeros((3, H, self.dp, Wd), device=device)
                wq[:, :, :self.d] = wi.reshape(3, H, self.d, Wd)
                bq = torch.zeros((3, H, self.dp), device=device)
                bq[:, :, :self.d] = bi.reshape(3, H, self.d)
                self.W[f"{i}.qkv.w"], self.W[f"{i}.qkv.b"] = _bf16(wq.reshape(3 * H * self.dp, Wd)), bq.reshape(-1).contiguous()
                wop = torch.zeros((Wd, H, self.dp), device=device)
                wop[:, :, :self.d] = wo.reshape(Wd, H, self.d)
                self.W[f"{i}.o.w"] = _bf16(wop.reshape(Wd, H * self.dp))
            self.W[f"{i}.o.b"] = g(p + "attn.out_proj.bias").contiguous()
            self.W[f"{i}.fc.w"], self.W[f"{i}.fc.b"] = _bf16(g(p + "mlp.c_fc.weight")), g(p + "mlp.c_fc.bias").contiguous()
            self.W[f"{i}.pj.w"], self.W[f"{i}.pj.b"] = _bf16(g(p + "mlp.c_proj.weight")), g(p + "mlp.c_proj.bias").contiguous()
            self.layers += 1
        if not self.layers:
            raise KeyError(f"no transformer.resblocks under '{prefix}'")

    # ------------------------------------------------------------------
    def _attention(self, qkv, B, S):
        H, d, dp = self.heads, self.d, self.dp
        if dp == 64:
            return ops.self_attention_fused_qkv(qkv, B, S, H, scale=d ** -0.5)
        HD = H * dp
        S4 = (S + 3) // 4 * 4                                  # the GEMM wants N % 4 == 0: up to 3 key rows of padding
        vt = ops.transpose_v(qkv[:, 2 * HD:], B, HD // 64, S, 3 * HD)           # [B, HD/64, 64, S_pad] == V^T [B][HD][S_pad]
        S_pad = vt.shape[-1]
        vt = vt.reshape(B, HD, S_pad)
        o = torch.empty((B * S, HD), device=qkv.device, dtype=torch.bfloat16)
        for b in range(B):
            rows = qkv[b * S:]
            for h in range(H):
                q, k = rows[:, h * dp:], rows[:, HD + h * dp:]
                sc = ops.gemm(q, k, M=S, N=S4, K=dp, lda=3 * HD, ldw=3 * HD, out_fp32=True)       # [S, S4] fp32
                pr = ops.softmax_rows(sc, S, S, S_pad, d ** -0.5)                                   # keys >= S: zeros
                ops.gemm(pr, vt[b, h * dp:(h + 1) * dp], M=S, N=dp, K=S_pad, lda=S_pad, ldw=S_pad,
                         out=o[b * S:(b + 1) * S, h * dp:(h + 1) * dp])
        return o

    @torch.no_grad()
    def forward(self, img):
        """img: [B, 3, H, W] fp32, CLIP-normalised, H = W = patch * grid -> fp32 [B, out_dim]."""
        W_, Wd, P, G = self.W, self.width, self.patch, self.grid
        B = img.shape[0]
        if tuple(img.shape[1:]) != (3, P * G, P * G):
            raise ops._l.Hi3dError(f"ViT input must be [B,3,{P * G},{P * G}], got {tuple(img.shape)}")
        with torch.cuda.device(self.dev):
            S, T = self.tokens, G * G
            # unfold (kernel = stride: a re-layout) -> [B*T, 3*P*P] in conv1.weight's (c, ky, kx) order, K padded to 64
            pt = torch.zeros((B * T, self.k0p), device=self.dev, dtype=torch.bfloat16)
            pt[:, :self.k0] = img.to(self.dev, torch.float32).reshape(B, 3, G, P, G, P).permute(0, 2, 4, 1, 3, 5).reshape(B * T, self.k0)
            # +3 rows: the padded-head attention reads up to 3 key rows past the last image (ignored columns)
            x = torch.zeros((B * S + 3, Wd), device=self.dev, dtype=torch.bfloat16)[:B * S]
            for b in range(B):
                x[b * S] = W_["cls_pos"]
                ops.gemm(pt[b * T:(b + 1) * T], W_["conv"], M=T, N=Wd, K=self.k0p, R1=W_["pos"], out=x[b * S + 1:(b + 1) * S])
            x = ops.layernorm(x, W_["ln_pre.g"], W_["ln_pre.b"], B * S, Wd)
            nq = 3 * self.heads * self.dp
            for i in range(self.layers):
                n = ops.layernorm(x, W_[f"{i}.ln_1.g"], W_[f"{i}.ln_1.b"], B * S, Wd)
                qkv = torch.zeros((B * S + 3, nq), device=self.dev, dtype=torch.bfloat16)[:B * S]
                ops.gemm(n, W_[f"{i}.qkv.w"], M=B * S, N=nq, K=Wd, bias=W_[f"{i}.qkv.b"], out=qkv)
                a = self._attention(qkv, B, S)
                x = ops.gemm(a, W_[f"{i}.o.w"], M=B * S, N=Wd, K=a.shape[1], bias=W_[f"{i}.o.b"], R1=x)
                n = ops.layernorm(x, W_[f"{i}.ln_2.g"], W_[f"{i}.ln_2.b"], B * S, Wd)
                h = ops.gemm(n, W_[f"{i}.fc.w"], M=B * S, N=4 * Wd, K=Wd, bias=W_[f"{i}.fc.b"])
                ops.act_(h, self.act)
                x = ops.gemm(h, W_[f"{i}.pj.w"], M=B * S, N=Wd, K=4 * Wd, bias=W_[f"{i}.pj.b"], R1=x)
            cls = x.reshape(B, S, Wd)[:, 0].contiguous()                     # the class token of every image
            cls = ops.layernorm(cls, W_["ln_post.g"], W_["ln_post.b"], B, Wd)
            return ops.gemm(cls, W_["proj"], M=B, N=self.out_dim, K=Wd, out_fp32=True)
