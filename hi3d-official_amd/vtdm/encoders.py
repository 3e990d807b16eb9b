"""vtdm.encoders mirror (reference: vtdm/encoders.py): the conditioner embedders Hi3D adds to sgm.

Built: `AesEmbedder` (:56-91) -- OpenAI CLIP ViT-L/14 image features of the clip's middle frame on the gfx950 ViT
runtime, L2-normalised, through the 5-layer aesthetic MLP (tools/aes_score.py:14-33), concatenated with a 255-wide
sinusoidal embedding of 100 x score.  Not built: `DepthEmbedder` (:15-53, MiDaS DPT-hybrid: a ResNet-50 + ViT-B hybrid
backbone with a DPT decoder, `annotator/midas` + timm; once per clip, v02 only) -- it raises by name.
"""
import torch
import torch.nn as nn

from sgm.modules.encoders.modules import CLIP_MEAN, CLIP_STD, CLIP_VISUAL_ARCHS, AbstractEmbModel, _ClipVisualTower
from sgm.util import ParamTree


class AesEmbedder(AbstractEmbModel):
    """state_dict names follow the reference module tree: `aesthetic_model.visual.*` (clip.load's model) and
    `aesthetic_mlp.layers.{0,2,4,6,7}.{weight,bias}` (tools/aes_score.MLP: Linear 768-1024-128-64-16-1, dropouts between)."""

    MLP_DIMS = ((0, 768, 1024), (2, 1024, 128), (4, 128, 64), (6, 64, 16), (7, 16, 1))

    def __init__(self, freeze=True, arch="ViT-L-14"):
        """arch: not in the reference signature (it hard-wires clip.load("ckpts/ViT-L-14.pt")); tests pass a reduced tower."""
        super().__init__()
        if CLIP_VISUAL_ARCHS[arch]["out_dim"] != 768:
            raise ValueError("AesEmbedder: the aesthetic MLP takes 768-wide CLIP image features")
        self.aesthetic_model = _ClipVisualTower(CLIP_VISUAL_ARCHS[arch])
        self.aesthetic_mlp = ParamTree({f"layers.{i}.{n}": (s if n == "bias" else (o, k))
                                        for i, k, o in self.MLP_DIMS for n, s in (("weight", None), ("bias", (o,)))})
        self._mlp = None
        if freeze:
            for p in self.parameters():
                p.requires_grad = False

    @staticmethod
    def _split(t, kp):
        """fp32 [R, K] -> bf16 [R, 3*kp] = [hi | hi | lo] (hi = bf16(t), lo = bf16(t - hi)), K zero-padded to kp."""
        hi = t.to(torch.bfloat16)
        lo = (t - hi.float()).to(torch.bfloat16)
        out = torch.zeros((t.shape[0], 3 * kp), device=t.device, dtype=torch.bfloat16)
        k = t.shape[1]
        out[:, :k], out[:, kp:kp + k], out[:, 2 * kp:2 * kp + k] = hi, hi, lo
        return out

    def _packed_mlp(self, dev):
        """The aesthetic MLP has to stay fp32-accurate (its output x 100 goes through a sinusoidal embedding): every
        Linear runs as ONE bf16 GEMM over a tripled K -- activations [x_hi | x_hi | x_lo] against weights
        [w_hi | w_lo | w_hi], fp32 accumulate -- i.e. x w = x_hi w_hi + x_hi w_lo + x_lo w_hi, error ~2^-16."""
        from sgm.util import params_key
        key = params_key(self.aesthetic_mlp, dev)
        if self._mlp is None or self._mlp[0] != key:
            sd, out = self.aesthetic_mlp.state_dict(), []
            for i, k, o in self.MLP_DIMS:
                kp, op = (k + 63) // 64 * 64, (o + 3) // 4 * 4
                w = torch.zeros((op, k), device=dev)
                w[:o] = sd[f"layers.{i}.weight"].to(dev).float()
                hi = w.to(torch.bfloat16)
                lo = (w - hi.float()).to(torch.bfloat16)
                wp = torch.zeros((op, 3 * kp), device=dev, dtype=torch.bfloat16)
                wp[:, :k], wp[:, kp:kp + k], wp[:, 2 * kp:2 * kp + k] = hi, lo, hi
                b = torch.zeros((op,), device=dev)
                b[:o] = sd[f"layers.{i}.bias"].to(dev).float()
                out.append((wp.contiguous(), b, o, op, kp))
            self._mlp = (key, out)
        return self._mlp[1]

    @torch.no_grad()
    def forward(self, x):
        from hi3d_hip import ops
        B, C, T, H, W = x.shape
        dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
        y = x[:, :, T // 2].to(dev).float()
        y = torch.nn.functional.interpolate(y, [224, 384], mode="bilinear")[:, :, :, 80:304]
        y = (y + 1) * 0.5
        y = (y - torch.tensor(CLIP_MEAN, device=dev).view(1, 3, 1, 1)) / torch.tensor(CLIP_STD, device=dev).view(1, 3, 1, 1)
        with torch.cuda.device(dev):
            h = self.aesthetic_model.runtime(dev).forward(y.contiguous())     # [B, 768] fp32
            ops.l2_normalize_rows_(h)
            for wp, b, o, op, kp in self._packed_mlp(dev):
                h = ops.gemm(self._split(h, kp), wp, M=B, N=op, K=3 * kp, bias=b, out_fp32=True)[:, :o].contiguous()
            emb = ops.timestep_embedding(h[:, 0] * 100, 255)
        return torch.cat([h, emb], dim=1)


class DepthEmbedder(AbstractEmbModel):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError(
            "vtdm.encoders.DepthEmbedder (MiDaS DPT-hybrid depth, v02 conditioner, once per clip) is not part of the "
            "MI355X hot-path framework: feed a precomputed depth `concat` (9 x h x w per frame, 3 x 3 pixel-unshuffled, "
            "min-max normalised: vtdm/encoders.py:36-50) or run the reference embedder once per clip")
