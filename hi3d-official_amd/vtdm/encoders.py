"""vtdm.encoders mirror (reference: vtdm/encoders.py): the conditioner embedders Hi3D adds to sgm.

`AesEmbedder` (:56-91) -- OpenAI CLIP ViT-L/14 image features of the clip's middle frame on the gfx950 ViT
runtime, L2-normalised, through the 5-layer aesthetic MLP (tools/aes_score.py:14-33), concatenated with a 255-wide
sinusoidal embedding of 100 x score.  `DepthEmbedder` (:15-53) -- MiDaS DPT-hybrid (a BiT ResNet-50 + ViT-B hybrid
backbone with a DPT decoder: `annotator/midas` + timm in the reference) on hi3d_hip/runtime_dpt.py; once per clip.
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
        mean, std = torch.tensor(CLIP_MEAN, device=dev), torch.tensor(CLIP_STD, device=dev)
        with torch.cuda.device(dev):
            # F.interpolate(y, [224, 384], "bilinear")[..., 80:304] -> (y + 1) / 2 -> CLIP mean / std (reference :80-83):
            # two banded resampling passes, the crop in the column table, the affine fused into the second pass
            y = ops.resample_image(y, "aes", scale=(0.5 / std).contiguous(), shift=((0.5 - mean) / std).contiguous())
            h = self.aesthetic_model.runtime(dev).forward(y.contiguous())     # [B, 768] fp32
            ops.l2_normalize_rows_(h)
            for wp, b, o, op, kp in self._packed_mlp(dev):
                h = ops.gemm(self._split(h, kp), wp, M=B, N=op, K=3 * kp, bias=b, out_fp32=True)[:, :o].contiguous()
            emb = ops.timestep_embedding(h[:, 0] * 100, 255)
        return torch.cat([h, emb], dim=1)


class _MidasInference(nn.Module):
    """`MiDaSInference` (annotator/midas/api.py:145-165): `.model` is the DPTDepthModel -- here its parameters under the
    reference's names (`model.pretrained.model.*`, `model.pretrained.act_postprocess*`, `model.scratch.*`) and the packed
    gfx950 runtime, rebuilt when a parameter changes."""

    def __init__(self):
        super().__init__()
        from hi3d_hip.runtime_dpt import dpt_hybrid_shapes
        self.model = ParamTree(dpt_hybrid_shapes())
        self._rt = None

    def runtime(self, device):
        from hi3d_hip.runtime_dpt import DPTHybridRuntime
        from sgm.util import params_key
        key = params_key(self, device)
        if self._rt is None or self._rt[0] != key:
            self._rt = (key, DPTHybridRuntime(self.state_dict(), "model.", device))
        return self._rt[1]


class DepthEmbedder(AbstractEmbModel):
    """vtdm/encoders.py:15-53: MiDaS DPT-hybrid inverse depth of every conditioning frame at 1 / 2.6666 of its size,
    resampled to 3/8 of the frame, min-max normalised per frame and 3 x 3 pixel-unshuffled: [(b t), 9, H/8, W/8] (the
    `concat` conditioning of stage 2), or [b, 9, t, H/8, W/8] with `use_3d`.  The reference loads
    "ckpts/dpt_hybrid_384.pt" in its constructor; here the parameters are a ParamTree under the same names
    (`model.model.*`): `init_from_midas_ckpt(path)` or the Hi3D checkpoint's `conditioner.embedders.*` entries fill them."""

    def __init__(self, freeze=True, use_3d=False, shuffle_size=3, scale_factor=2.6666):
        super().__init__()
        self.model = _MidasInference()
        self.use_3d, self.shuffle_size, self.scale_factor = use_3d, shuffle_size, scale_factor
        if freeze:
            self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for p in self.parameters():
            p.requires_grad = False

    def init_from_midas_ckpt(self, path):
        """dpt_hybrid_384.pt / dpt_hybrid-midas-501f0c75.pt: a DPTDepthModel state_dict (optionally under 'model')."""
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = sd.get("model", sd) if isinstance(sd, dict) and "model" in sd and not torch.is_tensor(sd["model"]) else sd
        missing, unexpected = self.model.model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
        if missing:
            raise KeyError(f"MiDaS checkpoint lacks {len(missing)} keys, e.g. {missing[:3]}")

    @torch.no_grad()
    def forward(self, x):
        T = 16                                                   # (hard-wired in the reference: vtdm/encoders.py:34)
        if x.dim() == 4:
            if x.shape[0] % T:
                raise ValueError(f"DepthEmbedder: {x.shape[0]} frames are not a multiple of t = {T}")
            y = x
        else:
            B, C, T, H, W = x.shape
            y = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
        dev = y.device if y.is_cuda else torch.device("cuda", torch.cuda.current_device())
        out = self.model.runtime(dev).depth_embed(y, self.shuffle_size, self.scale_factor)
        if self.use_3d:
            out = out.reshape(-1, T, *out.shape[1:]).permute(0, 2, 1, 3, 4)
        return out
