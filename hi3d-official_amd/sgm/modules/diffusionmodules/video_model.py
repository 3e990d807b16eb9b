"""VideoUNet for MI355X (reference: sgm/modules/diffusionmodules/video_model.py:84-501).

Same constructor keywords, forward signature and state_dict keys as the reference class,
so `target: sgm.modules.diffusionmodules.video_model.VideoUNet` in a Hi3D YAML and a Hi3D
checkpoint are drop-ins.  The module itself is only a parameter container; the forward
pass is hi3d_hip.runtime_unet.UNetRuntime (hand-written gfx950 kernels, one channels-last
bf16 layout, fused epilogues).  There is no eager/PyTorch fallback: without the HIP
library or on a CPU tensor, forward raises.
"""
from typing import List, Optional, Union

import torch
import torch.nn as nn

from ...util import ParamTree, params_key


def unet_param_shapes(cfg):
    """{state_dict key: shape} of the reference VideoUNet for this configuration."""
    from hi3d_hip.runtime_unet import unet_layout

    mc, te, ctx = cfg["model_channels"], 4 * cfg["model_channels"], cfg["context_dim"]
    S = {}

    def lin(p, o, i, bias=True):
        S[p + ".weight"] = (o, i)
        if bias:
            S[p + ".bias"] = (o,)

    def norm(p, c):
        S[p + ".weight"] = (c,); S[p + ".bias"] = (c,)

    def conv(p, o, i, *k):
        S[p + ".weight"] = (o, i) + tuple(k); S[p + ".bias"] = (o,)

    def res(p, cin, cout):
        norm(p + ".in_layers.0", cin); conv(p + ".in_layers.2", cout, cin, 3, 3)
        lin(p + ".emb_layers.1", cout, te)
        norm(p + ".out_layers.0", cout); conv(p + ".out_layers.3", cout, cout, 3, 3)
        if cin != cout:
            conv(p + ".skip_connection", cout, cin, 1, 1)
        q = p + ".time_stack"
        norm(q + ".in_layers.0", cout); conv(q + ".in_layers.2", cout, cout, 3, 1, 1)
        lin(q + ".emb_layers.1", cout, te)
        norm(q + ".out_layers.0", cout); conv(q + ".out_layers.3", cout, cout, 3, 1, 1)
        S[p + ".time_mixer.mix_factor"] = (1,)

    def attn(p, c, kv):
        lin(p + ".to_q", c, c, False); lin(p + ".to_k", c, kv, False); lin(p + ".to_v", c, kv, False)
        lin(p + ".to_out.0", c, c)

    def ff(p, c):
        lin(p + ".net.0.proj", 8 * c, c); lin(p + ".net.2", c, 4 * c)

    def transformer(p, c):
        norm(p + ".norm", c); lin(p + ".proj_in", c, c); lin(p + ".proj_out", c, c)
        b = p + ".transformer_blocks.0"
        attn(b + ".attn1", c, c); attn(b + ".attn2", c, ctx); ff(b + ".ff", c)
        for n in ("norm1", "norm2", "norm3"):
            norm(f"{b}.{n}", c)
        t = p + ".time_stack.0"
        norm(t + ".norm_in", c); ff(t + ".ff_in", c)
        attn(t + ".attn1", c, c); attn(t + ".attn2", c, ctx); ff(t + ".ff", c)
        for n in ("norm1", "norm2", "norm3"):
            norm(f"{t}.{n}", c)
        lin(p + ".time_pos_embed.0", 4 * c, c); lin(p + ".time_pos_embed.2", c, 4 * c)
        S[p + ".time_mixer.mix_factor"] = (1,)

    lin("time_embed.0", te, mc); lin("time_embed.2", te, te)
    lin("label_emb.0.0", te, cfg["adm_in_channels"]); lin("label_emb.0.2", te, te)
    blocks_in, middle, blocks_out = unet_layout(cfg)
    named = [(f"input_blocks.{i}", L) for i, L in enumerate(blocks_in)] + [("middle_block", middle)] + \
            [(f"output_blocks.{i}", L) for i, L in enumerate(blocks_out)]
    for base, layers in named:
        for j, L in enumerate(layers):
            p = f"{base}.{j}"
            if L[0] == "conv_in":
                conv(p, mc, cfg["in_channels"], 3, 3)
            elif L[0] == "res":
                res(p, L[1], L[2])
            elif L[0] == "attn":
                transformer(p, L[1])
            elif L[0] == "down":
                conv(p + ".op", L[1], L[1], 3, 3)
            elif L[0] == "up":
                conv(p + ".conv", L[1], L[1], 3, 3)
    norm("out.0", mc); conv("out.2", cfg["out_channels"], mc, 3, 3)
    return S


class VideoUNet(ParamTree):
    def __init__(
        self,
        in_channels: int,
        model_channels: int,
        out_channels: int,
        num_res_blocks: int,
        attention_resolutions: List[int],
        dropout: float = 0.0,
        channel_mult: List[int] = (1, 2, 4, 8),
        conv_resample: bool = True,
        dims: int = 2,
        num_classes: Optional[Union[int, str]] = None,
        use_checkpoint: bool = False,
        num_heads: int = -1,
        num_head_channels: int = -1,
        num_heads_upsample: int = -1,
        use_scale_shift_norm: bool = False,
        resblock_updown: bool = False,
        transformer_depth: Union[List[int], int] = 1,
        transformer_depth_middle: Optional[int] = None,
        context_dim: Optional[int] = None,
        time_downup: bool = False,
        time_context_dim: Optional[int] = None,
        extra_ff_mix_layer: bool = False,
        use_spatial_context: bool = False,
        merge_strategy: str = "fixed",
        merge_factor: float = 0.5,
        spatial_transformer_attn_type: str = "softmax",
        video_kernel_size: Union[int, List[int]] = 3,
        use_linear_in_transformer: bool = False,
        adm_in_channels: Optional[int] = None,
        disable_temporal_crossattention: bool = False,
        max_ddpm_temb_period: int = 10000,
    ):
        assert context_dim is not None
        # The gfx950 runtime implements the configuration family Hi3D ships
        # (configs/inference-v01.yaml:18-48 / inference-v02.yaml); anything else fails here,
        # loudly, instead of silently running different math.
        unsupported = []
        if dims != 2: unsupported.append("dims != 2")
        if num_classes != "sequential": unsupported.append("num_classes != 'sequential'")
        if num_head_channels != 64: unsupported.append("num_head_channels != 64")
        if use_scale_shift_norm: unsupported.append("use_scale_shift_norm")
        if resblock_updown: unsupported.append("resblock_updown")
        if not conv_resample: unsupported.append("conv_resample=False")
        if time_downup: unsupported.append("time_downup")
        if not extra_ff_mix_layer: unsupported.append("extra_ff_mix_layer=False")
        if not use_spatial_context: unsupported.append("use_spatial_context=False")
        if merge_strategy != "learned_with_images": unsupported.append(f"merge_strategy={merge_strategy}")
        if list(video_kernel_size) != [3, 1, 1] if not isinstance(video_kernel_size, int) else True:
            unsupported.append(f"video_kernel_size={video_kernel_size}")
        if not use_linear_in_transformer: unsupported.append("use_linear_in_transformer=False")
        if disable_temporal_crossattention: unsupported.append("disable_temporal_crossattention")
        if spatial_transformer_attn_type not in ("softmax", "softmax-xformers"):
            unsupported.append(f"attn type {spatial_transformer_attn_type}")
        depth = transformer_depth if isinstance(transformer_depth, int) else None
        if depth != 1 or (transformer_depth_middle not in (None, 1)):
            unsupported.append("transformer_depth != 1")
        if dropout != 0.0: unsupported.append("dropout")
        if unsupported:
            raise NotImplementedError("VideoUNet (MI355X runtime): unsupported options: " + ", ".join(unsupported))
        self.cfg = dict(in_channels=in_channels, model_channels=model_channels, out_channels=out_channels,
                        num_res_blocks=num_res_blocks, attention_resolutions=list(attention_resolutions),
                        channel_mult=list(channel_mult), num_head_channels=num_head_channels,
                        context_dim=context_dim, adm_in_channels=adm_in_channels, transformer_depth=1,
                        max_ddpm_temb_period=max_ddpm_temb_period)
        super().__init__(unet_param_shapes(self.cfg))
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_classes, self.use_checkpoint = num_classes, use_checkpoint
        self._runtime = None
        self._runtime_key = None

    # ------------------------------------------------------------------
    def runtime(self, device=None):
        """Pack the current parameters for the GPU (once; re-packed if they are replaced)."""
        from hi3d_hip.runtime_unet import UNetRuntime

        p0 = next(self.parameters())
        device = torch.device(device) if device is not None else p0.device
        if device.type != "cuda":
            raise RuntimeError("VideoUNet runs on the MI355X only: move the model or its inputs to cuda "
                               "(no CPU path exists in this framework)")
        key = params_key(self, device)
        if self._runtime is None or self._runtime_key != key:
            self._runtime = UNetRuntime(self.state_dict(), self.cfg, device)
            self._runtime_key = key
        return self._runtime

    def forward(
        self,
        x: torch.Tensor,
        timesteps: torch.Tensor,
        context: Optional[torch.Tensor] = None,
        y: Optional[torch.Tensor] = None,
        time_context: Optional[torch.Tensor] = None,
        num_video_frames: Optional[int] = None,
        image_only_indicator: Optional[torch.Tensor] = None,
    ):
        assert y is not None, "must specify y: the model is class-conditional (num_classes='sequential')"
        assert context is not None and num_video_frames is not None
        if time_context is not None:
            raise NotImplementedError("use_spatial_context=True ignores time_context; pass None")
        if image_only_indicator is None:
            image_only_indicator = torch.zeros(x.shape[0] // num_video_frames, num_video_frames, device=x.device)
        rt = self.runtime(x.device if x.is_cuda else None)
        out = rt.forward_nchw(x, timesteps, context, y, int(num_video_frames), image_only_indicator)
        return out.to(x.dtype)
