"""First-stage autoencoder container (reference: sgm/models/autoencoder.py:437-520).

Holds the reference's parameter names (`encoder.*`, `decoder.*`, `quant_conv`,
`post_quant_conv`); `decode` runs hi3d_hip.runtime_vae.VAEDecoderRuntime (gfx950 kernels).
`encode` runs VAEEncoderRuntime (stage-2 pre-loop: once per frame, once per clip)."""
import torch

from ..util import ParamTree, params_key


def _shape_helpers(S, vks=(3, 1, 1)):
    def conv(p, o, i, *k):
        S[p + ".weight"] = (o, i) + tuple(k); S[p + ".bias"] = (o,)

    def norm(p, c):
        S[p + ".weight"] = (c,); S[p + ".bias"] = (c,)

    def resnet(p, cin, cout, temporal=False):
        norm(p + ".norm1", cin); conv(p + ".conv1", cout, cin, 3, 3)
        norm(p + ".norm2", cout); conv(p + ".conv2", cout, cout, 3, 3)
        if cin != cout:
            conv(p + ".nin_shortcut", cout, cin, 1, 1)
        if temporal:        # VideoResBlock (temporal_ae.py:18-81): ResBlock(dims=3, skip_t_emb) + mix_factor
            q = p + ".time_stack"
            norm(q + ".in_layers.0", cout); conv(q + ".in_layers.2", cout, cout, *vks)
            norm(q + ".out_layers.0", cout); conv(q + ".out_layers.3", cout, cout, *vks)
            S[p + ".mix_factor"] = (1,)

    def mid(p, c, temporal=False):
        resnet(p + ".block_1", c, c, temporal)
        norm(p + ".attn_1.norm", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(f"{p}.attn_1.{n}", c, c, 1, 1)
        resnet(p + ".block_2", c, c, temporal)
    return conv, norm, resnet, mid


def _check_dd(dd):
    if dd.get("attn_resolutions"):
        raise NotImplementedError("attn_resolutions must be empty (only the mid-block attention exists in Hi3D)")
    if dd.get("attn_type", "vanilla") not in ("vanilla", "vanilla-xformers"):
        raise NotImplementedError(f"attn_type {dd.get('attn_type')}")


def encoder_param_shapes(dd, prefix="encoder."):
    """Encoder (reference sgm/modules/diffusionmodules/model.py:487-575)."""
    _check_dd(dd)
    ch, mult, nres, zc = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"], dd["z_channels"]
    S = {}
    conv, norm, resnet, mid = _shape_helpers(S)
    conv(prefix + "conv_in", ch, dd["in_channels"], 3, 3)
    in_mult = [1] + mult
    for lvl in range(len(mult)):
        cin, cout = ch * in_mult[lvl], ch * mult[lvl]
        for b in range(nres):
            resnet(f"{prefix}down.{lvl}.block.{b}", cin, cout); cin = cout
        if lvl != len(mult) - 1:
            conv(f"{prefix}down.{lvl}.downsample.conv", cout, cout, 3, 3)
    top = ch * mult[-1]
    mid(prefix + "mid", top)
    norm(prefix + "norm_out", top)
    conv(prefix + "conv_out", 2 * zc if dd.get("double_z", True) else zc, top, 3, 3)
    return S


def decoder_param_shapes(dd, prefix="decoder.", temporal=False, vks=(3, 1, 1)):
    """Decoder (model.py:604-714) / VideoDecoder 'conv-only' (temporal_ae.py:293-349); vks = its video_kernel_size as a
    (kt, ky, kx) triple: (3, 1, 1) as SVD / Hi3D configure it, (3, 3, 3) for the class default `video_kernel_size=3`."""
    _check_dd(dd)
    ch, mult, nres, zc = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"], dd["z_channels"]
    S = {}
    conv, norm, resnet, mid = _shape_helpers(S, tuple(vks))
    top = ch * mult[-1]
    conv(prefix + "conv_in", top, zc, 3, 3)
    mid(prefix + "mid", top, temporal)
    cin = top
    for lvl in reversed(range(len(mult))):
        cout = ch * mult[lvl]
        for b in range(nres + 1):
            resnet(f"{prefix}up.{lvl}.block.{b}", cin, cout, temporal); cin = cout
        if lvl != 0:
            conv(f"{prefix}up.{lvl}.upsample.conv", cout, cout, 3, 3)
    norm(prefix + "norm_out", cin)
    conv(prefix + "conv_out", dd["out_ch"], cin, 3, 3)
    if temporal:                                  # AE3DConv.time_mix_conv (temporal_ae.py:84-107)
        conv(prefix + "conv_out.time_mix_conv", dd["out_ch"], dd["out_ch"], *vks)
    return S


def vae_param_shapes(dd, embed_dim):
    zc, dz = dd["z_channels"], 1 + bool(dd.get("double_z", True))
    S = {}
    S.update(encoder_param_shapes(dd)); S.update(decoder_param_shapes(dd))
    S["quant_conv.weight"] = (dz * embed_dim, dz * zc, 1, 1); S["quant_conv.bias"] = (dz * embed_dim,)
    S["post_quant_conv.weight"] = (zc, embed_dim, 1, 1); S["post_quant_conv.bias"] = (zc,)
    return S


class AutoencoderKL(ParamTree):
    is_video_decoder = False

    def __init__(self, embed_dim=None, ddconfig=None, lossconfig=None, monitor=None, ckpt_path=None, **kwargs):
        if ddconfig is None:
            raise ValueError("ddconfig is required")
        if ddconfig.get("attn_type", "vanilla") not in ("vanilla", "vanilla-xformers"):
            raise NotImplementedError(f"attn_type {ddconfig.get('attn_type')}")
        self.ddconfig, self.embed_dim = dict(ddconfig), embed_dim
        super().__init__(vae_param_shapes(self.ddconfig, embed_dim))
        self._runtime, self._runtime_key = None, None
        if ckpt_path is not None:
            from safetensors.torch import load_file
            sd = load_file(ckpt_path) if ckpt_path.endswith("safetensors") else torch.load(ckpt_path, map_location="cpu")
            self.load_state_dict(sd.get("state_dict", sd), strict=False)

    def runtime(self, device):
        from hi3d_hip.runtime_vae import VAEDecoderRuntime
        key = params_key(self, device)
        if self._runtime is None or self._runtime_key != key:
            self._runtime = VAEDecoderRuntime(self.state_dict(), self.ddconfig, device)
            self._runtime_key = key
        return self._runtime

    @torch.no_grad()
    def decode(self, z, **kwargs):
        """z: [N, 4, h, w] latents ALREADY divided by scale_factor -> images [N, 3, 8h, 8w]."""
        if not z.is_cuda:
            raise RuntimeError("AutoencoderKL.decode runs on the MI355X only (no CPU path in this framework)")
        return self.runtime(z.device).decode(z).to(z.dtype)

    sample_posterior = True      # DiagonalGaussianRegularizer(sample=True), models/autoencoder.py:508-520

    def encoder_runtime(self, device):
        from hi3d_hip.runtime_vae import VAEEncoderRuntime
        key = params_key(self, device)
        if getattr(self, "_enc_runtime", None) is None or self._enc_key != key:
            self._enc_runtime = VAEEncoderRuntime(self.state_dict(), self.ddconfig, device)
            self._enc_key = key
        return self._enc_runtime

    @torch.no_grad()
    def encode(self, x, return_reg_log=False, noise=None, **kwargs):
        """x: [N, 3, H, W] -> z [N, 4, H/8, W/8].  Like the reference, the posterior is SAMPLED
        with noise drawn on the CPU generator (distributions.py:37-41) unless `noise` is given;
        AutoencoderKLModeOnly returns the mode."""
        if not x.is_cuda:
            raise RuntimeError("AutoencoderKL.encode runs on the MI355X only (no CPU path in this framework)")
        n, _, h, w = x.shape
        if self.sample_posterior and noise is None:
            noise = torch.randn((n, self.ddconfig["z_channels"], h // 8, w // 8))
        z = self.encoder_runtime(x.device).encode(x, noise if self.sample_posterior else None).to(x.dtype)
        return (z, {}) if return_reg_log else z

    def forward(self, x, **kwargs):
        raise NotImplementedError("training forward is out of scope")


class AutoencoderKLModeOnly(AutoencoderKL):
    sample_posterior = False


class AutoencodingEngine(torch.nn.Module):
    """Generic autoencoder wrapper (reference models/autoencoder.py:96-250, inference surface):
    `encoder_config` / `decoder_config` name Encoder / Decoder / VideoDecoder; no quant convs.
    This is how the north_star's temporal decoder (sgm.modules.autoencoding.temporal_ae.VideoDecoder)
    is wired in SVD-style configs; DiffusionEngine.decode_first_stage passes `timesteps`."""

    def __init__(self, *args, encoder_config, decoder_config, loss_config=None, regularizer_config=None,
                 optimizer_config=None, lr_g_factor=1.0, ckpt_path=None, **kwargs):
        super().__init__()
        from ..util import instantiate_from_config
        self.encoder = instantiate_from_config(encoder_config)
        self.decoder = instantiate_from_config(decoder_config)
        self.sample_posterior = True
        if regularizer_config is not None:
            self.sample_posterior = bool((regularizer_config.get("params") or {}).get("sample", True))
        self._dec_rt = self._enc_rt = None
        self._dec_key = self._enc_key = None
        if ckpt_path is not None:
            from safetensors.torch import load_file
            sd = load_file(ckpt_path) if ckpt_path.endswith("safetensors") else torch.load(ckpt_path, map_location="cpu")
            self.load_state_dict(sd.get("state_dict", sd), strict=False)

    @property
    def is_video_decoder(self):
        return getattr(self.decoder, "temporal", False)

    def _key(self, device):
        return params_key(self, device)

    def runtime(self, device):
        """the decoder's HIP runtime (weights re-laid-out once per parameter state), as AutoencoderKL.runtime"""
        from hi3d_hip.runtime_vae import VAEDecoderRuntime, VideoDecoderRuntime
        key = self._key(device)
        if self._dec_rt is None or self._dec_key != key:
            cls = VideoDecoderRuntime if self.is_video_decoder else VAEDecoderRuntime
            self._dec_rt, self._dec_key = cls(self.state_dict(), self.decoder.ddconfig, device), key
        return self._dec_rt

    def encoder_runtime(self, device):
        from hi3d_hip.runtime_vae import VAEEncoderRuntime
        key = self._key(device)
        if self._enc_rt is None or self._enc_key != key:
            self._enc_rt, self._enc_key = VAEEncoderRuntime(self.state_dict(), self.encoder.ddconfig, device), key
        return self._enc_rt

    @torch.no_grad()
    def decode(self, z, **kwargs):
        if not z.is_cuda:
            raise RuntimeError("decode runs on the MI355X only (no CPU path in this framework)")
        rt = self.runtime(z.device)
        if self.is_video_decoder:
            return rt.decode(z, timesteps=kwargs.get("timesteps")).to(z.dtype)
        return rt.decode(z).to(z.dtype)

    @torch.no_grad()
    def encode(self, x, return_reg_log=False, unregularized=False, noise=None):
        """reference models/autoencoder.py:196-209: z = encoder(x); unregularized -> the raw moments; otherwise the
        DiagonalGaussianRegularizer (regularizers/__init__.py:21-31): posterior.sample() -- noise drawn on the CPU generator
        like the reference's torch.randn(...).to(device) unless `noise` is given -- or .mode() with `sample: false`."""
        if not x.is_cuda:
            raise RuntimeError("encode runs on the MI355X only (no CPU path in this framework)")
        rt = self.encoder_runtime(x.device)
        if unregularized:
            return rt.encode(x, moments=True).to(x.dtype), {}
        n, _, h, w = x.shape
        if self.sample_posterior and noise is None:
            noise = torch.randn((n, self.encoder.ddconfig["z_channels"], h // 8, w // 8))
        z = rt.encode(x, noise if self.sample_posterior else None).to(x.dtype)
        return (z, {}) if return_reg_log else z
