"""First-stage autoencoder container (reference: sgm/models/autoencoder.py:437-520).

Holds the reference's parameter names (`encoder.*`, `decoder.*`, `quant_conv`,
`post_quant_conv`); `decode` runs hi3d_hip.runtime_vae.VAEDecoderRuntime (gfx950 kernels).
`encode` runs VAEEncoderRuntime (stage-2 pre-loop: once per frame, once per clip)."""
import torch

from ..util import ParamTree


def vae_param_shapes(dd, embed_dim):
    ch, mult, nres, zc = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"], dd["z_channels"]
    if dd.get("attn_resolutions"):
        raise NotImplementedError("attn_resolutions must be empty (only the mid-block attention exists in Hi3D)")
    S = {}

    def conv(p, o, i, k):
        S[p + ".weight"] = (o, i, k, k); S[p + ".bias"] = (o,)

    def norm(p, c):
        S[p + ".weight"] = (c,); S[p + ".bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin); conv(p + ".conv1", cout, cin, 3)
        norm(p + ".norm2", cout); conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".nin_shortcut", cout, cin, 1)

    def mid(p, c):
        resnet(p + ".block_1", c, c)
        norm(p + ".attn_1.norm", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(f"{p}.attn_1.{n}", c, c, 1)
        resnet(p + ".block_2", c, c)

    # encoder (model.py:487-575)
    conv("encoder.conv_in", ch, dd["in_channels"], 3)
    in_mult = [1] + mult
    for lvl in range(len(mult)):
        cin, cout = ch * in_mult[lvl], ch * mult[lvl]
        for b in range(nres):
            resnet(f"encoder.down.{lvl}.block.{b}", cin, cout); cin = cout
        if lvl != len(mult) - 1:
            conv(f"encoder.down.{lvl}.downsample.conv", cout, cout, 3)
    top = ch * mult[-1]
    mid("encoder.mid", top)
    norm("encoder.norm_out", top)
    conv("encoder.conv_out", 2 * zc if dd.get("double_z", True) else zc, top, 3)
    # decoder (model.py:604-714)
    conv("decoder.conv_in", top, zc, 3)
    mid("decoder.mid", top)
    cin = top
    for lvl in reversed(range(len(mult))):
        cout = ch * mult[lvl]
        for b in range(nres + 1):
            resnet(f"decoder.up.{lvl}.block.{b}", cin, cout); cin = cout
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv", cout, cout, 3)
    norm("decoder.norm_out", cin)
    conv("decoder.conv_out", dd["out_ch"], cin, 3)
    conv("quant_conv", (1 + dd.get("double_z", True)) * embed_dim, (1 + dd.get("double_z", True)) * zc, 1)
    conv("post_quant_conv", zc, embed_dim, 1)
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
        p0 = next(self.parameters())
        key = (torch.device(device), p0.data_ptr(), p0._version, p0.dtype)
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
        p0 = next(self.parameters())
        key = (torch.device(device), p0.data_ptr(), p0._version, p0.dtype)
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
