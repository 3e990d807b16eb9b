"""First-stage autoencoder container (reference: sgm/models/autoencoder.py:437-520).

Holds the reference's parameter names (`encoder.*`, `decoder.*`, `quant_conv`,
`post_quant_conv`); `decode` runs hi3d_hip.runtime_vae.VAEDecoderRuntime (gfx950 kernels).
`encode` (stage-2 pre-loop only) is not built yet and says so."""
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

    def encode(self, x, **kwargs):
        raise NotImplementedError("VAE encoder (stage-2 pre-loop, once per clip) is not built yet: "
                                  "SURVEY.md section 8f rank 2")

    def forward(self, x, **kwargs):
        raise NotImplementedError("training forward is out of scope")


class AutoencoderKLModeOnly(AutoencoderKL):
    pass
