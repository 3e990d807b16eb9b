"""First-stage (AutoencoderKL) DECODER runtime on MI355X.

Reference path: DiffusionEngine.decode_first_stage (sgm/models/diffusion.py:117-135) ->
AutoencodingEngineLegacy.decode (sgm/models/autoencoder.py:490-505, post_quant_conv) ->
Decoder.forward (sgm/modules/diffusionmodules/model.py:715-748).

Same design as the UNet runtime: channels-last bf16 frames [N, H*W, C], every 3x3 conv is
the implicit-GEMM MFMA kernel (nearest-2x upsample folded into its gather, residual add in
its epilogue), GroupNorm(eps 1e-6)+swish is one fused pass.  The mid-block attention is a
single 512-wide head over up to 16384 tokens; it is run un-fused -- S = q k^T (GEMM, fp32
scores), row softmax, O = P V (GEMM) -- because with 288 GB of HBM the 1 GiB score matrix
of one 1024x1024 frame is affordable and both products then run on the tuned GEMM kernel.
Frames are independent: decode() can be called on any slice of the clip (frame sharding
across GPUs, all-gather of the decoded frames afterwards).
"""
import functools

import os

import torch

from . import ops, pack

CZ_PAD = 64


def _on_own_device(fn):
    """Run a runtime method with the runtime's GPU as the current device: the kernels launch on the CURRENT device's
    stream, so a model living on cuda:1 in a process whose current device is cuda:0 must switch for the call."""
    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        dev = getattr(self, "dev", None) or torch.device(args[2] if len(args) > 2 else kwargs["device"])
        if dev.type != "cuda":
            return fn(self, *args, **kwargs)
        with torch.cuda.device(dev):
            return fn(self, *args, **kwargs)
    return wrapped


# HI3D_VAE_FLASH=1: the mid-block attention as ONE flash-style launch (hi3d_attn_d512: no score matrix in memory).  Measured
# on a 1024 x 1024 frame (16384 tokens, profiles/r04j_vae_flash_ab.log): 1.14 ms against 1.02 ms for GEMM -> fp32 scores (1 GiB)
# -> softmax -> GEMM, whose two products run on the tuned wide GEMM tile -- so the three-launch form stays the default where
# the score buffer fits, and the flash kernel is the opt-in for memory-constrained use (and SURVEY 8b's `attn_fwd_d512`).
VAE_FLASH = os.environ.get("HI3D_VAE_FLASH", "0") == "1"


# The chunks of a clip (decode_first_stage / encode_first_stage: `en_and_decode_n_samples_a_time` frames per call, 1 at stage 2)
# are independent 2-D problems.  HI3D_VAE_STREAMS=n (default 2) issues them round-robin on the caller's stream and n - 1 side
# streams, so that the HBM-bound launches of one frame (GroupNorm: 20 % of a frame, the score softmax) share the chip with the
# MFMA-bound convolutions of the other and the tails of one frame's short launches are filled by the other's.  Each stream has
# its own split-K / GroupNorm scratch (ops keys them by stream); results are the same launches on the same data.
VAE_STREAMS = int(os.environ.get("HI3D_VAE_STREAMS", "2"))
UP_PHASES = os.environ.get("HI3D_UP_PHASES", "1") != "0"   # up-sampling convs as four 2x2 phase convs (see VAEDecoderRuntime._conv)
_SIDE = {}


def run_chunks(fn, chunks, device, prepare=None):
    """[fn(lo, hi) for lo, hi in chunks], chunk i on stream i % VAE_STREAMS (0 = the caller's; the side streams are joined before
    returning); fn returns one tensor.  prepare(): called first, on the caller's stream -- the engine builds the model's HIP
    runtime there (weight re-layout), so that nothing a side stream reads is still being made (do not rely on chunk
    0 for that); per-stream state (split-K / GroupNorm scratch, grown on demand for a ragged last chunk) is keyed by stream."""
    dev = torch.device(device)
    if prepare is not None:
        prepare()
    n = min(VAE_STREAMS, len(chunks))
    if dev.type != "cuda" or n < 2 or ops.PROFILER is not None or torch.cuda.is_current_stream_capturing():
        return [fn(lo, hi) for lo, hi in chunks]                 # (per-kernel timing wants kernels that do not share the chip)
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream()
        pool = _SIDE.setdefault(dev.index, [])
        while len(pool) < n - 1:
            pool.append(torch.cuda.Stream(device=dev))
        # Chunk 0 is ISSUED on the caller's stream before the side streams are let go: whatever the first call builds lazily on
        # that stream (the model's HIP runtime re-lays its weights out on first use) must be complete before another stream
        # reads it -- the side streams therefore start behind chunk 0 and run next to chunks 2, 3, ...  (Found as NaN frames in
        # a decode that was the model's first, in a long pytest process: the side stream read weights still being packed.)
        outs = [fn(*chunks[0])]
        for side in pool[:n - 1]:
            side.wait_stream(main)                               # (also: the latents were produced on the caller's stream)
        for i, (lo, hi) in enumerate(chunks):
            if i == 0:
                continue
            if i % n == 0:
                outs.append(fn(lo, hi))
            else:
                with torch.cuda.stream(pool[i % n - 1]):
                    o = fn(lo, hi)
                o.record_stream(main)                            # allocated in the side stream's pool, consumed on `main`
                outs.append(o)
        for side in pool[:n - 1]:
            main.wait_stream(side)
    return outs


class VAEDecoderRuntime:
    @_on_own_device
    def __init__(self, state_dict, ddconfig, device, prefix=""):
        self.dd, self.dev = dict(ddconfig), torch.device(device)
        dd = self.dd
        self.ch, self.mult, self.nres = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
        if self.ch % 64:
            raise ops._l.Hi3dError("VAE decoder runtime needs ch % 64 == 0 (Hi3D uses ch=128)")
        if dd["z_channels"] > 8:
            raise ops._l.Hi3dError("z_channels > 8 not supported")
        self._pack(state_dict, prefix)

    def _pack(self, sd, P):
        dev = self.dev
        g = lambda k: sd[P + k].detach().to(dev)
        f32 = lambda k: pack.f32(g(k))
        W = {}
        zc = self.dd["z_channels"]
        if (P + "post_quant_conv.weight") in sd:
            W["pq.w"] = f32("post_quant_conv.weight").reshape(zc, -1).contiguous()
            W["pq.b"] = f32("post_quant_conv.bias")
        else:                                   # AutoencodingEngine (SVD-style) has no quant convs
            W["pq.w"] = torch.eye(zc, device=dev); W["pq.b"] = torch.zeros(zc, device=dev)
        D = "decoder."
        W["conv_in.w"] = pack.pack_conv3x3(g(D + "conv_in.weight"), cin_pad=CZ_PAD); W["conv_in.b"] = f32(D + "conv_in.bias")

        def resnet(p):
            for n in ("norm1", "norm2"):
                W[f"{p}.{n}.g"] = f32(f"{D}{p}.{n}.weight"); W[f"{p}.{n}.b"] = f32(f"{D}{p}.{n}.bias")
            for n in ("conv1", "conv2"):
                W[f"{p}.{n}.w"] = pack.pack_conv3x3(g(f"{D}{p}.{n}.weight")); W[f"{p}.{n}.b"] = f32(f"{D}{p}.{n}.bias")
            if (P + D + p + ".nin_shortcut.weight") in sd:
                W[p + ".nin.w"] = pack.pack_conv1x1(g(f"{D}{p}.nin_shortcut.weight")); W[p + ".nin.b"] = f32(f"{D}{p}.nin_shortcut.bias")

        resnet("mid.block_1"); resnet("mid.block_2")
        a = "mid.attn_1"
        W[a + ".norm.g"] = f32(D + a + ".norm.weight"); W[a + ".norm.b"] = f32(D + a + ".norm.bias")
        W[a + ".qkv.w"] = pack._bf16(torch.cat([pack.pack_conv1x1(g(f"{D}{a}.{n}.weight")) for n in ("q", "k", "v")], 0))
        W[a + ".qkv.b"] = torch.cat([f32(f"{D}{a}.{n}.bias") for n in ("q", "k", "v")]).contiguous()
        W[a + ".o.w"] = pack.pack_conv1x1(g(D + a + ".proj_out.weight")); W[a + ".o.b"] = f32(D + a + ".proj_out.bias")
        for lvl in range(len(self.mult)):
            for b in range(self.nres + 1):
                resnet(f"up.{lvl}.block.{b}")
            if lvl != 0:
                W[f"up.{lvl}.up.w"] = pack.pack_conv3x3(g(f"{D}up.{lvl}.upsample.conv.weight"))
                W[f"up.{lvl}.up.b"] = f32(f"{D}up.{lvl}.upsample.conv.bias")
                for ph, (wp, _) in enumerate(pack.pack_conv3x3_up_phases(g(f"{D}up.{lvl}.upsample.conv.weight"))):
                    W[f"up.{lvl}.up.w.ph{ph}"] = wp      # (HI3D_UP_PHASES: four 2x2 filters on the low-resolution image)
        W["norm_out.g"] = f32(D + "norm_out.weight"); W["norm_out.b"] = f32(D + "norm_out.bias")
        self.out_ch = self.dd["out_ch"]
        ocp = (self.out_ch + 3) // 4 * 4
        W["conv_out.w"] = pack.pack_conv3x3(g(D + "conv_out.weight"), cout_pad=ocp)
        W["conv_out.b"] = pack.pad_vec(g(D + "conv_out.bias"), ocp)
        self.W = W

    # ------------------------------------------------------------------
    def _conv(self, x, key, N, H, Wd, Cin, Cout, up=0, R1=None, out_fp32=False, gn=None):
        if up and UP_PHASES and (key + ".w.ph0") in self.W and R1 is None and not out_fp32 and Cin == Cout:
            # Upsample (nearest 2x) + conv3x3 (model.py:67-71) as four 2x2 convolutions on the low-resolution image, one per
            # output phase, with summed weights (pack.pack_conv3x3_up_phases): 4/9 of the multiply-adds, then one row interleave
            # (gn: the next norm's partial sums from the four placed launches -- one-frame calls)
            return ops.upsample_conv_phases(x, [self.W[f"{key}.w.ph{ph}"] for ph in range(4)], self.W[key + ".b"], N, H, Wd, Cout,
                                            gn=gn is not None)
        Ho, Wo = (2 * H, 2 * Wd) if up else (H, Wd)
        return ops.gemm(x, self.W[key + ".w"], M=N * Ho * Wo, N=Cout, K=9 * Cin, bias=self.W[key + ".b"], R1=R1,
                        out_fp32=out_fp32, conv3x3=dict(Hin=H, Win=Wd, Cin=Cin, Hout=Ho, Wout=Wo, stride=1, up2x=up), gn=gn)

    def _take_gp(self, t):
        """The GroupNorm partial sums the producer of `t` emitted with it (round 6: also conv2 + skip and proj_out + x, whose
        epilogues add a residual), or None.  One slot: they live in the stream's shared GroupNorm workspace and are valid only for
        the norm that follows the producer directly -- every consumer below is that norm; the identity check drops them otherwise."""
        g, self._gp = getattr(self, "_gp", None), None
        return g[1] if g is not None and g[0] is t else None

    def _resnet(self, p, x, N, H, Wd, Cin, Cout):
        W, HW = self.W, H * Wd
        h = ops.groupnorm_silu(x, W[p + ".norm1.g"], W[p + ".norm1.b"], N, HW, Cin, 1e-6, partials=self._take_gp(x))
        h, gp = self._conv(h, p + ".conv1", N, H, Wd, Cin, Cout, gn=(N, HW))      # (+ norm2's partial sums where the tile allows)
        h = ops.groupnorm_silu(h, W[p + ".norm2.g"], W[p + ".norm2.b"], N, HW, Cout, 1e-6, partials=gp)
        skip = x
        if (p + ".nin.w") in W:
            skip = ops.gemm(x, W[p + ".nin.w"], M=N * HW, N=Cout, K=Cin, bias=W[p + ".nin.b"])
        out, gpo = self._conv(h, p + ".conv2", N, H, Wd, Cout, Cout, R1=skip, gn=(N, HW))   # (+ the next norm's sums: conv2 + skip)
        self._gp = None if gpo is None else (out, gpo)
        return out

    def _attn(self, p, x, N, S, C):
        W = self.W
        if C % 64:
            raise ops._l.Hi3dError("attention width must be a multiple of 64")
        n = ops.groupnorm_silu(x, W[p + ".norm.g"], W[p + ".norm.b"], N, S, C, 1e-6, silu=False, partials=self._take_gp(x))
        qkv = ops.gemm(n, W[p + ".qkv.w"], M=N * S, N=3 * C, K=C, bias=W[p + ".qkv.b"])
        if C == 512 and VAE_FLASH:
            # round 4: one flash-style launch (hi3d_attn_d512), no score matrix in memory
            o = ops.attention_d512(qkv, N, S)
            out, gpo = ops.gemm(o, W[p + ".o.w"], M=N * S, N=C, K=C, bias=W[p + ".o.b"], R1=x, gn=(N, S))
            self._gp = None if gpo is None else (out, gpo)
            return out
        vt = ops.transpose_v(qkv[:, 2 * C:], N, C // 64, S, 3 * C)          # [N, C/64, 64, S_pad] == V^T [N][C][S_pad]
        S_pad = vt.shape[-1]
        o = torch.empty((N * S, C), device=x.device, dtype=torch.bfloat16)
        for f in range(N):                                                   # one frame's score matrix at a time
            q = qkv[f * S:(f + 1) * S]
            k = q[:, C:]
            sc = ops.gemm(q, k, M=S, N=S, K=C, lda=3 * C, ldw=3 * C, out_fp32=True)     # [S, S] fp32
            pr = ops.softmax_rows(sc, S, S, S_pad, float(C) ** -0.5)
            ops.gemm(pr, vt[f], M=S, N=C, K=S_pad, lda=S_pad, ldw=S_pad, out=o[f * S:(f + 1) * S])
        out, gpo = ops.gemm(o, W[p + ".o.w"], M=N * S, N=C, K=C, bias=W[p + ".o.b"], R1=x, gn=(N, S))
        self._gp = None if gpo is None else (out, gpo)
        return out

    @_on_own_device
    @torch.no_grad()
    def decode(self, z):
        """z: [N, Cz, h, w] latents already divided by scale_factor -> fp32 [N, out_ch, 8h, 8w]."""
        W = self.W
        N, _, H, Wd = z.shape
        top = self.ch * self.mult[-1]
        h = ops.vae_latent_prepare(z.to(self.dev), W["pq.w"], W["pq.b"], CZ_PAD)
        h = self._conv(h, "conv_in", N, H, Wd, CZ_PAD, top)
        h = self._resnet("mid.block_1", h, N, H, Wd, top, top)
        h = self._attn("mid.attn_1", h, N, H * Wd, top)
        h = self._resnet("mid.block_2", h, N, H, Wd, top, top)
        cin = top
        for lvl in reversed(range(len(self.mult))):
            cout = self.ch * self.mult[lvl]
            for b in range(self.nres + 1):
                h = self._resnet(f"up.{lvl}.block.{b}", h, N, H, Wd, cin, cout)
                cin = cout
            if lvl != 0:
                h, gpu_ = self._conv(h, f"up.{lvl}.up", N, H, Wd, cout, cout, up=1, gn=(N, 4 * H * Wd))
                self._gp = None if gpu_ is None else (h, gpu_)      # (the next level's first norm1)
                H, Wd = 2 * H, 2 * Wd
        h = ops.groupnorm_silu(h, W["norm_out.g"], W["norm_out.b"], N, H * Wd, cin, 1e-6, partials=self._take_gp(h))
        ocp = W["conv_out.b"].numel()
        out = self._conv(h, "conv_out", N, H, Wd, cin, ocp, out_fp32=True)
        return self._finish(out, N, H, Wd, ocp)

    def _finish(self, out, N, H, Wd, ocp):
        return ops.tokens_to_nchw(out, N, self.out_ch, H, Wd, ocp)


class _ResnetMixin:
    """conv / ResnetBlock / attention helpers shared by the encoder (same arithmetic as the
    decoder's; reference model.py:131-151,161-200)."""
    _conv = VAEDecoderRuntime._conv
    _take_gp = VAEDecoderRuntime._take_gp
    _resnet = VAEDecoderRuntime._resnet
    _attn = VAEDecoderRuntime._attn


class VAEEncoderRuntime(_ResnetMixin):
    """AutoencoderKL ENCODER (stage-2 pre-loop, once per clip):
    Encoder.forward (sgm/modules/diffusionmodules/model.py:576-601) + quant_conv +
    DiagonalGaussianRegularizer (models/autoencoder.py:468-488, regularizers/__init__.py:21-31).
    The stride-2 Downsample pads bottom/right only (model.py:76-90): `pad_br_only` conv."""

    @_on_own_device
    def __init__(self, state_dict, ddconfig, device, prefix=""):
        self.dd, self.dev = dict(ddconfig), torch.device(device)
        dd = self.dd
        self.ch, self.mult, self.nres = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
        if self.ch % 64:
            raise ops._l.Hi3dError("VAE encoder runtime needs ch % 64 == 0 (Hi3D uses ch=128)")
        if not dd.get("double_z", True):
            raise ops._l.Hi3dError("double_z=False not supported")
        sd, P, dev = state_dict, prefix, self.dev
        g = lambda k: sd[P + k].detach().to(dev)
        f32 = lambda k: pack.f32(g(k))
        W = {}
        E = "encoder."
        W["conv_in.w"] = pack.pack_conv3x3(g(E + "conv_in.weight"), cin_pad=CZ_PAD); W["conv_in.b"] = f32(E + "conv_in.bias")

        def resnet(p):
            for n in ("norm1", "norm2"):
                W[f"{p}.{n}.g"] = f32(f"{E}{p}.{n}.weight"); W[f"{p}.{n}.b"] = f32(f"{E}{p}.{n}.bias")
            for n in ("conv1", "conv2"):
                W[f"{p}.{n}.w"] = pack.pack_conv3x3(g(f"{E}{p}.{n}.weight")); W[f"{p}.{n}.b"] = f32(f"{E}{p}.{n}.bias")
            if (P + E + p + ".nin_shortcut.weight") in sd:
                W[p + ".nin.w"] = pack.pack_conv1x1(g(f"{E}{p}.nin_shortcut.weight")); W[p + ".nin.b"] = f32(f"{E}{p}.nin_shortcut.bias")

        for lvl in range(len(self.mult)):
            for b in range(self.nres):
                resnet(f"down.{lvl}.block.{b}")
            if lvl != len(self.mult) - 1:
                W[f"down.{lvl}.ds.w"] = pack.pack_conv3x3(g(f"{E}down.{lvl}.downsample.conv.weight"))
                W[f"down.{lvl}.ds.b"] = f32(f"{E}down.{lvl}.downsample.conv.bias")
        resnet("mid.block_1"); resnet("mid.block_2")
        a = "mid.attn_1"
        W[a + ".norm.g"] = f32(E + a + ".norm.weight"); W[a + ".norm.b"] = f32(E + a + ".norm.bias")
        W[a + ".qkv.w"] = pack._bf16(torch.cat([pack.pack_conv1x1(g(f"{E}{a}.{n}.weight")) for n in ("q", "k", "v")], 0))
        W[a + ".qkv.b"] = torch.cat([f32(f"{E}{a}.{n}.bias") for n in ("q", "k", "v")]).contiguous()
        W[a + ".o.w"] = pack.pack_conv1x1(g(E + a + ".proj_out.weight")); W[a + ".o.b"] = f32(E + a + ".proj_out.bias")
        W["norm_out.g"] = f32(E + "norm_out.weight"); W["norm_out.b"] = f32(E + "norm_out.bias")
        self.zc = dd["z_channels"]
        W["conv_out.w"] = pack.pack_conv3x3(g(E + "conv_out.weight")); W["conv_out.b"] = f32(E + "conv_out.bias")
        if (P + "quant_conv.weight") in sd:
            W["q.w"] = f32("quant_conv.weight").reshape(2 * self.zc, -1).contiguous(); W["q.b"] = f32("quant_conv.bias")
        else:                                   # AutoencodingEngine (models/autoencoder.py:96-210) has no quant convs
            W["q.w"] = torch.eye(2 * self.zc, device=dev); W["q.b"] = torch.zeros(2 * self.zc, device=dev)
        self.W = W

    @_on_own_device
    @torch.no_grad()
    def encode(self, x, noise=None, moments=False):
        """x: [N, 3, H, W] in [-1, 1] -> z fp32 [N, Cz, H/8, W/8]; `noise` (same shape as z)
        selects posterior.sample(), None the mode.  moments=True: the encoder's raw output [N, 2 Cz, H/8, W/8]
        (AutoencodingEngine.encode(unregularized=True), models/autoencoder.py:201-203)."""
        W = self.W
        N, _, H, Wd = x.shape
        h = ops.nchw_to_tokens(x.to(self.dev), CZ_PAD)
        cin = self.ch
        h = self._conv(h, "conv_in", N, H, Wd, CZ_PAD, cin)
        for lvl in range(len(self.mult)):
            cout = self.ch * self.mult[lvl]
            for b in range(self.nres):
                h = self._resnet(f"down.{lvl}.block.{b}", h, N, H, Wd, cin, cout)
                cin = cout
            if lvl != len(self.mult) - 1:
                if H % 2 or Wd % 2:
                    raise ops._l.Hi3dError("encoder input height/width must be divisible by 8")
                h = ops.gemm(h, W[f"down.{lvl}.ds.w"], M=N * (H // 2) * (Wd // 2), N=cin, K=9 * cin, bias=W[f"down.{lvl}.ds.b"],
                             conv3x3=dict(Hin=H, Win=Wd, Cin=cin, Hout=H // 2, Wout=Wd // 2, stride=2, up2x=0, pad_br_only=1))
                H, Wd = H // 2, Wd // 2
        h = self._resnet("mid.block_1", h, N, H, Wd, cin, cin)
        h = self._attn("mid.attn_1", h, N, H * Wd, cin)
        h = self._resnet("mid.block_2", h, N, H, Wd, cin, cin)
        h = ops.groupnorm_silu(h, W["norm_out.g"], W["norm_out.b"], N, H * Wd, cin, 1e-6)
        mom = self._conv(h, "conv_out", N, H, Wd, cin, 2 * self.zc, out_fp32=True)
        if moments:
            return ops.tokens_to_nchw(mom, N, 2 * self.zc, H, Wd, mom.shape[-1])
        if noise is not None:
            noise = noise.to(self.dev, torch.float32).contiguous()
        return ops.vae_posterior(mom, W["q.w"], W["q.b"], noise, N, self.zc, H, Wd)


class VideoDecoderRuntime(VAEDecoderRuntime):
    """Temporal VAE decoder, time_mode 'conv-only' (sgm/modules/autoencoding/temporal_ae.py:18-107,293-349; hooked in by
    DiffusionEngine.decode_first_stage, models/diffusion.py:126-129).  Every ResnetBlock is followed by a temporal ResBlock
    (GroupNorm over t,h,w -> SiLU -> Conv3d, twice, no timestep embedding) blended in as x_s + sigmoid(mix_factor) * h_t,
    and conv_out is followed by a 3-channel Conv3d.  video_kernel_size (read off the weights' shape):

    * [3, 1, 1] (SVD / Hi3D): the UNet's time_stack kernels -- the frame axis is indexed inside the Conv3d (3,1,1) gather and
      the norm kernels, nothing is permuted;
    * 3 = [3, 3, 3] (the reference class's default, temporal_ae.py:299): an isotropic Conv3d is the sum over kt of a 2-D 3x3
      conv of the frame t + kt - 1 -- three launches of the conv3x3 gather on a clip buffer with one zero frame at either end
      (the GroupNorm writes its output between them), the partial sums carried as the R1 operand of the next launch and the
      residual / blend tail fused into the last; conv_out's 3-channel time_mix_conv on hi3d_time_mix_small_k3."""

    @_on_own_device
    def __init__(self, state_dict, ddconfig, device, prefix=""):
        super().__init__(state_dict, ddconfig, device, prefix)
        dev, sd, P = self.dev, state_dict, prefix
        g = lambda k: sd[P + k].detach().to(dev)
        f32 = lambda k: pack.f32(g(k))
        D = "decoder."
        W = self.W
        oc = self.out_ch
        tm = g(D + "conv_out.time_mix_conv.weight")
        ks = tuple(tm.shape[2:])
        if ks not in ((3, 1, 1), (3, 3, 3)):
            raise ops._l.Hi3dError(f"VideoDecoder: video_kernel_size {list(ks)} is not built ([3, 1, 1] and 3 are)")
        self.iso = ks == (3, 3, 3)
        names = ["mid.block_1", "mid.block_2"] + [f"up.{l}.block.{b}" for l in range(len(self.mult)) for b in range(self.nres + 1)]
        for p in names:
            q = p + ".time_stack"
            for n in ("in_layers.0", "out_layers.0"):
                W[f"{q}.{n}.g"] = f32(f"{D}{q}.{n}.weight"); W[f"{q}.{n}.b"] = f32(f"{D}{q}.{n}.bias")
            for n in ("in_layers.2", "out_layers.3"):
                w = g(f"{D}{q}.{n}.weight")
                if tuple(w.shape[2:]) != ks:
                    raise ops._l.Hi3dError(f"VideoDecoder: {q}.{n} has kernel {list(w.shape[2:])}, time_mix_conv {list(ks)}")
                if self.iso:                     # [O, I, kt, ky, kx] -> three 2-D filters, one per frame offset
                    for kt in range(3):
                        W[f"{q}.{n}.w{kt}"] = pack.pack_conv3x3(w[:, :, kt])
                else:
                    W[f"{q}.{n}.w"] = pack.pack_convt3(w)
                W[f"{q}.{n}.b"] = f32(f"{D}{q}.{n}.bias")
            W[p + ".alpha"] = torch.sigmoid(g(f"{D}{p}.mix_factor").float()).reshape(1)
        W["tmix.w"] = pack.f32(tm).contiguous() if self.iso else pack.f32(tm).reshape(oc, oc, 3).contiguous()
        W["tmix.b"] = f32(D + "conv_out.time_mix_conv.bias")
        self._T = None

    def _conv3d_iso(self, hpad, key, b, T, H, Wd, C, R2=None, a1=None, out=None):
        """Conv3d(C, C, 3, padding 1) of clip b: hpad [B, T + 2, H * W, C] (frames 0 and T + 1 zero) -> [T * H * W, C];
        with R2 / a1 the tail a1 * (conv + bias) + R2 (AlphaBlender folded, see _resnet) rides the last launch.
        Rounding: the partial sum over the frame taps travels between the three launches as the bf16 R1 operand (the
        ABI's residual type), so it is rounded to bf16 TWICE more than a single fp32 accumulation over all 27 taps would be --
        each rounding 2^-9 relative, against the 4e-2 / PSNR 35 dB bound of the `videodec_*_k3` goldens (measured with it:
        rel <= 1.2e-2).  `video_kernel_size=3` is the reference class's default but no shipped Hi3D / SVD config uses it."""
        W, HW = self.W, H * Wd
        geo = dict(Hin=H, Win=Wd, Cin=C, Hout=H, Wout=Wd, stride=1, up2x=0)
        clip = hpad[b].reshape((T + 2) * HW, C)
        acc = ops.gemm(clip[HW:], W[key + ".w1"], M=T * HW, N=C, K=9 * C, bias=W[key + ".b"], conv3x3=geo)            # frame t
        acc = ops.gemm(clip, W[key + ".w0"], M=T * HW, N=C, K=9 * C, R1=acc, conv3x3=geo)                              # frame t - 1
        return ops.gemm(clip[2 * HW:], W[key + ".w2"], M=T * HW, N=C, K=9 * C, R1=acc, R2=R2, a1=a1, out=out,           # frame t + 1
                        rows_per_group=HW, conv3x3=geo)

    def _resnet(self, p, x, N, H, Wd, Cin, Cout):
        xs = super()._resnet(p, x, N, H, Wd, Cin, Cout)
        W, T, HW = self.W, self._T, H * Wd
        B = N // T
        q = p + ".time_stack"
        if self.iso:
            out = torch.empty_like(xs)
            # one persistent padded clip buffer per shape: only its two border frames are zero and stay zero (the GroupNorms
            # below overwrite frames 1 .. T in full); round 5 allocated and zero-filled all T + 2 frames in every block
            key_ = (B, T, HW, Cout, str(xs.device), torch.cuda.current_stream().cuda_stream)   # (per stream: run_chunks alternates two)
            hpad = self._hpad.get(key_) if hasattr(self, "_hpad") else None
            if hpad is None:
                if not hasattr(self, "_hpad"):
                    self._hpad = {}
                hpad = self._hpad[key_] = torch.zeros((B, T + 2, HW, Cout), device=xs.device, dtype=torch.bfloat16)
            a1 = W[p + ".alpha"].expand(T).contiguous()
            for b in range(B):
                rows = slice(b * T * HW, (b + 1) * T * HW)
                mid = hpad[b, 1:T + 1].reshape(T * HW, Cout)
                ops.groupnorm_silu(xs[rows], W[q + ".in_layers.0.g"], W[q + ".in_layers.0.b"], 1, T * HW, Cout, 1e-5, out=mid)
                h = self._conv3d_iso(hpad, q + ".in_layers.2", b, T, H, Wd, Cout)
                ops.groupnorm_silu(h, W[q + ".out_layers.0.g"], W[q + ".out_layers.0.b"], 1, T * HW, Cout, 1e-5, out=mid)
                self._conv3d_iso(hpad, q + ".out_layers.3", b, T, H, Wd, Cout, R2=xs[rows], a1=a1, out=out[rows])
            return out
        tg = dict(T=T, HW=HW, Cin=Cout)
        h = ops.groupnorm_silu(xs, W[q + ".in_layers.0.g"], W[q + ".in_layers.0.b"], B, T * HW, Cout, 1e-5, partials=self._take_gp(xs))
        h, gp = ops.gemm(h, W[q + ".in_layers.2.w"], M=N * HW, N=Cout, K=3 * Cout, bias=W[q + ".in_layers.2.b"], convt3=tg,
                         gn=(B, T * HW))
        h = ops.groupnorm_silu(h, W[q + ".out_layers.0.g"], W[q + ".out_layers.0.b"], B, T * HW, Cout, 1e-5, partials=gp)
        a1 = W[p + ".alpha"].expand(N).contiguous()
        # alpha*(x_s + h_t) + (1-alpha)*x_s == x_s + alpha*h_t          (temporal_ae.py:72-79)
        out, gpo = ops.gemm(h, W[q + ".out_layers.3.w"], M=N * HW, N=Cout, K=3 * Cout, bias=W[q + ".out_layers.3.b"],
                            a1=a1, R2=xs, rows_per_group=HW, convt3=tg, gn=(N, HW))
        self._gp = None if gpo is None else (out, gpo)
        return out

    @_on_own_device
    @torch.no_grad()
    def decode(self, z, timesteps=None):
        N = z.shape[0]
        T = N if timesteps is None else int(timesteps)
        if N % T:
            raise ops._l.Hi3dError("VideoDecoder: batch is not a multiple of timesteps")
        self._T = T
        self._defer_out = True
        return super().decode(z)

    def _finish(self, out, N, H, Wd, ocp):
        # conv_out's 2-D result (fp32 [N*H*W, ocp]) -> time_mix_conv -> NCHW
        if self.iso:
            return ops.time_mix_small_k3(out, self.W["tmix.w"], self.W["tmix.b"], N // self._T, self._T, H, Wd, self.out_ch)
        return ops.time_mix_small(out, self.W["tmix.w"], self.W["tmix.b"], N // self._T, self._T, H, Wd, self.out_ch)
