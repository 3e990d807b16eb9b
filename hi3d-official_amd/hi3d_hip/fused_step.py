"""One Euler-EDM sampler step with classifier-free guidance as ONE kernel sequence -- and, from the
third step on, ONE HIP-graph replay.

Reference chain that this replaces, per step (25 per clip):
  sampling.py:93-107   EDMSampler.sampler_step       x + (s' - s) (x - D)/s
  guiders.py:78-99     LinearPredictionGuider        cat([x, x]), cat(uc[k], c[k]); x_u + w_t (x_c - x_u)
  denoiser.py:23-39    Denoiser.forward              F(x c_in, ln(s)/4) c_out + x c_skip
  wrappers.py:23-34    OpenAIWrapper.forward         cat((x, c["concat"]), dim=1)
  video_model.py:442   VideoUNet.forward
i.e. ~30 small elementwise / cat launches around the UNet, and a 13-channel `concat` tensor that is
re-concatenated 25 times although it is constant for the clip.

Here:  hi3d_cfg_update_x  (x * c_in -> the 4 latent channels of a PERSISTENT bf16 token buffer whose
       conditioning channels were written once per clip by hi3d_cfg_prepare; also c_noise)
    -> UNetRuntime.forward_tokens  (HIP kernels only)
    -> hi3d_sampler_step_dev  (denoiser scaling + CFG + Euler update in one pass)
with sigma / sigma_next read from a 2-float device buffer, so the kernel arguments are identical
every step and the sequence is captured once into a HIP graph (shapes are static).

How it is reached without changing the reference's API: the reference hands the sampler an opaque
closure `denoiser(input, sigma, c)` (pipeline_i2v_eval_v01.py:85-88).  `EDMSampler.step_call` sends a
`StepRequest` through that closure in place of the batch-doubled input; both reference pipelines pass
`input` on untouched to `Denoiser.forward`, which recognises the request and runs this module.  A
closure / network / guider combination that is not the Hi3D one gets the generic path (same math,
one launch per reference op).  HI3D_FUSED_STEP=0 disables the request, HI3D_STEP_GRAPH=0 the graph.
"""
import os

import torch

from . import ops
from .runtime_unet import CIN_PAD

_KNOWN_KEYS = {"crossattn", "vector", "concat"}


def enabled():
    return os.environ.get("HI3D_FUSED_STEP", "1") != "0"


class StepResult:
    """What Denoiser.forward returns for a StepRequest it could serve: the next latent state."""

    def __init__(self, x):
        self.x = x


class StepRequest:
    """Travels through the caller's denoiser closure as `input`."""

    def __init__(self, guider, x, sigmas, i, c, uc):
        self.guider, self.x, self.sigmas, self.i, self.c, self.uc = guider, x, sigmas, i, c, uc

    @staticmethod
    def eligible(sampler, x, c, uc):
        from sgm.modules.diffusionmodules.guiders import LinearPredictionGuider
        g = sampler.guider
        return (enabled() and type(g) is LinearPredictionGuider and not g.additional_cond_keys
                and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and x.shape[1] == 4 and x.shape[0] == g.num_frames
                and isinstance(c, dict) and isinstance(uc, dict) and set(c) <= _KNOWN_KEYS and set(c) == set(uc)
                and {"crossattn", "vector"} <= set(c))

    # what the generic path would have been given (used when the request cannot be served)
    def materialize(self):
        s = self.sigmas[self.i].expand(self.x.shape[0])
        return self.guider.prepare_inputs(self.x, s, self.c, self.uc)

    def __mul__(self, other):       # a closure that does arithmetic on `input` is not transparent
        raise TypeError("StepRequest is not a tensor")
    __rmul__ = __add__ = __radd__ = __mul__


def serve(denoiser_module, network, req, extra):
    """Called by Denoiser.forward.  Returns StepResult, or None when the structure is not the fused kind."""
    from sgm.modules.diffusionmodules.denoiser_scaling import VScalingWithEDMcNoise
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    if type(denoiser_module.scaling) is not VScalingWithEDMcNoise or type(network) is not OpenAIWrapper:
        return None
    unet = network.diffusion_model
    if not isinstance(unet, VideoUNet):
        return None
    T = extra.get("num_video_frames")
    if set(extra) - {"num_video_frames", "image_only_indicator"} or T is None or int(T) != req.x.shape[0]:
        return None
    concat = req.c.get("concat")
    Cc = 0 if concat is None or concat.numel() == 0 else concat.shape[1]
    if 4 + Cc != unet.cfg["in_channels"] or (Cc and tuple(concat.shape) != (req.x.shape[0], Cc) + tuple(req.x.shape[2:])):
        return None
    rt = unet.runtime(req.x.device)
    key = (int(T),) + tuple(req.x.shape[2:])
    stepper = rt.steppers.get(key)
    if stepper is None:
        stepper = rt.steppers[key] = FusedStepper(rt, int(T), req.x.shape[2], req.x.shape[3])
    return StepResult(stepper.step(req, extra.get("image_only_indicator")))


class FusedStepper:
    def __init__(self, rt, T, H, W):
        dev = rt.dev
        self.rt, self.T, self.H, self.W = rt, T, H, W
        self.tok = torch.zeros((2 * T * H * W, CIN_PAD), device=dev, dtype=torch.bfloat16)   # network input, persistent
        self.sig = torch.zeros(2, device=dev, dtype=torch.float32)                          # sigma_i, sigma_{i+1}
        self.tvec = torch.zeros(2 * T, device=dev, dtype=torch.float32)                     # c_noise per batch row
        self.x = torch.zeros((T, 4, H, W), device=dev, dtype=torch.float32)                 # latent state of the graph
        self.scale = torch.zeros(T, device=dev, dtype=torch.float32)                        # guidance scale per frame: the graph holds
        self._scale_src = None                                                              # THIS buffer, refreshed in place
        self._concat = None            # (concat_c, version, concat_uc, version): what self.tok currently holds
        self.merged = {}               # key -> (c tensor, ver, uc tensor, ver, cat(uc, c))
        self.graph = None
        self._cap_stream = None
        self.eager_steps = 0
        self.use_graph = os.environ.get("HI3D_STEP_GRAPH", "1") != "0"

    # ---- clip constants ---------------------------------------------------------------
    def _cat(self, k, c, uc):
        """cat(uc[k], c[k]) (guiders.py:93-95) built once per clip, identity-checked with the refs held."""
        m = self.merged.get(k)
        if m is None or m[0] is not c or m[1] != c._version or m[2] is not uc or m[3] != uc._version:
            m = self.merged[k] = (c, c._version, uc, uc._version, torch.cat((uc.to(self.rt.dev), c.to(self.rt.dev)), 0))
        return m[4]

    def _refresh(self, req, ioi):
        c, uc, g = req.c, req.uc, req.guider
        cc, cu = c.get("concat"), uc.get("concat")
        if cc is not None and cc.numel() == 0:
            cc = cu = None
        k = None if cc is None else (cc, cc._version, cu, cu._version)
        if self._concat is None or (k is None) != (self._concat[0] is None) or \
                (k is not None and any(a is not b if torch.is_tensor(a) else a != b for a, b in zip(k, self._concat))):
            f32 = lambda t: None if t is None else t.to(self.rt.dev, torch.float32).contiguous()
            ops.cfg_prepare(self.x, f32(cu), f32(cc), CIN_PAD, 0.0, out=self.tok)    # conditioning channels, once per clip
            self._concat = k if k is not None else (None,)
        # guider.scale is re-read whenever the tensor the guider holds is another one or was written to (a second
        # sampler on the same model, a CFG sweep, a direct assignment): copied INTO the buffer the captured graph reads
        gs = g.scale
        if self._scale_src is None or self._scale_src[0] is not gs or self._scale_src[1] != gs._version:
            flat = gs.reshape(-1)
            if flat.numel() != self.T:
                raise ValueError(f"guider.scale has {flat.numel()} entries for {self.T} frames")
            self.scale.copy_(flat.to(self.rt.dev, torch.float32))
            self._scale_src = (gs, gs._version)
        ctx = self._cat("crossattn", c["crossattn"], uc["crossattn"])
        y = self._cat("vector", c["vector"], uc["vector"])
        return self.rt.clip_consts(ctx, y, ioi, 2 * self.T, self.T)

    # ---- the step -----------------------------------------------------------------------
    def _body(self, st):
        T, HW = self.T, self.H * self.W
        ops.cfg_update_x(self.x, self.tok, self.sig, self.tvec, T, HW, CIN_PAD)
        net = self.rt.forward_tokens(self.tok, 2 * T, self.H, self.W, self.tvec, st, T)
        ops.sampler_step_dev(self.x, self.x, net, self.scale, self.sig, T, HW, net.stride(0))

    @torch.no_grad()
    def step(self, req, ioi):
        with torch.cuda.device(self.rt.dev):
            st = self._refresh(req, ioi)
            self.sig.copy_(req.sigmas[req.i:req.i + 2])          # device-to-device, 8 bytes
            self.x.copy_(req.x)
            if self.graph is not None and not ops.PROFILER:
                self.graph.replay()
            elif self.use_graph and self.eager_steps >= 2 and not ops.PROFILER:
                # two eager steps have raised every kernel's LDS limit and filled the lazy caches
                # (frame-position embeddings); shapes and pointers are static from here on
                g = torch.cuda.CUDAGraph()
                # captured on a stream of our own whose split-K scratch is registered BEFORE the capture (the graph bakes the
                # pointer in; a scratch buffer belongs to one stream: hi3d_gemm_set_workspace_for_stream)
                if self._cap_stream is None:
                    self._cap_stream = torch.cuda.Stream(device=self.rt.dev)
                    ops._ensure_gemm_workspace(self.rt.dev, self._cap_stream)
                with torch.cuda.graph(g, stream=self._cap_stream):
                    self._body(st)
                self.graph = g
                g.replay()
            else:
                self._body(st)
                self.eager_steps += 1
            return self.x.clone()
