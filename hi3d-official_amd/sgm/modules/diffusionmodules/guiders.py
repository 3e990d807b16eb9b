"""Classifier-free guidance with a per-frame linear scale (reference: guiders.py:60-99)."""
import torch

from ...util import append_dims


class IdentityGuider:
    def __call__(self, x, sigma):
        return x

    def prepare_inputs(self, x, s, c, uc):
        return x, s, {k: c[k] for k in c}


class LinearPredictionGuider:
    def __init__(self, max_scale, num_frames, min_scale=1.0, additional_cond_keys=None):
        self.min_scale, self.max_scale, self.num_frames = min_scale, max_scale, num_frames
        self.scale = torch.linspace(min_scale, max_scale, num_frames).unsqueeze(0)
        if additional_cond_keys is None:
            additional_cond_keys = []
        elif isinstance(additional_cond_keys, str):
            additional_cond_keys = [additional_cond_keys]
        self.additional_cond_keys = list(additional_cond_keys)
        self._scale_dev = {}
        self._merged = {}

    def _scale_on(self, device):
        # keyed on the tensor the attribute holds NOW (and its version): `guider.scale = ...` or an in-place edit is seen
        m = self._scale_dev.get(device)
        if m is None or m[0] is not self.scale or m[1] != self.scale._version:
            m = self._scale_dev[device] = (self.scale, self.scale._version, self.scale.to(device))
        return m[2]

    def __call__(self, x, sigma):
        T = self.num_frames
        x_u, x_c = x.chunk(2)
        b = x_u.shape[0] // T
        scale = append_dims(self._scale_on(x.device).expand(b, T), x_u.ndim + 1).to(x.dtype)
        x_u, x_c = x_u.reshape((b, T) + x_u.shape[1:]), x_c.reshape((b, T) + x_c.shape[1:])
        return (x_u + scale * (x_c - x_u)).reshape((b * T,) + x_u.shape[2:])

    def _cat(self, k, u, v):
        """cat(uc[k], c[k]) is constant over the steps of a clip: built once and reused while the two
        source tensors are the same objects at the same version (the cache holds them, so their memory
        cannot be handed to another clip's conditioning).  Downstream caches (UNetRuntime.clip_consts)
        key on the identity of what this returns."""
        m = self._merged.get(k)
        if m is None or m[0] is not u or m[1] != u._version or m[2] is not v or m[3] != v._version:
            m = self._merged[k] = (u, u._version, v, v._version, torch.cat((u, v), 0))
        return m[4]

    def prepare_inputs(self, x, s, c, uc):
        doubled = ["vector", "crossattn", "concat"] + self.additional_cond_keys
        c_out = {}
        for k in c:
            if k in doubled:
                c_out[k] = self._cat(k, uc[k], c[k])
            else:
                assert c[k] == uc[k]
                c_out[k] = c[k]
        return torch.cat([x, x]), torch.cat([s, s]), c_out
