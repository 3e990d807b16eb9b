"""Temporal VAE decoder container (reference: sgm/modules/autoencoding/temporal_ae.py:293-349).
Supported: time_mode 'conv-only' (the default and what SVD ships), merge_strategy 'learned', and video_kernel_size
[3, 1, 1] (what SVD / Hi3D configure) or 3 / [3, 3, 3] (the reference CLASS default, temporal_ae.py:299: an int makes every
time_stack conv and conv_out.time_mix_conv an isotropic Conv3d(3, padding 1), temporal_ae.py:87-98 and
openaimodel.py:257-261 with dims=3).  Runs as hi3d_hip.runtime_vae.VideoDecoderRuntime."""
from ..diffusionmodules.model import Decoder


class VideoDecoder(Decoder):
    temporal = True
    available_time_modes = ["all", "conv-only", "attn-only"]

    def __init__(self, *args, video_kernel_size=3, alpha=0.0, merge_strategy="learned", time_mode="conv-only", **kwargs):
        if time_mode != "conv-only":
            # the reference cannot construct these either: _make_attn hands the function make_time_attn to partialclass,
            # which subclasses it -> TypeError (temporal_ae.py:326, sgm/util.py:99)
            raise NotImplementedError(f"VideoDecoder time_mode={time_mode} (only 'conv-only' is built; the reference "
                                      "raises TypeError when constructing the other modes)")
        vks = [video_kernel_size] * 3 if isinstance(video_kernel_size, int) else [int(k) for k in video_kernel_size]
        if vks not in ([3, 1, 1], [3, 3, 3]):
            raise NotImplementedError(f"VideoDecoder video_kernel_size={video_kernel_size}: built for [3, 1, 1] and 3 (= [3, 3, 3])")
        if merge_strategy != "learned":
            raise NotImplementedError(f"merge_strategy={merge_strategy}")
        self.video_kernel_size, self.alpha, self.merge_strategy, self.time_mode = vks, alpha, merge_strategy, time_mode
        super().__init__(*args, **kwargs)
