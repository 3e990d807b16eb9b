"""Temporal VAE decoder container (reference: sgm/modules/autoencoding/temporal_ae.py:293-349).
Supported: time_mode 'conv-only' (the default and what SVD ships) with video_kernel_size [3,1,1],
merge_strategy 'learned'.  Runs as hi3d_hip.runtime_vae.VideoDecoderRuntime."""
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
        if isinstance(video_kernel_size, int) or list(video_kernel_size) != [3, 1, 1]:
            raise NotImplementedError("VideoDecoder needs video_kernel_size [3, 1, 1]")
        if merge_strategy != "learned":
            raise NotImplementedError(f"merge_strategy={merge_strategy}")
        self.video_kernel_size, self.alpha, self.merge_strategy, self.time_mode = [3, 1, 1], alpha, merge_strategy, time_mode
        super().__init__(*args, **kwargs)
