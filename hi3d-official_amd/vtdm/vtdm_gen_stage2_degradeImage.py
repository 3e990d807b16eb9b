"""Stage-2 (refiner) VideoLDM (reference: vtdm/vtdm_gen_stage2_degradeImage.py:28-86).
At inference it differs from stage 1 only in what the conditioner is fed: every frame of
the stage-1 video (not just frame 0) is the conditioning input."""
import torch

from .vtdm_gen_v01 import VideoLDM as _Stage1


class VideoLDM(_Stage1):
    @torch.no_grad()
    def add_custom_cond(self, batch, infer=False):
        batch["num_video_frames"] = self.num_samples
        video = batch["video"]                                   # b c t h w
        b, c, t, h, w = video.shape
        frames = video.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        dev = video.device
        batch["cond_frames_without_noise"] = video[:, :, 0].half()
        if infer:
            cond_aug = torch.full((b,), 0.02, device=dev)
        else:
            cond_aug = torch.exp(-3.0 + 0.5 * torch.randn((b,), device=dev))
        batch["cond_aug"] = cond_aug.half()
        noise_scale = cond_aug.repeat_interleave(t).reshape(b * t, 1, 1, 1)
        batch["cond_frames"] = (frames + noise_scale * torch.randn_like(frames)).half()
        if "image_only_indicator" not in batch:
            batch["image_only_indicator"] = torch.zeros((b, self.num_samples), device=dev).half()
        return batch
