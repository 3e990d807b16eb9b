"""`tensor2vid`: device -> host hand-off at the tail of the path (reference: vtdm/util.py:13-21)."""
import torch


def tensor2vid(video, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """video [1, 3, T, H, W] in [-1, 1] -> list of T uint8 HWC numpy frames."""
    m = torch.tensor(mean, device=video.device, dtype=video.dtype).reshape(1, -1, 1, 1, 1)
    s = torch.tensor(std, device=video.device, dtype=video.dtype).reshape(1, -1, 1, 1, 1)
    v = (video * s + m).clamp_(0, 1)
    b, c, t, h, w = v.shape
    frames = (v.permute(2, 3, 0, 4, 1).reshape(t, h, b * w, c) * 255).to(torch.uint8).cpu().numpy()
    return [f for f in frames]
