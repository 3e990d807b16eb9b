"""Device -> host hand-off at the tail of the path (reference: vtdm/util.py:13-50):
`tensor2vid` (un-normalise, clamp, uint8 HWC frames) and `export_to_video`."""
import struct

import numpy as np
import torch


def tensor2vid(video, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """video [i, 3, f, h, w] in [-1, 1] -> list of i*f uint8 HWC numpy frames, 'i c f h w -> (i f) h w c'
    (reference vtdm/util.py:13-21; values are truncated to uint8 like `astype('uint8')` there)."""
    m = torch.tensor(mean, device=video.device, dtype=video.dtype).reshape(1, -1, 1, 1, 1)
    s = torch.tensor(std, device=video.device, dtype=video.dtype).reshape(1, -1, 1, 1, 1)
    v = (video * s + m).clamp_(0, 1)
    i, c, f, h, w = v.shape
    frames = (v.permute(0, 2, 3, 4, 1).reshape(i * f, h, w, c) * 255).to(torch.uint8).cpu().numpy()
    return [fr for fr in frames]


def _write_avi_rgb24(frames, path, fps):
    """Uncompressed AVI (RIFF 'AVI ', one video stream of bottom-up BGR24 DIB frames): readable by
    ffmpeg / VLC / OpenCV, needs no codec library.  Rows are padded to 4 bytes as DIBs require."""
    h, w, _ = frames[0].shape
    stride = (w * 3 + 3) & ~3
    fsize = stride * h
    n = len(frames)

    def chunk(tag, data):
        return tag + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")

    def lst(tag, data):
        return b"LIST" + struct.pack("<I", len(data) + 4) + tag + data

    avih = struct.pack("<14I", int(1e6 / fps), fsize * int(round(fps)), 0, 0x10, n, 0, 1, fsize, w, h, 0, 0, 0, 0)
    strh = b"vids" + b"DIB " + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, 1, int(round(fps)), 0, n, fsize, 0xFFFFFFFF, 0, 0, 0, w, h)
    strf = struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, fsize, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi, index, off = [], [], 4
    for fr in frames:
        bgr = np.ascontiguousarray(fr[::-1, :, ::-1])                      # bottom-up, BGR
        if stride != w * 3:
            bgr = np.concatenate([bgr.reshape(h, w * 3), np.zeros((h, stride - w * 3), np.uint8)], axis=1)
        data = bgr.tobytes()
        movi.append(chunk(b"00db", data))
        index.append(b"00db" + struct.pack("<III", 0x10, off, len(data)))
        off += len(movi[-1])
    body = b"AVI " + hdrl + lst(b"movi", b"".join(movi)) + chunk(b"idx1", b"".join(index))
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def export_to_video(video_frames, output_video_path=None, save_to_gif=False, use_cv2=True, fps=8):
    """Frames (uint8 HWC RGB) -> file, returns the path written (reference vtdm/util.py:24-50).
    mp4 goes through OpenCV ('mp4v') when it is installed; without it (this image has no cv2 / imageio)
    the frames are written as an uncompressed AVI next to the requested name, so the pipeline's tail
    still produces a playable file."""
    if save_to_gif:
        import imageio                                              # reference behaviour; raises if absent
        path = output_video_path[:-3] + "gif" if output_video_path.endswith("mp4") else output_video_path
        imageio.mimsave(path, list(video_frames), fps=fps)
        return path
    if use_cv2 and not output_video_path.endswith(".avi"):
        try:
            import cv2
        except ImportError:
            cv2 = None
        if cv2 is not None:
            h, w, _ = video_frames[0].shape
            vw = cv2.VideoWriter(output_video_path, cv2.VideoWriter_fourcc(*"mp4v"), fps=fps, frameSize=(w, h))
            for fr in video_frames:
                vw.write(cv2.cvtColor(fr, cv2.COLOR_RGB2BGR))
            vw.release()
            return output_video_path
    path = output_video_path if output_video_path.endswith(".avi") else output_video_path.rsplit(".", 1)[0] + ".avi"
    _write_avi_rgb24(list(video_frames), path, fps)
    return path
