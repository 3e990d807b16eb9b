#!/usr/bin/env python
"""decode_first_stage of a 16-frame clip at 1024^2 (and 512^2) under different chunkings: frames per decode call x HIP streams
(hi3d_hip.runtime_vae.run_chunks).  The shipped configs decode `en_and_decode_n_samples_a_time` = 1 frame per call on 2 streams;
the frames of an AutoencoderKL clip are independent 2-D problems, so any chunking gives the same frames (up to the tile variant
the larger M selects).   usage: python tools/vae_chunk_ab.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd"))
import torch  # noqa: E402

from hi3d_hip import runtime_vae, synth  # noqa: E402
from sgm.models.autoencoder import AutoencoderKL  # noqa: E402

dev = torch.device("cuda:0")
dd = dict(attn_type="vanilla-xformers", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
          ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
ae = AutoencoderKL(embed_dim=4, ddconfig=dd)
synth.fill_module_(ae, 1, prefix="first_stage_model.")
ae = ae.to(dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
T = 16
for lat in (128, 64):
    z = torch.randn(T, 4, lat, lat, device=dev)
    ref = None
    combos = [(1, 2), (1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (1, 3), (8, 1), (16, 1)]
    best = {}
    for rep in range(reps + 1):                       # (round 0 = warm-up of every combination: scratch, re-layout)
        for chunk, streams in combos:
            runtime_vae.VAE_STREAMS = streams
            chunks = [(i, min(T, i + chunk)) for i in range(0, T, chunk)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = runtime_vae.run_chunks(lambda lo, hi: ae.decode(z[lo:hi]), chunks, dev)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3 / T
            if rep == 0:
                out = torch.cat(outs, 0)
                if ref is None:
                    ref = out
                    print(f"latent {lat}: reference = chunk 1 x 2 streams", flush=True)
                else:
                    d = (out.float() - ref.float()).abs().max().item()
                    print(f"  chunk {chunk:2d} x {streams} stream(s): max |diff| vs chunk 1 = {d:.3e}{' (bit-identical)' if d == 0 else ''}", flush=True)
                del out
            else:
                best.setdefault((chunk, streams), []).append(dt)
            del outs
    for (chunk, streams), v in best.items():
        print(f"latent {lat} ({lat * 8}^2): {chunk:2d} frame(s) per call x {streams} stream(s): " + "  ".join(f"{x:6.2f}" for x in v) + f"   ms/frame (best {min(v):.2f})", flush=True)
    del z, ref
    torch.cuda.empty_cache()
