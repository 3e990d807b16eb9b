#!/usr/bin/env python
"""Time the VAE decoder runtime at the Hi3D sizes (random-init full-width AutoencoderKL)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd"))
import torch  # noqa: E402

from hi3d_hip import ops, synth  # noqa: E402
from sgm.models.autoencoder import AutoencoderKL  # noqa: E402

dev = torch.device("cuda:0")
dd = dict(attn_type="vanilla-xformers", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
          ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
ae = AutoencoderKL(embed_dim=4, ddconfig=dd)
synth.fill_module_(ae, 1, prefix="first_stage_model.")
ae = ae.to(dev)
for n, lat, tf_per_frame in ((16, 64, 2.51), (1, 128, 10.47)):
    z = torch.randn(n, 4, lat, lat, device=dev)
    ae.decode(z)
    torch.cuda.synchronize()
    prof = ops.Profiler(); ops.PROFILER = prof
    t0 = time.perf_counter()
    out = ae.decode(z)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.PROFILER = None
    assert torch.isfinite(out).all()
    print(f"decode {n} frame(s) @ {lat * 8}^2: {dt * 1e3:.1f} ms  ({n * tf_per_frame / dt:.0f} TFLOP/s, "
          f"{dt * 1e3 / n:.1f} ms/frame)")
    for fam, d in sorted(prof.summary().items(), key=lambda kv: -kv[1]["ms"]):
        print(f"    {fam:16s} {d['ms']:8.2f} ms {d['launches']:4d} launches {d['flops'] / max(d['ms'], 1e-9) / 1e9:8.1f} TFLOP/s")
