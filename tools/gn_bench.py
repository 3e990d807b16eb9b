#!/usr/bin/env python
"""Per-shape timing of hi3d_groupnorm_silu (three-pass / from-partials / single-pass forms) -- run once per HI3D_GN_ONEPASS setting:
    for v in 0 24 48 96; do HI3D_GN_ONEPASS=$v python tools/gn_bench.py; done
Shapes: the GroupNorms of the 16^2 / 32^2 levels of the stage-2 UNet and of stage 1's lower half (inst, P, C, two-source split)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd"))
import torch  # noqa: E402
from hi3d_hip import ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(32, 256, 1280, 0), (2, 4096, 1280, 0), (32, 256, 2560, 1280), (32, 1024, 1280, 0), (2, 16384, 1280, 0), (32, 1024, 2560, 1280),
          (32, 64, 1280, 0), (2, 1024, 1280, 0), (32, 64, 2560, 1280), (16, 256, 1280, 0), (1, 4096, 1280, 0), (16, 1024, 1280, 0)]


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print(f"HI3D_GN_ONEPASS={os.environ.get('HI3D_GN_ONEPASS', '(default)')}")
for inst, P, C, C1 in SHAPES:
    g = torch.rand(C, device=dev) + 0.5
    b = torch.randn(C, device=dev)
    if C1:
        x1 = torch.randn((inst * P, C1), device=dev).to(torch.bfloat16)
        x2 = torch.randn((inst * P, C - C1), device=dev).to(torch.bfloat16)
        us = timeit(lambda: ops.groupnorm_silu(x1, g, b, inst, P, C, 1e-5, x2=x2))
    else:
        x = torch.randn((inst * P, C), device=dev).to(torch.bfloat16)
        us = timeit(lambda: ops.groupnorm_silu(x, g, b, inst, P, C, 1e-5))
    mb = inst * P * C * 2 / 1e6
    print(f"  inst={inst:3d} P={P:6d} C={C:5d} {'cat2' if C1 else '    '} {mb:7.1f} MB  {us:7.1f} us  {2 * mb / us * 1e3:7.0f} GB/s(alg)")
