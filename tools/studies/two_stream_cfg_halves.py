#!/usr/bin/env python
"""Study (no product change): do the two CFG halves of a sampler step overlap usefully on ONE GPU?

The unconditional and the conditional half of the doubled batch (guiders.py:88-99) never mix inside the VideoUNet, so a
step can be issued as two independent half-batch forwards on two HIP streams.  A step is a chain of ~750 kernels of which
~30 ms are bandwidth-bound (GroupNorm, LayerNorm, the N = 320 GEMMs, temporal attention) and the rest MFMA-bound, and every
kernel ends in a tail during which CUs idle: two independent chains could fill each other's tails and overlap HBM-bound
with MFMA-bound work -- or lose more to the smaller launches (M halves; the simulated clip-parallel rank showed how
quickly tile quantisation costs at M / 4).  Measured here with each variant captured into HIP graphs:

  one graph, full batch (the product path's network part)   vs   two half-batch graphs replayed on two streams
                                                             vs   the same two graphs replayed back to back on one stream

usage: python tools/studies/two_stream_cfg_halves.py [s2|s1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (round 4: the split-K scratch is per stream -- hi3d_gemm_set_workspace_for_stream -- so the two streams may both split)
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from hi3d_hip import synth  # noqa: E402
from hi3d_hip.runtime_unet import CIN_PAD  # noqa: E402
from sgm.modules.diffusionmodules.video_model import VideoUNet  # noqa: E402
from sgm.util import ParamTree  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
stage = 1 if (len(sys.argv) > 1 and sys.argv[1] == "s1") else 2
T, lat = 16, (64 if stage == 1 else 128)
cfg = bench.unet_cfg(stage)
ParamTree.skip_init = True
with torch.device(dev):
    unet = VideoUNet(**cfg)
ParamTree.skip_init = False
synth.fill_module_on_device_(unet, seed=1, prefix="model.diffusion_model.")
rt = unet.runtime(dev)
HW = lat * lat
g = torch.Generator(device=dev).manual_seed(3)
tok = torch.randn((2 * T * HW, CIN_PAD), device=dev, generator=g).to(torch.bfloat16)
tok[:, cfg["in_channels"]:] = 0
tvec = torch.full((2 * T,), 0.375, device=dev)
ctx2 = torch.randn((2, 1, cfg["context_dim"]), device=dev, generator=g)
y2 = torch.randn((2, cfg["adm_in_channels"]), device=dev, generator=g)
halves = [tok[:T * HW].contiguous(), tok[T * HW:].contiguous()]


def capture(fn, stream=None):
    """warm up twice, then capture fn() into a graph"""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = fn()
    return gr, out


with torch.no_grad(), torch.cuda.device(dev):
    st_full = rt.clip_consts(ctx2, y2, torch.zeros(2, T, device=dev), 2 * T, T)
    st_half = [rt.clip_consts(ctx2[i:i + 1].contiguous(), y2[i:i + 1].contiguous(), torch.zeros(1, T, device=dev), T, T) for i in range(2)]
    g_full, o_full = capture(lambda: rt.forward_tokens(tok, 2 * T, lat, lat, tvec, st_full, T))
    g_half, o_half = [], []
    for i in range(2):
        gr, o = capture(lambda i=i: rt.forward_tokens(halves[i], T, lat, lat, tvec[:T], st_half[i], T))
        g_half.append(gr); o_half.append(o)
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)

    def t_full():
        g_full.replay()

    def t_serial():
        g_half[0].replay(); g_half[1].replay()

    def t_two_streams():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            g_half[1].replay()
        g_half[0].replay()
        main.wait_stream(side)

    def timeit(fn, n=6):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    res = {}
    for rnd in range(2):
        for name, fn in (("full batch, one graph", t_full), ("two half graphs, one stream", t_serial), ("two half graphs, two streams", t_two_streams)):
            res.setdefault(name, []).append(round(timeit(fn), 2))
    # same results from the concurrent replay as from the serial one (the full-batch graph is not comparable here: the
    # runtime keeps ONE set of per-clip constants, which the three clip_consts calls above overwrote in turn)
    t_serial(); torch.cuda.synchronize()
    ser = [o.clone() for o in o_half]
    t_two_streams(); torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(ser, o_half))
print(f"stage {stage}, {T} views, latent {lat}x{lat}: network part of one step, ms (two rounds each)")
for k, v in res.items():
    print(f"  {k:32s} {v}")
print(f"  two-stream replay bit-identical to the serial replay of the same graphs: {same}")
