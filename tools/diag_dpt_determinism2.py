"""Diagnostic 2: the DepthEmbedder path at the shipped size, stage by stage, twin frames compared bitwise, 3 runs."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd"))
from hi3d_hip import ops, synth  # noqa: E402
from hi3d_hip.runtime_dpt import DPTHybridRuntime, dpt_hybrid_shapes  # noqa: E402

dev = torch.device("cuda:0")
sd = synth.damp_residual_tails(synth.synth_state_dict(dpt_hybrid_shapes("m."), 5), 0.25)
rt = DPTHybridRuntime(sd, "m.", dev)
g = torch.Generator().manual_seed(31)
base = F.interpolate(torch.rand((4, 3, 32, 32), generator=g), (1024, 1024), mode="bilinear") * 2 - 1
idx = [0, 1, 2, 3, 0, 1, 2, 3, 3, 2, 1, 0, 0, 0, 1, 1]
x = base[idx].to(dev)
pairs = [(i, j) for i in range(16) for j in range(i + 1, 16) if idx[i] == idx[j]]


def twins(name, t):
    r = t.reshape(16, -1).float()
    bad = [(i, j, (r[i] - r[j]).abs().max().item()) for i, j in pairs if not torch.equal(r[i], r[j])]
    print(f"   {name}: {len(bad)} of {len(pairs)} twin pairs differ" + (f", e.g. {bad[:4]}" if bad else ""))


for rep in range(3):
    print("run", rep)
    twins("input", x)
    xh = x.permute(0, 2, 3, 1).contiguous()
    y = ops.resize_bilinear(xh, 384, 384, align_corners=False)
    twins("resized input", y)
    d, layers = rt.forward_nhwc(y, return_layers=True)
    for i, l in enumerate(layers):
        twins(f"layer_{i + 1}", l)
    twins("depth", d)
    d2 = ops.resize_bilinear(d.reshape(16, 384, 384, 1), 384, 384, align_corners=False).reshape(16, 384, 384)
    twins("resized depth", d2)
    twins("normalised", ops.depth_normalize_unshuffle(d2, 3))
    torch.cuda.synchronize()
