"""Diagnostic: which op of the DPT-hybrid forward first gives different results for identical frames at different
batch positions (frames 0 and 4 of the batch are the same image)?  Wraps the ops the runtime calls and compares the
two frames' slices of every output whose leading dimension is a multiple of the batch."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd"))
from hi3d_hip import ops, synth  # noqa: E402
from hi3d_hip.runtime_dpt import DPTHybridRuntime, dpt_hybrid_shapes  # noqa: E402

dev = torch.device("cuda:0")
N = 16
sd = synth.damp_residual_tails(synth.synth_state_dict(dpt_hybrid_shapes("m."), 5), 0.25)
rt = DPTHybridRuntime(sd, "m.", dev)
g = torch.Generator().manual_seed(31)
base = F.interpolate(torch.rand((4, 3, 32, 32), generator=g), (1024, 1024), mode="bilinear") * 2 - 1
x0 = base[[0, 1, 2, 3, 0, 1, 2, 3, 3, 2, 1, 0, 0, 0, 1, 1]].to(dev)
x = ops.resize_bilinear(x0.permute(0, 2, 3, 1).contiguous(), 384, 384, align_corners=False)
log, count = [], [0]


keep = []


def wrap(name):
    fn = getattr(ops, name)

    def w(*a, **k):
        out = fn(*a, **k)
        count[0] += 1
        if torch.is_tensor(out) and "out" not in k and out.shape[0] % N == 0 and out.shape[0] >= N:
            desc = {kk: vv for kk, vv in k.items() if kk in ("M", "N", "K", "conv3x3", "out_fp32")}
            desc["args"] = [tuple(v.shape) if torch.is_tensor(v) else v for v in a[:6]]
            keep.append((count[0], name, desc, out.clone()))          # (no host sync here: compared after the pass)
        return out
    setattr(ops, name, w)


for n in ("gemm", "groupnorm_silu", "add_act", "act_", "pool2", "resize_bilinear", "self_attention_fused_qkv", "layernorm",
          "dpt_stem_conv", "dpt_head_out"):
    wrap(n)
pairs = ((0, 4), (0, 11), (0, 12), (1, 5), (2, 9), (3, 8))
for rep in range(3):
    keep.clear(); count[0] = 0
    d = rt.forward_nhwc(x)
    torch.cuda.synchronize()
    bad = []
    for num, name, desc, t in keep:
        r = t.reshape(N, -1)
        b = [(i, j) for i, j in pairs if not torch.equal(r[i], r[j])]
        if b:
            dm = max((r[i].float() - r[j].float()).abs().max().item() for i, j in b)
            bad.append(f"op#{num} {name} out{tuple(t.shape)} {desc} pairs {b} maxdiff {dm:.3e}")
    print(f"run {rep}: {count[0]} ops, {len(bad)} of {len(keep)} checked outputs with unequal twin frames; depth twins equal: {torch.equal(d[0], d[4])}")
    for line in bad[:4]:
        print("   ", line)
