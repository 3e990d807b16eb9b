"""Diagnostic: is the d=64 flash attention bitwise repeatable at ragged sequence lengths (S not a multiple of the
256-row query tile / 64-key tile)?  30 launches per case on fixed inputs; counts how many differ from the first."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd"))
from hi3d_hip import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
CASES = ((16, 12, 577, "fused"), (16, 12, 576, "fused"), (16, 12, 640, "fused"), (16, 12, 513, "fused"), (32, 5, 1024, "fused"))
for B, H, S, mode in CASES:
    C = H * 64
    qkv = (torch.randn((B * S, 3 * C), generator=g) * 1.5).to(torch.bfloat16).to(dev)
    outs, vt0 = [], None
    for i in range(60):
        if mode == "fused":
            o = ops.self_attention_fused_qkv(qkv, B, S, H)
        else:
            vt = ops.transpose_v(qkv[:, 2 * C:], B, H, S, 3 * C)
            if mode == "sync between":
                torch.cuda.synchronize()
            if mode == "same vt":
                vt0 = vt if vt0 is None else vt0
                vt = vt0
            o = ops.attention_d64(qkv, qkv[:, C:], vt, B, H, S, S, 3 * C, 3 * C, 64 ** -0.5)
        outs.append(o)
    torch.cuda.synchronize()
    bad = [i for i in range(1, 60) if not torch.equal(outs[i], outs[0])]
    info = ""
    if bad:
        d = (outs[bad[0]].float() - outs[0].float()).abs()
        rows = d.amax(dim=1).nonzero().flatten()
        info = f"; first differing launch {bad[0]}: {rows.numel()} rows differ, tokens (row % S) {sorted(set((rows % S).tolist()))[:12]}, max {d.max().item():.3e}"
    print(f"B={B} H={H} S={S} [{mode}]: {len(bad)} of 59 repeat launches differ from the first{info}")
