#!/usr/bin/env python
"""Where does the persistent wide tile (csrc/gemm_persist.hip) differ from the one-block-per-tile form?  Per 256 x 320 tile."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd"))
import torch
from hi3d_hip import ops
dev = torch.device("cuda:0")
os.environ["HI3D_GEMM_VARIANT"] = "7"
os.environ["HI3D_GEMM_PERSIST_MIN"] = "0"
g = torch.Generator(device=dev).manual_seed(1)
def rb(*s): return (torch.randn(s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
M, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 2560, 960, 320
A, W = rb(M, K), rb(N, K)
bias = torch.randn(N, device=dev, generator=g)
R1 = rb(M, N)
for grid in (64, 8, 3):
    os.environ["HI3D_GEMM_PERSIST_GRID"] = str(grid)
    for name, kw in (("plain", {}), ("bias", dict(bias=bias)), ("bias+R1", dict(bias=bias, R1=R1)), ("geglu", dict(bias=bias, geglu=True))):
        os.environ["HI3D_GEMM_PERSIST"] = "0"
        a = ops.gemm(A, W, M=M, N=N, K=K, **kw).float()
        os.environ["HI3D_GEMM_PERSIST"] = "7"
        b = ops.gemm(A, W, M=M, N=N, K=K, **kw).float()
        torch.cuda.synchronize()
        d = (a - b).abs()
        tn = 160 if "geglu" in kw else 320
        bad = []
        for tm in range((M + 255) // 256):
            for t in range((d.shape[1] + tn - 1) // tn):
                blk = d[tm * 256:(tm + 1) * 256, t * tn:(t + 1) * tn]
                if blk.numel() and blk.max() > 0:
                    rows = (blk.max(1).values > 0).nonzero().flatten()
                    cols = (blk.max(0).values > 0).nonzero().flatten()
                    bad.append((tm, t, float(blk.max()), int(rows.min()), int(rows.max()), len(rows), int(cols.min()), int(cols.max()), len(cols)))
        print(f"grid {grid:3d} {name:8s}: {len(bad)} bad tiles of {((M + 255) // 256) * ((d.shape[1] + tn - 1) // tn)}", bad[:6])
