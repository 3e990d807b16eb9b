#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
python - <<'PY' 2>&1 | grep -v amdgpu.ids > $O/geglu_5_vs_7.log
import os, sys
sys.argv = ["kbench.py", "none"]
sys.path.insert(0, "tools")
import kbench as kb
import torch
from hi3d_hip import ops
for M, N, K in ((131072, 5120, 640), (32768, 10240, 1280), (8192, 10240, 1280), (524288, 2560, 320)):
    A, W = kb.rb(M, K), kb.rb(N, K)
    bias = torch.randn(N, device=kb.dev)
    out = torch.empty((M, N // 2), device=kb.dev, dtype=torch.bfloat16)
    best = {}
    for rnd in range(3):
        for v in ("3", "5", "7"):
            os.environ["HI3D_GEMM_VARIANT"] = v
            ms = kb.timeit(lambda: ops.gemm(A, W, M=M, N=N, K=K, bias=bias, geglu=True, out=out), iters=6, warm=2)
            best[v] = min(best.get(v, 1e9), ms)
    print(f"geglu M={M} N={N} K={K}: " + "  ".join(f"{v}: {best[v]:.3f} ms {2.0*M*N*K/best[v]/1e9:.0f} TF" for v in best))
PY
cat $O/geglu_5_vs_7.log
