#!/bin/bash
# ffn2 (half-chunk ping-pong FFN) first contact: correctness under a short timeout, then A/B vs the first form
O=gpurun_out/r02d; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_at_size_gpu.py -m gpu -q -x -k ffn --timeout 120 2>&1 | tail -15 > $O/pytest_ffn2.log
cat $O/pytest_ffn2.log
echo "--- ffn2 (default)"; timeout 200 python tools/kbench.py ffn 2>&1 | grep fused | tee $O/kbench_ffn2.log
echo "--- first form + nodrain store (HI3D_FFN_V=1)"; HI3D_FFN_V=1 timeout 200 python tools/kbench.py ffn 2>&1 | grep -v "^$" | tee $O/kbench_ffn1_nodrain.log
