#!/bin/bash
# ffn2 as the default: kernel + at-size + UNet/pipeline suites, then the stage-2 step with either form
O=gpurun_out/r02d; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_at_size_gpu.py tests/test_unet_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -6 | tee $O/pytest_ffn2_default.log
python bench.py --config s2 --steps 8 --warmup 3 --no-cpu-baseline > $O/s2_ffn2.json 2> $O/s2_ffn2.log; cut -c1-260 $O/s2_ffn2.json
HI3D_FFN_V=1 python bench.py --config s2 --steps 8 --warmup 3 --no-cpu-baseline > $O/s2_ffn1.json 2> $O/s2_ffn1.log; cut -c1-260 $O/s2_ffn1.json
