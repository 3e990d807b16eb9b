"""Build hygiene that no numerics test can see: a hot kernel instantiation that spills registers to scratch
still computes the right answer -- 2-3x slower (round 2: one extra runtime branch in the conv K walk put 32
B/lane of scratch into every conv gather and took the step from 234 to 351 ms).  hi3d-official_amd/build.py
records hipcc's kernel-resource-usage remarks; this checks them."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = os.path.join(ROOT, "hi3d-official_amd", "hi3d_hip", "kernel_resources.json")


def test_no_hot_kernel_spills_to_scratch():
    if not os.path.exists(RES):
        pytest.skip("library not built here (python hi3d-official_amd/build.py)")
    res = json.load(open(RES))
    assert len(res) > 40
    gemm = {k: v for k, v in res.items() if "gemm_bf16_kernel" in k}
    assert gemm and any("ffn_geglu" in k for k in res) and any("attn_d64" in k for k in res)
    # one bounded, measured exception (round 6, build._spill_tolerated): the single-stage 128 x 128 tile is held to 128 registers for
    # FOUR blocks per CU and parks <= 32 B per lane outside its K loop
    spilled = {k: v["scratch"] for k, v in res.items()
               if v.get("scratch", 0) and not ("gemm_bf16_kernelILi2ELi4ELi1E" in k and v["scratch"] <= 32)}
    assert not spilled, spilled
    four = [v for k, v in gemm.items() if "ILi2ELi4ELi1E" in k]
    assert four and all(v["vgprs"] <= 128 and v["occupancy"] >= 4 for v in four), four
    # the instantiations the stage-2 step actually dispatches keep two blocks per CU (occupancy 2 with 256 threads)
    for k, v in gemm.items():
        if "ILi2ELi5ELi2E" in k:
            assert v["occupancy"] >= 2 and v["vgprs"] <= 256, (k, v)
    # the ping-pong wide tile the stage-2 convs / QKV / GEGLU run on: 8 waves, two per SIMD
    wide = [v for k, v in gemm.items() if "ILi4ELi10ELi2E" in k and "Lb1E" in k]
    assert wide and all(v["vgprs"] <= 256 and v["occupancy"] >= 2 for v in wide)
