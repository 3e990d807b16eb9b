"""Resident blocks per CU the HIP runtime reports for the hot kernels (run on the GPU box)."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd"))
import torch  # noqa: E402,F401  (initialises the HIP runtime the library shares)
from hi3d_hip import lib as _l  # noqa: E402
lib = _l.load()
torch.zeros(1, device="cuda")
lib.hi3d_debug_attn_occupancy.restype = ctypes.c_int
lib.hi3d_debug_gemm_occupancy.restype = ctypes.c_int
print("attn_d64<pre>  blocks/CU:", lib.hi3d_debug_attn_occupancy(0))
print("attn_d64<scale> blocks/CU:", lib.hi3d_debug_attn_occupancy(1))
for wm, nt, ns in ((2, 5, 2), (2, 4, 2), (4, 5, 3), (4, 4, 3)):
    print(f"gemm<{wm},{nt},{ns}> blocks/CU:", lib.hi3d_debug_gemm_occupancy(wm, nt, ns))
