"""torch.ops.hi3d.* -- the dispatcher face of the C ABI (csrc/torch_ops.cpp, TORCH_LIBRARY(hi3d, ...)).

`load()` maps libhi3d_torch.so (built by hi3d-official_amd/build.py, linked against libhi3d_hip.so beside it) into the process; after
that a module written against torch tensors reaches the gfx950 kernels as

    torch.ops.hi3d.self_attention(qkv, B, S, H, scale)      # MemoryEfficientCrossAttention / CrossAttention self-attn
    torch.ops.hi3d.attn_d64(q, k, v, B, H, S_q, S_kv, scale)
    torch.ops.hi3d.attn_temporal(qkv, B, T, S, H, scale)    # VideoTransformerBlock's attention over frames
    torch.ops.hi3d.groupnorm_silu / layernorm / linear / conv3x3 / ffn_geglu

which is the binding INTEGRATION.md (B) shows for the reference's ATTENTION_MODES plug-in point
(sgm/modules/attention.py:457-460).  CUDA (= HIP) dispatch key only: a CPU tensor fails in the dispatcher.  The framework's own
runtimes keep calling the C ABI through ctypes (hi3d_hip/lib.py) -- same entry points, fewer layers.
"""
import os

import torch

from . import lib as _l

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhi3d_torch.so")
OPS = ("self_attention", "attn_d64", "attn_temporal", "groupnorm_silu", "layernorm", "linear", "conv3x3", "ffn_geglu")
_loaded = False


def load():
    global _loaded
    if not _loaded:
        if not os.path.exists(LIB_PATH):
            raise _l.Hi3dError(f"{LIB_PATH} not found: build it with `python hi3d-official_amd/build.py`")
        _l.load()                         # libhi3d_hip.so first (same HIP runtime instance as torch)
        torch.ops.load_library(LIB_PATH)
        _loaded = True
    return torch.ops.hi3d
