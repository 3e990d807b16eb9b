"""On-disk cache of the UNet's re-laid-out weights.

`UNetRuntime._pack` turns a reference `state_dict` (fp32 / fp16, NCHW conv kernels, separate to_q/k/v,
44 emb_layers) into what the gfx950 kernels read (bf16 K-major matrices, fused QKV with the softmax
scale folded into to_q, interleaved GEGLU rows, one stacked emb matrix ...).  For a real checkpoint
that is 6 GB moved to the device and converted on every start; the packed form is 3 GB and needs no
work.  This module stores it next to a fingerprint of what it was made from:

    fingerprint = sha256( PACK_VERSION, cfg, key prefix, every (key, shape, dtype) of the source
                          state_dict, and EVERY BYTE of every tensor )

so that a different checkpoint, a changed config or a changed packing scheme (bump PACK_VERSION
together with pack.py / runtime_unet.py) never loads a stale file -- including a checkpoint that differs
from a cached one in a few rows only (partial fine-tune): hashing 6 GB costs seconds, once, and is far
cheaper than the packing it saves.  The file holds CPU tensors plus the runtime's small index tables
(lists / dicts of str and int) and is read back with `weights_only=True`: no pickled code is executed
from the cache directory.
"""
import hashlib
import json
import os

import torch

PACK_VERSION = 4   # 3: to_q rows of spatial blocks carry softmax scale * log2(e); 4: + the 2x2 phase filters of the up-sampling convs


def fingerprint(state_dict, cfg, prefix=""):
    h = hashlib.sha256()
    h.update(f"hi3d-pack-v{PACK_VERSION}|{prefix}|".encode())
    h.update(json.dumps(cfg, sort_keys=True, default=str).encode())
    for k in sorted(state_dict):
        if not k.startswith(prefix):
            continue
        t = state_dict[k]
        h.update(f"|{k}:{tuple(t.shape)}:{t.dtype}".encode())
        flat = t.detach().reshape(-1)
        if flat.numel():                                           # the whole tensor, bit pattern as stored
            h.update(flat.contiguous().cpu().view(torch.uint8).numpy().tobytes())
    return h.hexdigest()


def cache_path(cache_dir, fp):
    return os.path.join(cache_dir, f"unet-packed-{fp[:24]}.pt")


def save(runtime, path, fp):
    """Write the packed weights and index tables of a UNetRuntime (atomically)."""
    blob = {
        "fingerprint": fp, "pack_version": PACK_VERSION,
        "W": {k: v.detach().to("cpu") for k, v in runtime.W.items()},
        "mix": runtime.mix.detach().to("cpu"),
        "emb_slices": {k: list(v) for k, v in runtime.emb_slices.items()}, "emb_total": int(runtime.emb_total),
        "mix_index": dict(runtime.mix_index), "transformers": [list(t) for t in runtime.transformers],
    }
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = f"{path}.tmp.{os.getpid()}"
    torch.save(blob, tmp)
    os.replace(tmp, path)


def load_into(runtime, path, fp):
    """Fill a UNetRuntime from a cache file.  Returns False (and loads nothing) when the file is
    missing, unreadable, from another packing version or from other weights."""
    if not os.path.exists(path):
        return False
    try:
        blob = torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        return False
    if blob.get("pack_version") != PACK_VERSION or blob.get("fingerprint") != fp:
        return False
    dev = runtime.dev
    runtime.W = {k: v.to(dev) for k, v in blob["W"].items()}
    runtime.mix = blob["mix"].to(dev)
    runtime.emb_slices, runtime.emb_total = {k: tuple(v) for k, v in blob["emb_slices"].items()}, int(blob["emb_total"])
    runtime.mix_index = dict(blob["mix_index"])
    runtime.transformers = [tuple(t) for t in blob["transformers"]]
    return True
