"""hi3d_hip -- Python host side of libhi3d_hip.so (gfx950 kernels for the Hi3D
denoising hot path).  `lib` is the raw ctypes binding of include/hi3d_hip.h,
`ops` wraps it for torch device tensors (memory + streams only; no torch math)."""
from . import lib  # noqa: F401
