"""MI355X-native mirror of the reference's `sgm` package for the Hi3D denoising hot path.

Only the dotted paths the Hi3D inference YAMLs name as `target:` (and what the two
pipeline scripts call) exist here; each class keeps the reference's constructor
arguments, forward signature and state_dict keys, while the arithmetic runs in
libhi3d_hip.so (hand-written gfx950 kernels) through `hi3d_hip`.
"""
from .util import get_obj_from_str, instantiate_from_config  # noqa: F401
