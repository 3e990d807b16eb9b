"""(reference: sampling_utils.py:34-35)"""
from ...util import append_dims


def to_d(x, sigma, denoised):
    return (x - denoised) / append_dims(sigma, x.ndim)
