"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

Makes the *reference* Hi3D classes importable in the build container so that
oracle/gen_golden.py can run them on CPU fp32 and dump golden vectors.

The reference tree lives at /root/reference (absent on the GPU box), and its
`sgm/__init__.py` pulls in packages that are not installed here
(pytorch_lightning, omegaconf, kornia, open_clip).  None of them is used by the
hot path (VideoUNet / Denoiser / EulerEDMSampler / AutoencoderKL), so they are
replaced with inert stand-ins before the import.  See SURVEY.md section 8c.
"""
import importlib
import sys
import types

REFERENCE_ROOT = "/root/reference"
sys.dont_write_bytecode = True      # the reference tree is read-only: importing it must not leave __pycache__ behind


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    mod.__path__ = []  # behave like a package so `import a.b` works
    sys.modules[name] = mod
    return mod


def install():
    """Insert stand-in modules and put the reference tree on sys.path."""
    import torch.nn as nn

    if "pytorch_lightning" not in sys.modules:
        class _LM(nn.Module):
            def log(self, *a, **k):
                pass

            def log_dict(self, *a, **k):
                pass

        _stub("pytorch_lightning", LightningModule=_LM, seed_everything=lambda s: None)
    if "omegaconf" not in sys.modules:
        class ListConfig(list):
            pass

        class DictConfig(dict):
            pass

        class OmegaConf:  # placeholder type only
            pass

        _stub("omegaconf", ListConfig=ListConfig, DictConfig=DictConfig, OmegaConf=OmegaConf)
    for name in ("kornia", "open_clip"):
        if name not in sys.modules:
            _stub(name)
    if "xformers" in sys.modules:
        raise RuntimeError("xformers unexpectedly importable; oracle assumes SDPA path")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for clash in ("sgm", "vtdm"):
        m = sys.modules.get(clash)
        if m is not None and not getattr(m, "__file__", "").startswith(REFERENCE_ROOT):
            raise RuntimeError(
                f"module {clash!r} already imported from {m.__file__}; the reference "
                "and the product mirror cannot share one process")


def ref(path):
    """ref('sgm.modules.diffusionmodules.video_model.VideoUNet') -> class"""
    install()
    mod, name = path.rsplit(".", 1)
    return getattr(importlib.import_module(mod), name)


def ref_vtdm_util(name):
    """ref_vtdm_util('tensor2vid') -> the reference's vtdm/util.py function.  That module imports cv2 / imageio / torchvision
    at its top for the mp4 writer (vtdm/util.py:2-5); tensor2vid itself (vtdm/util.py:13-21) uses none of them, so they get the
    same inert stand-ins as the four packages above."""
    install()
    for dep in ("cv2", "imageio", "torchvision"):
        if dep not in sys.modules:
            try:
                importlib.import_module(dep)
            except ImportError:
                _stub(dep)
    return getattr(importlib.import_module("vtdm.util"), name)
