import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hi3d-official_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def shrink_conditioner(y):
    """Reduced CLIP towers in a loaded inference YAML (2 layers instead of 32 / 24: the full ones hold 0.94 B parameters);
    same classes, key names and output widths."""
    for e in y["model"]["params"]["conditioner_config"]["params"]["emb_models"]:
        if e["target"].endswith("FrozenOpenCLIPImagePredictionEmbedder"):
            e["params"]["open_clip_embedding_config"]["params"]["arch"] = "ViT-tiny-H"
        elif e["target"].endswith("AesEmbedder"):
            e.setdefault("params", {})["arch"] = "ViT-tiny-L"
    return y


_UNET_WEIGHTS = {}          # (state_dict key, shape, seed) -> fp32 tensor on the device: drawn once per session


def synth_unet(fx, dev):
    """A fresh VideoUNet(**fx["cfg"]) on `dev` carrying the fixture's seeded weights (hi3d_hip.synth: the CPU stream, the
    same values the reference modules got when the golden was made).  Every test gets its OWN module (its own runtime,
    built under the test's environment); what is shared for the session is the drawn state dict, kept on the device --
    drawing 1.5 B parameters on the host and torch's default init of a module that is overwritten anyway were half of the
    GPU suite's wall time."""
    import torch
    from hi3d_hip import synth
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from sgm.util import ParamTree
    ParamTree.skip_init = True
    try:
        with torch.device(dev):
            m = VideoUNet(**fx["cfg"])
    finally:
        ParamTree.skip_init = False
    sd = {}
    for k, v in m.state_dict().items():           # per TENSOR, so the stage-1 / stage-2 networks (same keys but the input conv) share
        if v.dtype.is_floating_point:
            ck = (fx["key_prefix"] + k, tuple(v.shape), fx["weight_seed"], str(dev))
            if ck not in _UNET_WEIGHTS:
                _UNET_WEIGHTS[ck] = synth.synth_tensor(ck[0], v.shape, fx["weight_seed"]).to(dev)
            sd[k] = _UNET_WEIGHTS[ck]
    m.load_state_dict(sd, strict=False)
    return m
