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
