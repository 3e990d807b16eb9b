"""`create_model(config_path)` (reference: vtdm/model.py:9-28).  omegaconf is not a
dependency here: the YAML is parsed with PyYAML into plain dicts, which is all
instantiate_from_config needs."""
import os

import torch
import yaml

from sgm.util import instantiate_from_config


def get_state_dict(d):
    return d.get("state_dict", d)


def load_state_dict(ckpt_path, location="cpu"):
    if os.path.splitext(ckpt_path)[1].lower() == ".safetensors":
        import safetensors.torch
        sd = safetensors.torch.load_file(ckpt_path, device=location)
    else:
        sd = get_state_dict(torch.load(ckpt_path, map_location=torch.device(location)))
    print(f"Loaded state_dict from [{ckpt_path}]")
    return get_state_dict(sd)


def create_model(config_path):
    with open(config_path) as fh:
        config = yaml.safe_load(fh)
    model = instantiate_from_config(config["model"]).cpu()
    print(f"Loaded model config from [{config_path}]")
    return model
