from .encoders.modules import GeneralConditioner  # noqa: F401

UNCONDITIONAL_CONFIG = {
    "target": "sgm.modules.GeneralConditioner",
    "params": {"emb_models": []},
}
