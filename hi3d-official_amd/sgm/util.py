"""Registry helpers: the YAML `target:` plugin boundary of the reference
(/root/reference/sgm/util.py:168-185) plus the few tensor utilities the sampler uses."""
import importlib

import torch


def exists(x):
    return x is not None


def default(val, d):
    if val is not None:
        return val
    return d() if callable(d) else d


def get_obj_from_str(string, reload=False, invalidate_cache=True):
    module_name, _, attr = string.rpartition(".")
    if not module_name:
        raise ValueError(f"'{string}' is not a dotted path")
    if invalidate_cache:
        importlib.invalidate_caches()
    module = importlib.import_module(module_name)
    if reload:
        module = importlib.reload(module)
    return getattr(module, attr)


def instantiate_from_config(config):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    params = config.get("params", None) or {}
    return get_obj_from_str(config["target"])(**dict(params))


def append_dims(x, target_dims):
    extra = target_dims - x.ndim
    if extra < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x.reshape(x.shape + (1,) * extra)


def append_zero(x):
    return torch.cat([x, x.new_zeros([1])])


def disabled_train(self, mode=True):
    return self


def count_params(model, verbose=False):
    n = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {n * 1e-6:.2f} M params.")
    return n


class ParamTree(torch.nn.Module):
    """A module that owns parameters under arbitrary dotted names, so that its
    state_dict reproduces a reference layout without reproducing the reference's module
    classes.  The compute lives in hi3d_hip runtimes that read the parameters."""

    # set to True to allocate parameters without initialising them (a checkpoint / synthetic
    # fill follows anyway; saves touching 1.5 B floats on the host)
    skip_init = False

    def __init__(self, shapes=None, init=None):
        super().__init__()
        for key, shape in (shapes or {}).items():
            self.add(key, shape, init)

    def add(self, key, shape, init=None):
        node, parts = self, key.split(".")
        for part in parts[:-1]:
            if part not in node._modules:
                node.add_module(part, ParamTree())
            node = node._modules[part]
        t = torch.empty(tuple(shape))
        if not ParamTree.skip_init:
            (init or _default_init)(key, t)
        node.register_parameter(parts[-1], torch.nn.Parameter(t, requires_grad=False))


def _default_init(key, t):
    """Cheap stand-in initialisation (real use loads a checkpoint): unit gains, zero
    biases, fan-in scaled uniform weights."""
    with torch.no_grad():
        if key.endswith("mix_factor"):
            t.fill_(0.5)
        elif t.ndim >= 2:
            fan_in = t[0].numel()
            t.uniform_(-1.0, 1.0).mul_((3.0 / fan_in) ** 0.5)
        elif key.endswith("weight"):
            t.fill_(1.0)
        else:
            t.zero_()


def params_key(module, device):
    """Identity of a module's current parameter VALUES for the packed-runtime caches: device, dtype,
    storage of the first parameter, and the sum of every parameter's version counter -- so an in-place
    update of ANY tensor (load_state_dict(strict=False) of a partial checkpoint, merged LoRA / EMA
    weights), a dtype cast or a move re-packs, not only a change of the first tensor."""
    ps = list(module.parameters())
    return (torch.device(device), ps[0].dtype, ps[0].data_ptr(), len(ps), sum(p._version for p in ps))
