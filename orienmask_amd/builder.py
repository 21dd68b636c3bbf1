"""The reference's type-keyed registry for the inference path.

Mirrors /root/reference/trainer/builder.py:61-77: ``build(cfg, module)`` looks ``cfg['type']`` up in a
module and calls it with the remaining keys; ``build_postprocess`` pops ``nms`` and injects the bound
NMS function as ``nms_func``.  Reference config dicts (/root/reference/config/base.py:219-236) go in
unchanged; unlike the reference, the caller's dicts are not mutated.
"""
import functools

from . import eval as _eval
from . import model as _model


def build(config, module, **kwargs):
    cfg = dict(config)
    cfg.update(kwargs)
    return getattr(module, cfg.pop("type"))(**cfg)


def build_func_partial(config, module, **kwargs):
    cfg = dict(config)
    cfg.update(kwargs)
    return functools.partial(getattr(module, cfg.pop("type")), **cfg)


def build_postprocess(config, device):
    cfg = dict(config)
    nms = build_func_partial(cfg.pop("nms"), _eval)
    return build(cfg, _eval, nms_func=nms, device=device)


def build_model(config, device, weights=None):
    """config['model'] -> HIP-backed model on `device`, optionally loading a reference checkpoint
    ({'state_dict': ...} or a raw state_dict, strict) as infer.py:79-83 does."""
    import torch
    from .pack import unwrap_checkpoint
    cfg = dict(config)
    cfg["pretrained"] = None
    net = build(cfg, _model)
    if weights is not None:
        sd = torch.load(weights, map_location="cpu") if isinstance(weights, str) else weights
        net.load_state_dict(unwrap_checkpoint(sd), strict=True)
    return net.to(device).eval()
