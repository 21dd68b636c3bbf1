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
    """/root/reference/trainer/builder.py:73-77: `nms` is optional (no entry -> the default batched_nms)."""
    cfg = dict(config)
    nms_config = cfg.pop("nms", None)
    nms = build_func_partial(nms_config, _eval) if nms_config else None
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
        sd = torch.load(weights, map_location="cpu", weights_only=False) if isinstance(weights, str) else weights
        net.load_state_dict(unwrap_checkpoint(sd), strict=True)
    return net.to(device).eval()


def load_checkpoint(path_or_obj):
    """Reads what the reference's trainer writes (/root/reference/trainer/base.py:143-152):
    {'epoch', 'state_dict', 'optimizer', 'lr_scheduler', 'monitor_best', 'config'} -- or a bare state_dict,
    which infer.py also accepts (/root/reference/infer.py:81-83).  Returns (state_dict, train_config or None)."""
    import torch
    obj = torch.load(path_or_obj, map_location="cpu", weights_only=False) if isinstance(path_or_obj, str) else path_or_obj
    if isinstance(obj, dict) and isinstance(obj.get("state_dict"), dict):
        return obj["state_dict"], obj.get("config")
    return obj, None


def build_tester(config, checkpoint, test_loader, device=None, on_batch=None):
    """trainer/builder.py:43-58 on the HIP path: the MODEL config comes from the checkpoint's own
    train config (builder.py:45,50), weights are loaded strictly (:52), the postprocess from the test config.
    `checkpoint` is a path or an already loaded object; `test_loader` any iterable of batches."""
    import torch
    from .tester import Tester
    state_dict, train_config = load_checkpoint(checkpoint)
    if train_config is None or "model" not in train_config:
        raise ValueError("checkpoint has no 'config' with a 'model' entry (needed by build_tester, builder.py:45)")
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    model = build_model(train_config["model"], device, weights=state_dict)
    postprocess = build_postprocess(config["postprocess"], device=device)
    return Tester(model, postprocess, test_loader, device, on_batch=on_batch)
