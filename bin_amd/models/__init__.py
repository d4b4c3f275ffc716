"""Model factory with the reference's entry point (models/__init__.py:4-20): `create_model(opt)` picks the wrapper
class from `opt['model']`: 'bin' (the hot path's wrapper) and 'video_base' (the single-tensor API over the same net,
Video_base_model.py).  The reference's other names ('sr', 'srgan') point at modules that are absent there."""
import logging

_log = logging.getLogger("base")


def _wrappers():
    from .bin_model import bin_model
    from .Video_base_model import VideoBaseModel
    return {"bin": bin_model, "video_base": VideoBaseModel}


def create_model(opt):
    kind = opt["model"]
    table = _wrappers()
    if kind not in table:
        raise NotImplementedError("Model [{:s}] not recognized.".format(kind))
    model = table[kind](opt)
    _log.info("Model [{:s}] is created.".format(type(model).__name__))
    return model
