"""Model factory with the reference's entry point (models/__init__.py:4-20): `create_model(opt)` picks the wrapper
class from `opt['model']`.  On the bin_stage4 hot path only 'bin' exists; the reference's other names ('sr', 'srgan',
'video_base') point at modules that are absent or un-importable there."""
import logging

_log = logging.getLogger("base")


def _wrappers():
    from .bin_model import bin_model
    return {"bin": bin_model}


def create_model(opt):
    kind = opt["model"]
    table = _wrappers()
    if kind not in table:
        raise NotImplementedError("Model [{:s}] not recognized.".format(kind))
    model = table[kind](opt)
    _log.info("Model [{:s}] is created.".format(type(model).__name__))
    return model
