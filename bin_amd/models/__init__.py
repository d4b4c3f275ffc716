"""create_model (reference models/__init__.py:4-20).  Only `model: bin` exists on the hot path."""
import logging

logger = logging.getLogger("base")


def create_model(opt):
    model = opt["model"]
    if model == "bin":
        from .bin_model import bin_model as M
    else:
        raise NotImplementedError("Model [{:s}] not recognized.".format(model))
    m = M(opt)
    logger.info("Model [{:s}] is created.".format(m.__class__.__name__))
    return m
