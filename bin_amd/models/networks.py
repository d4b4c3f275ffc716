"""Generator factory (reference models/networks.py:5-14)."""
from .archs import RDN as RDN_arch


def define_G(opt):
    opt_net = opt["network_G"]
    which_model = opt_net["which_model_G"]
    if which_model == "bin_stage4":
        netG = RDN_arch.bin_stage4_lstm()
        prec = opt_net.get("precision") if hasattr(opt_net, "get") else None
        if prec:
            netG.set_precision(prec)           # bin_amd extension: "f16" | "f16x3"
        bwd = opt_net.get("backward_precision") if hasattr(opt_net, "get") else None
        if bwd:                                # bin_amd extension: "f16" = single-product backward behind an f16x3 forward
            netG.set_backward_precision(bwd)   # stored on THIS network's sub-modules, not process-wide
        if hasattr(opt_net, "get") and opt_net.get("allow_f16_training"):
            # bin_amd extension: opens the gate of bin_amd/autograd.py::train_precision for `precision: f16` training (single fp16
            # products forward AND backward; per-parameter gradients verified to 25 % only, yet on the synthetic deblur /
            # interpolation task it converges like f16x3: profiles/r06_training_modes.md).  An explicit choice, never a default.
            for mod in netG.rdn_modules():
                mod.allow_f16_training = True
    else:
        raise NotImplementedError("Generator model [{:s}] not recognized".format(which_model))
    return netG
