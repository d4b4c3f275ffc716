"""Host-side base class of the model wrapper: same public surface as the reference's `BaseModel`
(models/base_model.py:8-125 — device pick, LR warm-up, network / training-state checkpoints), written
for the one-process-per-GPU design.

Checkpoint formats are the reference's, so files interchange both ways:
  `{iter}_G.pth`   CPU state_dict of the unwrapped generator (1332 aliased keys for bin_stage4)
  `{iter}.state`   {"epoch", "iter", "schedulers": [...], "optimizers": [...]}
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

_PREFIXES = ("module.", "InterpNet.")


def unwrap(network):
    """The bare generator behind a DataParallel / DDP / SingleProcessParallel style wrapper."""
    inner = getattr(network, "module", None)
    return inner if isinstance(inner, nn.Module) else network


class _NoCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _direct_param_grads(network):
    """The generator's own `direct_param_grads()` context (bin_amd/models/archs/RDN.py) when it has one — an injected
    CPU test generator does not — else a no-op."""
    net = unwrap(network)
    ctx = getattr(net, "direct_param_grads", None)
    return ctx() if callable(ctx) else _NoCtx()


def clean_state_dict_keys(loaded, strict):
    """Key clean-up of the reference's load_network (base_model.py:93-102): a leading 'module.' or
    'InterpNet.' is stripped.  The reference also keeps a 'module.'-prefixed key under its original name
    (its `else` belongs to the second `if`); that duplicate makes a strict load of a DataParallel-saved
    file impossible, so under strict=True it is dropped when the stripped twin exists."""
    out = OrderedDict()
    for key, value in loaded.items():
        stripped = next((key[len(p):] for p in _PREFIXES if key.startswith(p)), None)
        if stripped is not None:
            out[stripped] = value
        if not key.startswith("InterpNet."):
            out[key] = value
    if strict:
        for key in [k for k in out if k.startswith("module.") and k[len("module."):] in out]:
            del out[key]
    return out


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.is_train = opt["is_train"]
        # reference rule (base_model.py:11): any gpu_ids => 'cuda' (the process's current HIP device), else cpu
        self.device = torch.device("cpu" if opt["gpu_ids"] is None else "cuda")
        # The reference's non-distributed default wraps the generator in nn.DataParallel over `gpu_ids` (bin_model.py:41-42; the
        # shipped yml has gpu_ids: [2, 3] with dist: false).  Here a process drives ONE GPU: several ids without `dist` would
        # silently train on one device at the full batch, so that is an error with the way out spelled out.
        ids = opt["gpu_ids"]
        if ids is not None and len(ids) > 1 and not opt["dist"]:
            raise RuntimeError(
                f"bin_amd: gpu_ids = {list(ids)} with dist = false asks for nn.DataParallel over {len(ids)} GPUs (reference "
                "bin_model.py:41-42); this build runs one process per GPU over RCCL instead.  Launch\n"
                f"  python -m torch.distributed.run --nnodes=1 --nproc-per-node={len(ids)} --master-addr 127.0.0.1 "
                "-m bin_amd.train -opt <yml> --launcher pytorch\n"
                "(each rank takes gpu_ids = [LOCAL_RANK], batch_size is per process), or list a single id.")
        self.optimizers = []
        self.schedulers = []

    # ---- interface stubs the concrete wrapper overrides (kept for API parity) ----------------------
    def feed_data(self, data):
        pass

    def optimize_parameters(self):
        pass

    def get_current_visuals(self):
        pass

    def get_current_losses(self):
        pass

    def print_network(self):
        pass

    def save(self, label):
        pass

    def load(self):
        pass

    # ---- learning rate -----------------------------------------------------------------------------
    def _set_lr(self, lr_groups_l):
        """lr_groups_l[i][j]: learning rate of param group j of optimizer i."""
        for opt_, group_lrs in zip(self.optimizers, lr_groups_l):
            for group, lr in zip(opt_.param_groups, group_lrs):
                group["lr"] = lr

    def _get_init_lr(self):
        """The schedulers record each group's starting rate as 'initial_lr'."""
        return [[g["initial_lr"] for g in opt_.param_groups] for opt_ in self.optimizers]

    def update_learning_rate(self, cur_iter, warmup_iter=-1):
        """Step every scheduler, then override with the linear warm-up ramp while cur_iter < warmup_iter
        (reference base_model.py:51-63)."""
        for sched in self.schedulers:
            sched.step()
        if cur_iter < warmup_iter:
            # same arithmetic order as the reference: initial_lr / warmup_iter * cur_iter
            self._set_lr([[lr0 / warmup_iter * cur_iter for lr0 in grp] for grp in self._get_init_lr()])

    def get_current_learning_rate(self):
        return [g["lr"] for g in self.optimizers[0].param_groups]

    # ---- description / checkpoints -------------------------------------------------------------------
    def get_network_description(self, network):
        net = unwrap(network)
        return str(net), sum(p.numel() for p in net.parameters())

    def save_network(self, network, network_label, iter_label):
        target = os.path.join(self.opt["path"]["models"], f"{iter_label}_{network_label}.pth")
        torch.save(OrderedDict((k, v.cpu()) for k, v in unwrap(network).state_dict().items()), target)

    def load_network(self, load_path, network, strict=True):
        loaded = torch.load(load_path, map_location="cpu")
        net = unwrap(network)
        net.load_state_dict(clean_state_dict_keys(loaded, strict), strict=strict)
        for mod in net.modules():                    # load_state_dict copies in place (versions bump), but be explicit:
            if hasattr(mod, "invalidate_kernel_weights"):        # the relayouted kernel weights must be rebuilt
                mod.invalidate_kernel_weights()

    def save_training_state(self, epoch, iter_step):
        """`{iter}.state` with everything needed to resume (reference base_model.py:105-114)."""
        state = {"epoch": epoch, "iter": iter_step,
                 "schedulers": [s.state_dict() for s in self.schedulers],
                 "optimizers": [o.state_dict() for o in self.optimizers]}
        torch.save(state, os.path.join(self.opt["path"]["training_state"], f"{iter_step}.state"))

    def resume_training(self, resume_state):
        saved_opt, saved_sched = resume_state["optimizers"], resume_state["schedulers"]
        assert len(saved_opt) == len(self.optimizers), "Wrong lengths of optimizers"
        assert len(saved_sched) == len(self.schedulers), "Wrong lengths of schedulers"
        for target, state in zip(self.optimizers, saved_opt):
            target.load_state_dict(state)
        for target, state in zip(self.schedulers, saved_sched):
            target.load_state_dict(state)
