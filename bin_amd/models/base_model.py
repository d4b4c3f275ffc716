"""BaseModel (reference models/base_model.py:8-125): device pick, LR warm-up, checkpoint IO.
Checkpoints are the reference's format: `{iter}_G.pth` = CPU state_dict of the unwrapped generator
(1332 aliased keys), `{iter}.state` = {epoch, iter, schedulers[], optimizers[]}."""
import os
from collections import OrderedDict

import torch
import torch.nn as nn


def unwrap(network):
    """Strip a DataParallel / DDP / SingleProcessParallel wrapper."""
    return network.module if hasattr(network, "module") and isinstance(network.module, nn.Module) else network


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device("cuda" if opt["gpu_ids"] is not None else "cpu")
        self.is_train = opt["is_train"]
        self.schedulers = []
        self.optimizers = []

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

    def _set_lr(self, lr_groups_l):
        """set learning rate for warm-up; lr_groups_l: one list of group lrs per optimizer"""
        for optimizer, lr_groups in zip(self.optimizers, lr_groups_l):
            for param_group, lr in zip(optimizer.param_groups, lr_groups):
                param_group["lr"] = lr

    def _get_init_lr(self):
        return [[v["initial_lr"] for v in o.param_groups] for o in self.optimizers]

    def update_learning_rate(self, cur_iter, warmup_iter=-1):
        for scheduler in self.schedulers:
            scheduler.step()
        if cur_iter < warmup_iter:
            init = self._get_init_lr()
            self._set_lr([[v / warmup_iter * cur_iter for v in grp] for grp in init])

    def get_current_learning_rate(self):
        return [g["lr"] for g in self.optimizers[0].param_groups]

    def get_network_description(self, network):
        network = unwrap(network)
        return str(network), sum(p.numel() for p in network.parameters())

    def save_network(self, network, network_label, iter_label):
        save_path = os.path.join(self.opt["path"]["models"], "{}_{}.pth".format(iter_label, network_label))
        state_dict = unwrap(network).state_dict()
        for key, param in state_dict.items():
            state_dict[key] = param.cpu()
        torch.save(state_dict, save_path)

    def load_network(self, load_path, network, strict=True):
        """Strips 'module.' / 'InterpNet.' prefixes exactly like the reference (base_model.py:89-103,
        including its quirk that a 'module.'-prefixed key is also kept under its original name unless it
        starts with 'InterpNet.')."""
        network = unwrap(network)
        load_net = torch.load(load_path, map_location="cpu")
        clean = OrderedDict()
        for k, v in load_net.items():
            if k.startswith("module."):
                clean[k[7:]] = v
            if k.startswith("InterpNet."):
                clean[k[10:]] = v
            else:
                clean[k] = v
        if strict:
            # the quirk above would make strict loading of a DataParallel-saved file fail on the
            # duplicated 'module.*' keys; drop them when their stripped twin exists
            for k in [k for k in clean if k.startswith("module.") and k[7:] in clean]:
                del clean[k]
        network.load_state_dict(clean, strict=strict)

    def save_training_state(self, epoch, iter_step):
        state = {"epoch": epoch, "iter": iter_step, "schedulers": [], "optimizers": []}
        for s in self.schedulers:
            state["schedulers"].append(s.state_dict())
        for o in self.optimizers:
            state["optimizers"].append(o.state_dict())
        torch.save(state, os.path.join(self.opt["path"]["training_state"], "{}.state".format(iter_step)))

    def resume_training(self, resume_state):
        resume_optimizers = resume_state["optimizers"]
        resume_schedulers = resume_state["schedulers"]
        assert len(resume_optimizers) == len(self.optimizers), "Wrong lengths of optimizers"
        assert len(resume_schedulers) == len(self.schedulers), "Wrong lengths of schedulers"
        for i, o in enumerate(resume_optimizers):
            self.optimizers[i].load_state_dict(o)
        for i, s in enumerate(resume_schedulers):
            self.schedulers[i].load_state_dict(s)
