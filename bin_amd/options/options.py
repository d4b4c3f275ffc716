"""YAML option files (reference options/options.py:9-128): `parse` fills in the derived entries the wrappers read
(`is_train`, per-dataset `phase`/`data_type`/`scale`, the experiment / results directory tree under
`path.save_path`), `dict_to_nonedict` makes absent keys read as None, `check_resume` points the pretrain path at
the checkpoint that belongs to a resume state."""
import logging
import os
import os.path as osp
from collections import OrderedDict

import yaml


def _ordered_loader():
    class Loader(yaml.SafeLoader):
        pass

    Loader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG,
                           lambda loader, node: OrderedDict(loader.construct_pairs(node)))
    return Loader


def parse(opt_path, is_train=True):
    with open(opt_path) as f:
        opt = yaml.load(f, Loader=_ordered_loader())
    if is_train and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # the reference exports CUDA_VISIBLE_DEVICES from gpu_ids (torch on ROCm honours the same variable); under
        # a one-process-per-GPU launcher the launcher owns device visibility, so it is left alone there
        os.environ["CUDA_VISIBLE_DEVICES"] = ",".join(str(g) for g in opt["gpu_ids"])
    opt["is_train"] = is_train
    sr = opt["distortion"] == "sr"
    scale = opt["scale"] if sr else 1

    for phase, ds in opt["datasets"].items():
        ds["phase"] = phase
        if sr:
            ds["scale"] = scale
        lmdb = False
        for key in ("dataroot_GT", "dataroot_LQ"):
            if ds.get(key) is not None:
                ds[key] = osp.expanduser(ds[key])
                lmdb = lmdb or ds[key].endswith("lmdb")
        ds["data_type"] = "lmdb" if lmdb else "img"
        if ds["mode"].endswith("mc"):
            ds["data_type"] = "mc"
            ds["mode"] = ds["mode"].replace("_mc", "")

    paths = opt["path"]
    for key, value in paths.items():
        if value and key != "strict_load":
            paths[key] = osp.expanduser(value)
    paths["root"] = paths["save_path"]
    if is_train:
        exp = osp.join(paths["root"], "experiments", opt["name"])
        paths.update(experiments_root=exp, models=osp.join(exp, "models"),
                     training_state=osp.join(exp, "training_state"), log=exp,
                     val_images=osp.join(exp, "val_images"), train_images=osp.join(exp, "train_images"))
        if "debug" in opt["name"]:
            opt["train"]["val_freq"] = 1
            opt["logger"]["print_freq"] = 1
            opt["logger"]["save_checkpoint_freq"] = 1
    else:
        results = osp.join(paths["root"], "results", opt["name"])
        paths.update(results_root=results, log=results)
    if sr:
        opt["network_G"]["scale"] = scale
    return opt


def dict2str(opt, indent_l=1):
    """Indented dump of a (nested) option dict for the log."""
    pad = " " * (indent_l * 2)
    out = []
    for k, v in opt.items():
        if isinstance(v, dict):
            out.append(f"{pad}{k}:[\n{dict2str(v, indent_l + 1)}{pad}]\n")
        else:
            out.append(f"{pad}{k}: {v}\n")
    return "".join(out)


class NoneDict(dict):
    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{k: dict_to_nonedict(v) for k, v in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt


def check_resume(opt, resume_iter):
    """When resuming, the generator weights come from `<models>/<iter>_G.pth`, whatever pretrain path was set."""
    log = logging.getLogger("base")
    if not opt["path"]["resume_state"]:
        return
    if opt["path"].get("pretrain_model_G") is not None or opt["path"].get("pretrain_model_D") is not None:
        log.warning("pretrain_model path will be ignored when resuming training.")
    opt["path"]["pretrain_model_G"] = osp.join(opt["path"]["models"], f"{resume_iter}_G.pth")
    log.info("Set [pretrain_model_G] to " + opt["path"]["pretrain_model_G"])
    if "gan" in opt["model"]:
        opt["path"]["pretrain_model_D"] = osp.join(opt["path"]["models"], f"{resume_iter}_D.pth")
        log.info("Set [pretrain_model_D] to " + opt["path"]["pretrain_model_D"])
