"""Dataset / dataloader factories (reference data/__init__.py:7-46)."""
import logging

import torch
import torch.utils.data


def create_dataloader(dataset, dataset_opt, opt=None, sampler=None, vscode_debug=False):
    """train: per-rank batch = batch_size // world (dist) or the full batch (single process, workers x gpus),
    never shuffled by the loader (the sampler or the dataset's own shuffled window list does it), drop_last;
    other phases: batch 1, in order."""
    pin = torch.cuda.is_available()
    if dataset_opt["phase"] == "train":
        if opt["dist"]:
            world = torch.distributed.get_world_size()
            assert dataset_opt["batch_size"] % world == 0
            batch, workers = dataset_opt["batch_size"] // world, dataset_opt["n_workers"]
        else:
            batch, workers = dataset_opt["batch_size"], dataset_opt["n_workers"] * max(1, len(opt["gpu_ids"] or [0]))
        return torch.utils.data.DataLoader(dataset, batch_size=batch, shuffle=False, sampler=sampler, drop_last=True,
                                           num_workers=0 if vscode_debug else workers, pin_memory=pin)
    return torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False, num_workers=0 if vscode_debug else 1,
                                       pin_memory=pin)


def create_dataset(dataset_opt):
    if dataset_opt["mode"] != "BIN":
        raise NotImplementedError("Dataset [{:s}] is not recognized.".format(dataset_opt["mode"]))
    from .BIN_dataset import BINDataset
    dataset = BINDataset(dataset_opt)
    logging.getLogger("base").info("Dataset [%s - %s] is created.", type(dataset).__name__, dataset_opt["name"])
    return dataset


__all__ = ("create_dataloader", "create_dataset")
