"""Dataset / dataloader factories under the reference's names (data/__init__.py:7-46).

Loader policy, as there: the training loader never shuffles itself (order comes from the sampler or from the dataset's
own shuffled window list), drops the ragged last batch, and — under torch.distributed — gives every rank
batch_size // world_size samples with n_workers workers; single-process runs scale the workers by the number of GPUs
listed in the options.  Every other phase is an in-order loader of single samples."""
import logging

import torch
import torch.utils.data as tud


def _train_loader_shape(dataset_opt, opt):
    """(per-process batch, workers) of the training loader."""
    batch, workers = dataset_opt["batch_size"], dataset_opt["n_workers"]
    if opt["dist"]:
        world = torch.distributed.get_world_size()
        if batch % world:
            raise AssertionError(f"batch_size {batch} is not divisible by the {world} ranks")
        return batch // world, workers
    return batch, workers * max(1, len(opt["gpu_ids"] or [0]))


def create_dataloader(dataset, dataset_opt, opt=None, sampler=None, vscode_debug=False):
    common = dict(shuffle=False, pin_memory=torch.cuda.is_available())
    if dataset_opt["phase"] != "train":
        return tud.DataLoader(dataset, batch_size=1, num_workers=0 if vscode_debug else 1, **common)
    batch, workers = _train_loader_shape(dataset_opt, opt)
    return tud.DataLoader(dataset, batch_size=batch, sampler=sampler, drop_last=True,
                          num_workers=0 if vscode_debug else workers, **common)


def create_dataset(dataset_opt):
    # "synthetic_texture" is a bin_amd extension: clips rendered on the fly with the structure of the reference's windows
    kinds = {"BIN": ("BIN_dataset", "BINDataset"), "synthetic_texture": ("synthetic", "SyntheticTextureDataset")}
    mode = dataset_opt["mode"]
    if mode not in kinds:
        raise NotImplementedError("Dataset [{:s}] is not recognized.".format(mode))
    module, cls = kinds[mode]
    dataset = getattr(__import__(f"{__name__}.{module}", fromlist=[cls]), cls)(dataset_opt)
    logging.getLogger("base").info("Dataset [%s - %s] is created.", cls, dataset_opt["name"])
    return dataset


__all__ = ("create_dataloader", "create_dataset")
