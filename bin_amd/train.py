"""Training entry point.  The reference ships the wrapper API (bin_model.feed_data / optimize_parameters /
update_learning_rate / save / save_training_state, data.create_dataset / create_dataloader, DistIterSampler,
options.parse / check_resume) but no train.py; this is the loop those pieces imply.

    python -m bin_amd.train -opt bin_amd/options/bin_stage4_adobe240.yml
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m bin_amd.train -opt X.yml --launcher pytorch

One process per GPU; `datasets.train.batch_size` is the whole-job batch (// world_size per rank, data/__init__.py);
gradients are averaged with one flat RCCL all-reduce per step (bin_model.FlatGradAllReduce).  Iteration-oriented:
the sampler enlarges an epoch `ratio` times so the loader is rebuilt rarely."""
import argparse
import logging
import math
import os
import time

import torch

from .data import create_dataloader, create_dataset
from .data.data_sampler import DistIterSampler
from .models import create_model
from .options import options as option
from .utils import util


def init_dist(backend=None):
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    from .utils import dist_util
    dist.init_process_group(backend or dist_util.default_backend())
    return dist.get_rank(), dist.get_world_size()


def validate(model, loader, step, opt, log, max_batches=None):
    """Mean loss and PSNR/SSIM of the 14 outputs over the validation windows (bin_model.py:427-589 meters)."""
    model.val_loss_AverageMeter()
    model.val_AverageMeter_para()
    save_dir = os.path.join(opt["path"]["val_images"], str(step))
    for i, batch in enumerate(loader):
        if max_batches is not None and i >= max_batches:
            break
        model.feed_data(batch)
        model.test()
        with torch.no_grad():
            loss, loss_list = model.get_loss(ret=1)
        model.val_loss_AverageMeter_update(loss_list, loss)
        save = i < int(opt["train"]["val_save_images"] or 0)
        if save:
            util.mkdir(save_dir)
        psnr, ssim = model.compute_current_psnr_ssim(save=save, name=batch["key"][0], save_path=save_dir)
        model.val_AverageMeter_para_update(psnr, ssim)
    _, psnr_dict, psnr_avg, ssim_avg, loss_avg = model.get_current_log(mode="val")
    log.info("<val iter:%8d> loss %.4e  psnr %.3f dB  ssim %.4f  interp(I7''') psnr %.3f dB", step, loss_avg, psnr_avg,
             ssim_avg, psnr_dict["Ap13"])
    return loss_avg


def main(argv=None, model_factory=None, probe=None):
    """`model_factory(opt)` replaces create_model (the CPU host-logic tests inject the oracle-backed wrapper).
    `probe`: optional dict that receives this rank's window order and first-epoch sampler share (tests)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("-opt", "--opt", type=str, required=True, help="Path to option YAML file.")
    ap.add_argument("--launcher", choices=["none", "pytorch"], default="none")
    ap.add_argument("--local_rank", type=int, default=0)
    ap.add_argument("--max_iter", type=int, default=None, help="stop after this many steps (smoke runs)")
    args = ap.parse_args(argv)
    opt = option.parse(args.opt, is_train=True)

    if args.launcher == "none":
        opt["dist"], rank, world = False, -1, 1
    else:
        opt["dist"] = True
        rank, world = init_dist()
        opt["gpu_ids"] = [int(os.environ.get("LOCAL_RANK", "0"))]

    # ---- resume state, directories, loggers (rank 0 only writes)
    resume_state = None
    if opt["path"].get("resume_state"):
        resume_state = torch.load(opt["path"]["resume_state"], map_location="cpu", weights_only=False)
        option.check_resume(opt, resume_state["iter"])
    if rank <= 0:
        if resume_state is None:
            util.mkdir_and_rename(opt["path"]["experiments_root"])
            util.mkdirs(p for k, p in opt["path"].items()
                        if k in ("models", "training_state", "val_images", "train_images"))
        util.setup_logger("base", opt["path"]["log"], "train_" + opt["name"], screen=True, tofile=True)
    log = logging.getLogger("base")
    if rank <= 0:
        log.info(option.dict2str(opt))
    opt = option.dict_to_nonedict(opt)

    seed = opt["train"]["manual_seed"]
    if seed is None:
        seed = int(time.time()) % 10000
        if opt["dist"]:                     # a time-based seed must still be ONE seed: rank 0's
            import torch.distributed as dist
            box = [seed]
            dist.broadcast_object_list(box, src=0)
            seed = int(box[0])
    # The dataset shuffles its window list with the global RNG (BIN_dataset.list_windows, as the reference does), and
    # DistIterSampler hands every rank a disjoint share of ONE index permutation — which only partitions the data if all
    # ranks map index -> window identically.  So the datasets are built under the rank-INDEPENDENT seed; the ranks'
    # streams diverge (seed + rank: augmentation draws, worker seeds) only after that.
    util.set_random_seed(seed)

    # ---- data
    train_loader = val_loader = sampler = None
    total_iters = total_epochs = 0
    for phase, ds_opt in opt["datasets"].items():
        if phase == "train":
            train_set = create_dataset(ds_opt)
            per_epoch = int(math.ceil(len(train_set) / ds_opt["batch_size"]))
            total_iters = int(opt["train"]["niter"] or per_epoch * int(opt["train"]["epoch"] or 1))
            if opt["dist"]:
                ratio = int(ds_opt["dist_ratio"] or 100)
                sampler = DistIterSampler(train_set, world, rank, ratio)
                total_epochs = int(math.ceil(total_iters / (per_epoch * ratio)))
            else:
                total_epochs = int(math.ceil(total_iters / per_epoch))
            train_loader = create_dataloader(train_set, ds_opt, opt, sampler)
            if rank <= 0:
                log.info("train windows: %d, iters/epoch: %d, epochs: %d, iters: %d", len(train_set), per_epoch,
                         total_epochs, total_iters)
        elif phase == "val":
            try:
                val_loader = create_dataloader(create_dataset(ds_opt), ds_opt, opt, None)
            except (FileNotFoundError, NotADirectoryError):
                log.warning("validation set [%s] not found; validation is skipped", ds_opt["dataroot_LQ"])
    assert train_loader is not None, "the option file has no `train` dataset"
    if probe is not None:
        probe["windows"] = [w[3] for w in train_set.all_paths]
        probe["share"] = list(iter(sampler)) if sampler is not None else list(range(len(train_set)))
        probe["ratio"] = int(opt["datasets"]["train"]["dist_ratio"] or 100)
    util.set_random_seed(seed + max(rank, 0))

    # ---- model
    model = (model_factory or create_model)(opt)
    model.train_AverageMeter()
    start_epoch, step = 0, 0
    if resume_state is not None:
        log.info("Resuming training from epoch: %d, iter: %d.", resume_state["epoch"], resume_state["iter"])
        start_epoch, step = resume_state["epoch"], resume_state["iter"]
        model.resume_training(resume_state)

    plateau = opt["train"]["lr_scheme"] == "ReduceLROnPlateau"
    warmup = int(opt["train"]["warmup_iter"] or -1)
    print_freq = int(opt["logger"]["print_freq"])
    save_freq = int(opt["logger"]["save_checkpoint_freq"])
    val_freq = int(opt["train"]["val_freq"] or 0)
    limit = min(total_iters, args.max_iter) if args.max_iter else total_iters
    per_rank_batch = opt["datasets"]["train"]["batch_size"] // world
    t_print = time.time()
    done = step >= limit
    for epoch in range(start_epoch, total_epochs + 1):
        if done:
            break
        if sampler is not None:
            sampler.set_epoch(epoch)
        for batch in train_loader:
            step += 1
            if step > limit:
                done = True
                break
            if not plateau:
                model.update_learning_rate(step, warmup_iter=warmup)
            model.feed_data(batch)
            model.optimize_parameters(step)
            model.train_AverageMeter_update()
            if step % print_freq == 0 and rank <= 0:
                _, avg = model.get_current_log("train")
                dt = time.time() - t_print
                t_print = time.time()
                log.info("<epoch:%3d, iter:%8d, lr:%.3e> loss %.4e (I7''' %.4e)  %.1f samples/s", epoch, step,
                         model.get_current_learning_rate()[0], avg["Al"], avg["13"],
                         print_freq * per_rank_batch * world / max(dt, 1e-9))
                model.train_AverageMeter_reset()
            if val_loader is not None and val_freq and step % val_freq == 0:
                val_loss = 0.0
                if rank <= 0:
                    val_loss = validate(model, val_loader, step, opt, log, opt["train"]["val_max_batches"])
                if opt["dist"]:                      # every rank steps its scheduler with rank 0's value
                    t = torch.tensor([val_loss], dtype=torch.float64, device=model.device)
                    torch.distributed.broadcast(t, src=0)
                    val_loss = float(t.item())
                if plateau:
                    for sched in model.schedulers:
                        sched.step(val_loss)
            if step % save_freq == 0 and rank <= 0:
                log.info("Saving models and training states.")
                model.save(step)
                model.save_training_state(epoch, step)
    if rank <= 0:
        log.info("Saving the final model.")
        model.save("latest")
        log.info("End of training.")
    if opt["dist"]:
        torch.distributed.barrier()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
