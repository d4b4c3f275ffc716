"""Evaluate / run bin_stage4 over folders of blurry frames: the reference's test.py (with --gt_path: PSNR/SSIM of the
interpolated and deblurred frames against the sharp ground truth) and demo.py (without) in one entry point.

    python -m bin_amd.test --netName bin_stage4 --input_path DATA/test_blur --gt_path DATA/test \\
        --output_path OUT --opt bin_amd/options/bin_stage4_adobe240.yml [--time_step 0.5]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m bin_amd.test ... --launcher pytorch

What the reference does per input frame `index` of a clip (test.py:236-402) and this keeps:
  * the six blurry frames index + [-2..3], clamped to the clip (test.py:257-261, 333);
  * replicate padding to the next multiple of 128, or 32 px per side when already one (test.py:348-366);
  * outputs Ft_p[13] -> <num+8>.png (the interpolated frame), Ft_p[8] -> <num+4>.png and, except for the last
    window, Ft_p[12] -> <num+12>.png (the deblurred frames), `num` being the frame's own file number
    (test.py:286-293, 380-402); files that already exist are not recomputed;
  * tensor -> image: clamp [0,1], x255, round, RGB->BGR uint8 (util.py:113-137), done on the device.
What is different: PNG decode/encode and the metrics run on a thread pool beside the
GPU work instead of in line; decoded frames and their padded device copies are cached across the 5-of-6 overlap of
consecutive windows; stage-1 RDN results are reused across consecutive windows (exact); and with --launcher
pytorch the flattened (clip, frame) window list is sharded contiguously over the ranks, per-rank metric sums being
combined with one all-reduce at the end."""
import argparse
import logging
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import harness, ops
from .data import util as data_util
from .models import create_model
from .options import options as option
from .utils import dist_util, util
from .utils.util import AverageMeter

OUT_KEYS = (13, 8, 12)         # interpolated, first deblurred, second deblurred (test.py:380-382)
METRICS = ("interp_psnr", "interp_ssim", "interp_err", "deblur_psnr", "deblur_ssim", "blurry_psnr", "blurry_ssim")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--netName", type=str, default="bin_stage4")
    p.add_argument("--input_path", type=str, required=True)
    p.add_argument("--gt_path", type=str, default=None, help="sharp frames; omit for demo mode (no metrics)")
    p.add_argument("--output_path", type=str, required=True)
    p.add_argument("--gpu_id", type=str, default=None)
    p.add_argument("--time_step", type=float, default=0.5)
    p.add_argument("--opt", type=str, required=True, help="Path to option YAML file.")
    p.add_argument("--launcher", choices=["none", "pytorch"], default="none")
    p.add_argument("--precision", choices=["f16", "f16x3"], default=None)
    p.add_argument("--io_threads", type=int, default=8)
    p.add_argument("--batch", type=int, default=1, help="windows per forward (small frames: 8 fills the chip; no "
                   "cross-window stage-1 reuse when > 1)")
    p.add_argument("--no_reuse", action="store_true", help="recompute the stage-1 calls shared by consecutive windows")
    p.add_argument("--backend", default=None, help="torch.distributed backend for --launcher pytorch (default: nccl = "
                   "RCCL on a GPU box; gloo lets several ranks share one device)")
    p.add_argument("--manifest", action="store_true", help="each rank also writes written.rank<R>.txt under the result "
                   "folder: the files IT saved (shard-ownership audit)")
    p.add_argument("--ssim", action="store_true", help="also compute SSIM (host, ~0.2 s per 720p frame)")
    return p.parse_args(argv)


def list_windows(input_path):
    """[(clip, frame names, index)] for every window of every clip, in the reference's order."""
    wins = []
    for clip in sorted(os.listdir(input_path)):
        frames = sorted(f for f in os.listdir(os.path.join(input_path, clip)) if data_util.is_image_file(f))
        wins += [(clip, frames, i) for i in range(len(frames) - 1)]
    return wins


def output_names(frames, index):
    """(interp, first deblur, second deblur or None) file names of window `index` (test.py:286-293, 311-312, 394)."""
    num = int(frames[index][:-4])
    name = lambda n: str(n).zfill(5) + ".png"
    return name(num + 8), name(num + 4), (name(num + 12) if index < len(frames) - 2 else None)


class _Sums:
    """Thread-safe per-clip / total metric accumulators."""

    def __init__(self):
        self.lock = threading.Lock()
        self.total = {k: [0.0, 0] for k in METRICS}
        self.clips = {}

    def add(self, clip, key, value):
        with self.lock:
            for d in (self.total, self.clips.setdefault(clip, {k: [0.0, 0] for k in METRICS})):
                d[key][0] += float(value)
                d[key][1] += 1


def _score(sums, clip, kind, img_bgr, gt_path, want_ssim):
    if gt_path is None or not os.path.exists(gt_path):
        return
    gt = data_util.imread_u8(gt_path)[:, :, :3]
    sums.add(clip, kind + "_psnr", util.calculate_psnr(img_bgr, gt))
    if kind == "interp":
        sums.add(clip, "interp_err", np.mean(np.abs(img_bgr.astype(np.float64) - gt.astype(np.float64))))
    if want_ssim:
        sums.add(clip, kind + "_ssim", util.calculate_ssim(img_bgr, gt))


def main(argv=None, stats=None):
    """`stats` (optional dict, filled on rank 0): windows run, wall seconds incl. all IO, net + glue seconds per window —
    what bench.py's `harness` leg reports."""
    args = parse_args(argv)
    opt = option.parse(args.opt, is_train=False)
    if args.launcher == "pytorch":
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group(args.backend or dist_util.default_backend())
        rank, world = dist.get_rank(), dist.get_world_size()
        opt["gpu_ids"] = [local]
    else:
        rank, world = 0, 1
        if args.gpu_id is not None:
            torch.cuda.set_device(int(args.gpu_id))
            opt["gpu_ids"] = [int(args.gpu_id)]
    opt["dist"] = False                      # inference needs no gradient sync; ranks only share the window list
    if args.precision:
        opt["network_G"]["precision"] = args.precision
    opt = option.dict_to_nonedict(opt)

    n_out = round(1 / args.time_step)
    assert n_out == 2, "bin_stage4 interpolates the middle frame (time_step 0.5), as the reference asserts"
    result_root = os.path.join(args.output_path, f"{n_out * 30}fps_test_results", opt["name"])
    os.makedirs(result_root, exist_ok=True)
    if rank == 0:
        util.setup_logger("base", result_root, "test", screen=True, tofile=True)
    log = logging.getLogger("base")

    # Ramp (round 5): the window list is known before the network exists, so the decoder pool starts on the FIRST windows'
    # frames now and works while the model is constructed and its weights are laid out (into plain host arrays: the
    # page-locked staging buffers need the device context, which the main thread is about to create).
    windows = list_windows(args.input_path)
    begin, end = harness.shard_windows(len(windows), rank, world)
    pool = ThreadPoolExecutor(max_workers=max(2, args.io_threads))
    strip_pool = ThreadPoolExecutor(max_workers=max(4, args.io_threads))       # leaf tasks only (they never wait): no deadlock
    early = {}
    if begin < end:
        clip0, frames0, index0 = windows[begin]
        for ahead in range(0, 3):
            if begin + ahead < end and windows[begin + ahead][0] == clip0:
                # (the window's OWN index: list_windows need not hand out consecutive indices — advisor r05)
                for fid in harness.window_frame_ids(windows[begin + ahead][2], len(frames0)):
                    if (clip0, fid) not in early:
                        early[(clip0, fid)] = pool.submit(data_util.imread_u8, os.path.join(args.input_path, clip0, frames0[fid]))

    # ... and so do the page-locked buffers of the pipeline (a dozen decode staging images, ten 3-image output buffers):
    # hipHostMalloc costs 1-3 ms apiece, which the first ten windows used to pay in line
    stage_pool, stage_lock, pinned = [], threading.Lock(), []

    first_early = next(iter(early.values()), None)

    def preallocate():
        # best effort on a daemon thread: a failure here (unreadable first frame, no page-locked memory left) must be SAID — the
        # pipeline then allocates its buffers in line as before round 5 — not die silently with the thread (advisor r05)
        try:
            if first_early is None:
                return
            first = first_early.result()
            if first.ndim != 3 or first.shape[2] != 3:
                return
            for _ in range(12):
                buf = torch.empty(first.shape, dtype=torch.uint8, pin_memory=True)
                with stage_lock:
                    stage_pool.append(buf)
            for _ in range(10):
                pinned.append(torch.empty((3,) + tuple(first.shape), dtype=torch.uint8, pin_memory=True))
        except Exception as e:
            logging.getLogger("base").warning("bin_amd.test: pre-allocation of the page-locked buffers skipped (%s: %s)",
                                              type(e).__name__, e)
    prealloc = threading.Thread(target=preallocate, daemon=True)
    if torch.cuda.is_available():
        prealloc.start()

    model = create_model(opt)
    netG = model.netG
    netG.eval()
    dev = next(netG.parameters()).device
    inner = netG.module if hasattr(netG, "module") else netG
    reuse = (not args.no_reuse) and getattr(inner, "reuse_schedule", False)
    log.info("In Data: %s | model: %s | parameters: %d | ranks: %d | stage-1 reuse: %s", args.input_path,
             opt["path"]["pretrain_model_G"], sum(p.numel() for p in netG.parameters() if p.requires_grad), world, reuse)

    sums = _Sums()
    copy_stream = torch.cuda.Stream(device=dev)
    decoded, frames_dev, stage1_cache, pending = {}, {}, {}, []
    geom = None                                        # (h, w, pads) of the current clip
    claimed = set()                                    # (`pinned`: recycled page-locked output buffers, pre-allocated above)
    cur_clip, timer = None, AverageMeter()

    # Decoded frames reach the device WITHOUT ever blocking the launching thread (round 4): a copy from pageable memory is
    # stream-ordered but host-synchronous — the host sat in it until the previous window's kernels had drained, and the GPU
    # then idled for the ~5 ms the host needs to queue the next window (steady state 0.90 of the in-HBM rate).  Now the decoder
    # thread leaves the image in a recycled page-locked buffer and the upload runs on its own stream behind an event.
    upload_stream = torch.cuda.Stream(device=dev)
    in_flight = []
    if prealloc.ident is not None:                     # (started)
        prealloc.join()

    def decode(path, ready=None):                      # (worker thread)
        img = data_util.imread_u8(path) if ready is None else ready.result()
        src = torch.from_numpy(img)
        with stage_lock:
            buf = next((b for b in stage_pool if b.shape == src.shape), None)
            if buf is not None:
                stage_pool.remove(buf)
        if buf is None:
            buf = torch.empty(src.shape, dtype=torch.uint8, pin_memory=True)
        buf.copy_(src)
        return buf

    def upload(buf):
        """page-locked HWC uint8 image -> device tensor on the upload stream; the compute stream waits on the event."""
        with torch.cuda.stream(upload_stream):
            g = buf.to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        torch.cuda.current_stream(dev).wait_event(ev)
        g.record_stream(torch.cuda.current_stream(dev))
        in_flight.append((ev, buf))
        while in_flight and in_flight[0][0].query():     # finished uploads hand their staging buffer back
            with stage_lock:
                stage_pool.append(in_flight.pop(0)[1])
        return g

    def want(clip, frames, fid):                       # async decode, at most once per frame
        key = (clip, fid)
        if key not in decoded:
            decoded[key] = pool.submit(decode, os.path.join(args.input_path, clip, frames[fid]), early.pop(key, None))
        return decoded[key]

    def finish(job):
        """(worker thread) ONE output image of a window: wait for its D2H copy, encode + write it, score it.  One task per
        image (round 5; before: the window's three images in one task) — in steady state the pool is busy either way, but
        the LAST window's three encodes now run side by side instead of one after the other (the drain at the end of a clip)."""
        clip, name, mine, kind, img, done, blurry_path = job
        done.synchronize()
        img = img.numpy()
        if mine:
            if name.lower().endswith(".png") and img.ndim == 3 and img.shape[2] == 3:
                # DEFLATE in four bands on their own small pool (round 5): the last window of a clip waits for one band, not
                # for a whole image (util.png_bytes_striped; same pixels for any PNG reader)
                util.save_png_striped(img, os.path.join(result_root, clip, name), pool=strip_pool, strips=4)
            else:
                util.save_img(img, os.path.join(result_root, clip, name))
            with written_lock:
                written.append(os.path.join(clip, name))
        if mine or kind == "interp":               # the reference scores a deblurred frame when it writes it
            _score(sums, clip, kind, img, args.gt_path and os.path.join(args.gt_path, clip, name), args.ssim)
        if blurry_path is not None and args.gt_path:
            _score(sums, clip, "blurry", data_util.imread_u8(blurry_path)[:, :, :3],
                   os.path.join(args.gt_path, clip, name), args.ssim)

    written, written_lock = [], threading.Lock()       # files THIS rank saved (--manifest)
    stamps = []                                        # host time at which each window's work had been queued
    group = []                                         # windows of one clip waiting to go through the net together

    def flush():
        """One forward for the windows in `group` (batched along N when there are several), then hand each window's
        three u8 images to the writer pool through a copy stream."""
        if not group:
            return
        t0 = time.time()
        (h, w), (l, r, t, b) = geom[:2], geom[2]
        if len(group) == 1:
            inputs = group[0][3]
            Ft_p = netG(*inputs, stage1_cache=stage1_cache) if reuse else netG(*inputs)
        else:
            Ft_p = netG(*[torch.cat([g[3][k] for g in group], 0) for k in range(6)])
        for j, (clip, names, owned, _, blurry_path) in enumerate(group):
            outs = torch.stack([ops.frame_to_u8(Ft_p[k][j:j + 1], t, l, h, w) for k in OUT_KEYS])
            ready = torch.cuda.Event()
            ready.record()
            host = pinned.pop() if pinned and pinned[-1].shape == outs.shape else torch.empty(
                outs.shape, dtype=torch.uint8, pin_memory=True)
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ready)
                host.copy_(outs, non_blocking=True)
                outs.record_stream(copy_stream)
                done = torch.cuda.Event()
                done.record()
            jobs = [pool.submit(finish, (clip, name, mine, kind, host[k], done, blurry_path if k == 0 else None))
                    for k, (name, mine, kind) in enumerate(zip(names, owned, ("interp", "deblur", "deblur"))) if name is not None]
            pending.append((jobs, host))
        while len(pending) > 8 + len(group):             # bound host memory; surfaces worker exceptions
            jobs, buf = pending.pop(0)
            for job in jobs:
                job.result()
            pinned.append(buf)
        timer.update((time.time() - t0) / len(group), len(group))
        stamps.extend([time.time()] * len(group))
        group.clear()

    t_all = time.time()
    with torch.no_grad():
        for wi in range(begin, end):
            clip, frames, index = windows[wi]
            if clip != cur_clip:
                flush()
                cur_clip = clip
                decoded.clear(); frames_dev.clear(); stage1_cache.clear()
                os.makedirs(os.path.join(result_root, clip), exist_ok=True)
            if wi == begin + 3:
                early.clear()            # the early decodes belong to the first three windows; whatever was not taken is dropped
            names = output_names(frames, index)
            clip_dir = os.path.join(result_root, clip)
            # who writes what follows from the GLOBAL window index alone, as in the serial reference loop: <num+12>
            # belongs to the window that reaches it first (its Ft_p[12]); the next window's Ft_p[8] has the same name
            # and is dropped — so a window writes its first deblurred frame only when it opens the clip.  Decided
            # without looking at the file system, two ranks on either side of a shard boundary can never both write
            # one file (the existence check below only skips work a previous run already finished).
            owned = [True, index == 0, names[2] is not None]
            owned = [mine and (clip, n) not in claimed and not os.path.exists(os.path.join(clip_dir, n))
                     for n, mine in zip(names, owned)]
            claimed.update((clip, n) for n, mine in zip(names, owned) if mine)
            if not any(owned) and not args.gt_path:
                continue
            ids = harness.window_frame_ids(index, len(frames))
            for ahead in range(1, 3 + args.batch):       # keep the decoders a few windows ahead of the GPU
                if wi + ahead < end and windows[wi + ahead][0] == clip:
                    for fid in harness.window_frame_ids(index + ahead, len(frames)):
                        want(clip, frames, fid)
            six = []
            for fid in ids:
                if fid not in frames_dev:
                    img = want(clip, frames, fid).result()
                    if img.shape[2] != 3:
                        raise RuntimeError(f"{clip}/{frames[fid]}: expected a 3-channel image")
                    geom = (img.shape[0], img.shape[1], util.pad_sizes(img.shape[0], img.shape[1]))
                    frames_dev[fid] = ops.u8_to_frame(upload(img), geom[2])
                six.append(frames_dev[fid])
            for fid in [f for f in frames_dev if f < min(ids)]:      # the windows in `group` hold their own references
                del frames_dev[fid]
                decoded.pop((clip, fid), None)
            group.append((clip, names, owned, six, os.path.join(args.input_path, clip, frames[ids[3]])))
            if len(group) >= args.batch:
                flush()
        flush()
        t_loop = time.time()
        torch.cuda.current_stream(dev).synchronize()
        t_gpu = time.time()
        for jobs, _ in pending:
            for job in jobs:
                job.result()
    torch.cuda.synchronize()
    ops.check_status(dev)                 # a frame that left the fp16 storage range is an error, not a result
    wall = time.time() - t_all
    timeline = {"first_window_queued_s": round(stamps[0] - t_all, 4) if stamps else None,
                "all_windows_queued_s": round(t_loop - t_all, 4), "gpu_done_s": round(t_gpu - t_all, 4), "files_done_s": round(wall, 4)}
    pool.shutdown()
    strip_pool.shutdown()
    if args.manifest:
        with open(os.path.join(result_root, f"written.rank{rank}.txt"), "w") as f:
            f.write("".join(sorted(w + "\n" for w in written)))

    # ---- combine the ranks' sums (one small all-reduce) and report like test.py:466-502
    vec = torch.tensor([v for k in METRICS for v in sums.total[k]] + [end - begin, wall], dtype=torch.float64)
    if world > 1:
        import torch.distributed as dist
        if dist.get_backend() != "gloo":      # nccl reduces device tensors; gloo takes the host vector as it is
            vec = vec.to(dev)
        wall_t = vec[-1:].clone()
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
        vec[-1] = wall_t[0]
        vec = vec.cpu()
    if rank == 0:
        tot = {k: (vec[2 * i].item() / max(vec[2 * i + 1].item(), 1)) for i, k in enumerate(METRICS)}
        n_win, wall = int(vec[-2].item()), vec[-1].item()
        if world == 1:
            for clip, d in sums.clips.items():
                log.info("clip %s: " % clip + " ".join(f"{k} {d[k][0] / max(d[k][1], 1):.4f}" for k in METRICS if d[k][1]))
        log.info("Avg. testset " + " ".join(f"{k} {tot[k]:.4f}" for k in METRICS))
        log.info("windows: %d  wall: %.2f s  -> %.2f interpolated frames/s (IO included); net+glue per window %.4f s",
                 n_win, wall, n_win / max(wall, 1e-9), timer.avg)
        if stats is not None:
            stats.update(windows=n_win, wall=wall, net_s_per_window=timer.avg, stamps=[t - t_all for t in stamps], timeline=timeline)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
