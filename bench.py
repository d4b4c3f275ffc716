#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: interpolated frames/s at 1280x720 on N MI355X.

A "step" = one 6-frame forward of bin_stage4 (`netG(B1..B11)`, what the reference's test.py runs per
input frame, test.py:249-379) on synthetic 6 x [1,3,720,1280] U[0,1) frames replicate-padded to
[1,3,768,1344] by the test.py rule (:348-366), seeded random-init weights (bin_amd/weights.py), inputs
resident in HBM before the timed region.  One step = one interpolated frame (Ft_p[13]) + two restored
frames.  N>1: every rank runs its own windows (window-sharded inference, no data-path collective;
weak scaling), value = all ranks' windows / max-over-ranks time.

Extra objects on the JSON line:
  roofline     — dominant kernel (RDB 3x3 conv Cin->32, 70 % of FLOPs): algorithmic fp32 bytes per launch
                 / its mean duration measured with HIP events on the launch stream INSIDE the timed region
  cpu_baseline — the oracle (CPU restatement, kind "port") timed on the host cores on a bounded sample
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402

H, W = 720, 1280
HBM_PEAK_GBS = 8000.0
# SURVEY.md §8(d) / BASELINE.md §4 constants at 768x1344 padded
FLOP_20, FLOP_17 = 29.386e12, 25.03e12
BYTES_20, BYTES_17 = 304.5e9, 259.0e9


def rdb_conv_algorithmic_bytes(n, h2, w2):
    """fp32 layer-wise minimum of ONE launch of the dominant kernel, averaged over the RDB convs it runs
    (Cin = 96,128,160 -> 32; conv #3, Cin = 192, lives in the fused conv+LFF kernel): read Cin planes once,
    write 32 once, weights once (SURVEY.md §8d per-layer figure)."""
    px = n * h2 * w2
    per = [(cin + 32) * 4 * px + (cin * 32 * 9 + 32) * 4 for cin in (96, 128, 160)]
    return sum(per) / len(per)


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_pmc_traffic.md: separate FETCH_SIZE / WRITE_SIZE runs, FETCH doubled per the gfx950 calibration).
    PMC counters cannot be read from inside the process, so bench.py reports the profiled figure (or null)."""
    try:
        with open(os.path.join(REPO, "profiles", "r01_pmc_traffic.json")) as f:
            return int(json.load(f)["traffic_bytes_per_launch"])
    except Exception:
        return None


def _host_cores():
    """Cores this process may really use: affinity mask, clipped by the cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline_worker(budget_s=20.0):
    """(runs in a subprocess) Oracle forward on the host cores on a BOUNDED sample: 6-frame forwards of
    the reference's literal schedule on growing crops (48x96 -> 96x168 -> 192x336 = 1/16 of the padded
    768x1344 area), stopping before the time budget is exceeded; the largest crop timed is scaled to
    720p windows/s by the pixel ratio (conv FLOPs are linear in pixels)."""
    from oracle import rdn_oracle as O
    from bin_amd.weights import canonical_weights, synthetic_frames
    cores = min(_host_cores(), 64)
    torch.set_num_threads(cores)
    canon = {k: torch.from_numpy(v) for k, v in canonical_weights(0).items()}
    t_begin = time.time()
    best = None
    with torch.no_grad():
        O.bin_stage4_forward(synthetic_frames(1, 1, 32, 32, 6), canon)          # warm-up (thread pool, oneDNN)
        prev = None
        for hs, ws in ((48, 96), (96, 168), (192, 336)):
            if prev is not None:
                est = prev[2] * (hs * ws) / (prev[0] * prev[1])
                if (time.time() - t_begin) + 2 * est > budget_s:
                    break
            frames = synthetic_frames(1234, 1, hs, ws, 6)
            times = []
            for _ in range(2):
                t0 = time.time()
                O.bin_stage4_forward(frames, canon)
                times.append(time.time() - t0)
            prev = (hs, ws, min(times))
            best = prev
    hs, ws, t = best
    scale = (hs * ws) / (768.0 * 1344.0)
    return {"value": round(scale / t, 6), "unit": "interpolated frames/s at 1280x720", "cores": cores,
            "kind": "port",
            "sample": f"oracle (PyTorch-CPU restatement of the reference, 20-call schedule) 6-frame forward on a "
                      f"{hs}x{ws} crop = {t:.2f} s (best of 2), scaled by pixel ratio {scale:.5f} to 768x1344; the linear "
                      f"extrapolation flatters the CPU — one full-size 768x1344 oracle forward measured on the same kind of "
                      f"host (16 cores, tests/full_size_parity.py) takes 112.8 s = 0.0089 frames/s"}


def cpu_baseline(timeout_s=150):
    """Run the CPU baseline in a child process with a hard timeout so it can never stall the bench."""
    import subprocess
    code = ("import json,sys; sys.path.insert(0, %r); import bench; "
            "print('CPU_BASELINE ' + json.dumps(bench.cpu_baseline_worker()))" % REPO)
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s, env=env)
        for ln in out.stdout.splitlines():
            if ln.startswith("CPU_BASELINE "):
                return json.loads(ln[len("CPU_BASELINE "):])
        return {"value": None, "unit": "interpolated frames/s at 1280x720", "cores": _host_cores(), "kind": "port",
                "sample": "cpu baseline failed: " + (out.stderr.strip().splitlines() or ["?"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "interpolated frames/s at 1280x720", "cores": _host_cores(), "kind": "port",
                "sample": f"cpu baseline exceeded {timeout_s}s and was cut"}


def train_bench(args, rank, world, dev):
    """Secondary metric (BASELINE.json config 4): training samples/s, one process per GPU, DP gradient all-reduce."""
    import tempfile
    import torch.distributed as dist
    from bin_amd.models import create_model
    from bin_amd.weights import reference_state_dict
    prec = args.train_precision
    bwd_prec = None
    if prec == "mixed":                      # fp32-class forward (exact loss / ReLU masks), single-product backward
        prec, bwd_prec = "f16x3", "f16"
    if os.environ.get("BIN_AMD_WGRAD_DBG"):          # kernel A/B switch for tools/ experiments (binhip_wgrad_set_debug)
        from bin_amd import _lib
        _lib.lib().binhip_wgrad_set_debug(int(os.environ["BIN_AMD_WGRAD_DBG"]))
    tmp = tempfile.mkdtemp()
    opt = {"model": "bin", "gpu_ids": [0], "is_train": True, "dist": world > 1,
           "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2, "precision": prec,
                         "backward_precision": bwd_prec},
           "path": {"pretrain_model_G": None, "strict_load": True, "models": tmp, "training_state": tmp},
           "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None,
                     "lr_G": 1e-4, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000],
                     "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}
    m = create_model(opt)
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    g = torch.Generator().manual_seed(7 + rank)
    B, S = args.batch, 256
    batch = {"LQs": torch.rand(B, 6, 3, S, S, generator=g), "GTenh": torch.rand(B, 6, 3, S, S, generator=g),
             "GTinp": torch.rand(B, 5, 3, S, S, generator=g)}
    m.feed_data(batch)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        m.optimize_parameters(i + 1)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        m.optimize_parameters(args.warmup + i + 1)
    sync_all()
    dt = time.perf_counter() - t0
    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())
    if rank == 0:
        print(json.dumps({
            "metric": "training samples/sec (256x256 crops, 6-frame windows)", "value": round(world * B * args.steps / dt, 4),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.train_precision, "data": "synthetic", "loss": float(m.loss.detach()),
            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
            "config": {"workload": f"Adobe240 training, 256x256 crops, batch {B} per GPU, Charbonnier x17, Adam, "
                                   f"DP flat gradient all-reduce (45.77 MB)", "precision": prec, "backward_precision": bwd_prec or prec}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default=os.environ.get("BIN_AMD_BENCH_PRECISION", "f16"),
                    choices=["f16", "f16x3"])
    ap.add_argument("--streams", type=int, default=None, help="concurrent RDN calls (HIP streams) in the forward")
    ap.add_argument("--batched", action="store_true",
                    help="batch the shared-weight RDN calls of each pyramid stage (N>1 launches) instead of multi-stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--calib", action="store_true",
                    help="also run one 256 MiB device copy (known HBM bytes) to calibrate rocprofv3 FETCH/WRITE_SIZE")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="train: BASELINE config 4 — one optimize_parameters() (fwd + Charbonnier + bwd + grad "
                         "all-reduce + Adam) on 256x256 crops, --batch samples per GPU (secondary metric, own JSON line)")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--train-precision", default="f16x3", choices=["f16", "f16x3", "mixed"],
                    help="precision of --mode train (default f16x3: fp32-class gradients)")
    ap.add_argument("--pipeline", action="store_true",
                    help="let consecutive steps overlap (side streams wait on the resident frames, not on the previous join)")
    ap.add_argument("--reference-schedule", action="store_true",
                    help="run the reference's literal 20 RDN calls + 12 cells instead of the exact 17 + 6")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand with --gpus N: re-launch as one process per GPU (what the driver does itself for N > 1)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29531"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.mode == "train":
        return train_bench(args, rank, world, dev)

    from bin_amd import _lib as L
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.utils import util
    from bin_amd.weights import reference_state_dict, synthetic_frames

    net = bin_stage4_lstm()
    net.load_state_dict(reference_state_dict(0), strict=True)
    net = net.to(dev).eval().set_precision(args.precision)
    net.reuse_schedule = not args.reference_schedule
    if args.streams is not None:
        net.n_streams = args.streams
    if args.batched:
        net.batched = True

    pads = util.pad_sizes(H, W)
    frames = [util.replicate_pad(f, pads).to(dev) for f in synthetic_frames(1234 + rank, 1, H, W, 6)]
    hp, wp = frames[0].shape[2], frames[0].shape[3]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            out = net(*frames)
        if args.calib:
            src = torch.empty(64 << 20, dtype=torch.float32, device=dev).normal_()
            dst = torch.empty_like(src)
            for _ in range(3):
                dst.copy_(src)
            del src, dst
        sync_all()
        lib = L.lib()
        # --pipeline: the frames are resident and complete (sync_all above), so input_events=[] lets consecutive steps
        # overlap — the next window's stage-1 calls start on idle streams while the previous window's lone stage-4 call
        # still runs.  Measured +0.4 % (31.43 vs 31.30 frames/s): every kernel already fills both workgroup slots of
        # every CU, so another stream's kernels only slip into the ramp/drain.  Off by default.
        kw_in = {"input_events": []} if (net.n_streams > 1 and not net.batched and args.pipeline) else {}
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = net(*frames, **kw_in)
        sync_all()
        dt = time.perf_counter() - t0
        # ---- roofline leg: the dominant kernel's mean duration, HIP events on the launch stream.  With several
        # streams kernels of different RDN calls overlap and a per-kernel duration is not meaningful, so this pass
        # re-runs the same forward serially (n_streams = 1) right after the timed region; `value` is unaffected.
        launches_per_step = (17 if net.reuse_schedule else 20) * 36
        prof_steps = min(args.steps, 5)
        prof = rank == 0 and prof_steps * launches_per_step <= 16384
        kern_ms, kern_n = ctypes.c_double(0), ctypes.c_int(0)
        if prof:
            saved_streams = net.n_streams
            net.n_streams = 1
            out = net(*frames)
            torch.cuda.synchronize()
            L.check(lib.binhip_profile_begin(3, 32, L.EPI_PLANES, prof_steps * launches_per_step), "profile_begin")
            for _ in range(prof_steps):
                out = net(*frames)
            torch.cuda.synchronize()
            L.check(lib.binhip_profile_end(ctypes.byref(kern_ms), ctypes.byref(kern_n)), "profile_end")
            net.n_streams = saved_streams
        # ---- streaming leg (reported beside `value`, never instead of it): consecutive windows of
        # one clip, sliding by one frame, with the exact stage-1 reuse -> 13 instead of 17 RDN calls per window
        stream_fps = None
        if rank == 0 and net.n_streams > 1 and net.reuse_schedule:
            clip_frames = frames + [f.clone() for f in frames[:4]]      # 10 resident padded frames -> 5 windows
            cache = {}
            net(*clip_frames[0:6], stage1_cache=cache)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            nwin = 0
            for rep in range(2):
                for i in range(1, 5):
                    net(*clip_frames[i:i + 6], stage1_cache=cache)
                    nwin += 1
                for i in range(3, -1, -1):
                    net(*clip_frames[i:i + 6], stage1_cache=cache)      # sliding back also shares 4 of 5 pairs
                    nwin += 1
            torch.cuda.synchronize()
            stream_fps = nwin / (time.perf_counter() - ts)
        # ---- the same workload in the fp32-class precision mode, reported beside `value` (never instead of it)
        alt = None
        if rank == 0 and args.precision == "f16":
            net.set_precision("f16x3")
            for _ in range(2):
                net(*frames)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            n_alt = max(3, args.steps // 2)
            for _ in range(n_alt):
                net(*frames)
            torch.cuda.synchronize()
            alt = {"precision": "f16x3 (fp16 hi/lo split, 3 MFMA products; max-abs error ~1e-6 vs the fp32 reference)",
                   "value": round(n_alt / (time.perf_counter() - ta), 4), "unit": "interpolated frames/s", "n_gpus": 1}
            net.set_precision(args.precision)
        if world > 1:
            dist.barrier()
    assert all(torch.isfinite(o).all() for o in out)

    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())

    if rank == 0:
        value = world * args.steps / dt
        flops = FLOP_17 if net.reuse_schedule else FLOP_20
        abytes = BYTES_17 if net.reuse_schedule else BYTES_20
        ms = dt / args.steps * 1e3
        roof = None
        if prof and kern_n.value > 0:
            avg_s = kern_ms.value / kern_n.value * 1e-3
            ab = rdb_conv_algorithmic_bytes(1, hp // 2, wp // 2)
            ach = ab / avg_s / 1e9
            roof = {"bound": "hbm", "kernel": "conv_mfma_kernel<3,1,1,2,8,1,NT,2,0> (RDB conv3x3 Cin->32 +ReLU, convs 0-2 of each dense block)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": pmc_traffic(), "avg_kernel_us": round(avg_s * 1e6, 2), "launches": kern_n.value,
                    "algorithmic_bytes_per_launch": int(ab),
                    "whole_forward": {"algorithmic_GB": abytes / 1e9, "achieved_GBs": round(abytes / (ms * 1e-3) / 1e9, 1),
                                      "frac_hbm": round(abytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                      "achieved_TFLOPs": round(flops / (ms * 1e-3) / 1e12, 1)}}
        line = {
            "metric": "interpolated frames/sec at 1280x720", "value": round(value, 4),
            "unit": "interpolated frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 MFMA inputs, f32 accumulate" if args.precision == "f16"
                     else "f16 hi/lo split (3 MFMA products), f32 accumulate",
            "data": "synthetic",
            "config": {"workload": "Adobe240 test_blur 1280x720 inference, batch=1 per GPU "
                                   "(6-frame window padded to 768x1344 by the test.py rule), window-sharded",
                       "schedule": "17 RDN calls + 6 ConvLSTM cells (exact reuse)" if net.reuse_schedule
                                   else "20 RDN calls + 12 ConvLSTM cells (reference literal)",
                       "precision": args.precision, "streams": net.n_streams, "pipelined_steps": bool(kw_in), "batched_stages": bool(net.batched and net.n_streams > 1), "parity": "max-abs <= 1e-3 vs fp32 reference (tests/)"},
            "roofline": roof,
            "fp32_class": alt,
            "streaming": None if stream_fps is None else {
                "value": round(stream_fps, 3), "unit": "interpolated frames/s",
                "note": "consecutive windows of one clip (sliding by one frame) with exact stage-1 reuse: 13 RDN calls per "
                        "window instead of 17; same outputs bit for bit (tests/test_gpu_net.py); not the headline value"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
