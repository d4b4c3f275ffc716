"""-m gpu: the N > 1 paths on one GPU — nccl world 1 forcing every collective, world 2 over gloo on the shared device, folder sharding, bench.py launched the way the driver launches it."""
import hashlib
import os
import socket
import sys
import threading

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _train_opt_r2(tmp_path, lr=1e-4, precision=None, dist=False):
    return {"model": "bin", "gpu_ids": [0], "is_train": True, "dist": dist,
            "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2, "precision": precision},
            "path": {"pretrain_model_G": None, "strict_load": True, "models": str(tmp_path), "training_state": str(tmp_path)},
            "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None,
                      "lr_G": lr, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000],
                      "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}


def _batch(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    return {"LQs": torch.rand(B, 6, 3, S, S, generator=g), "GTenh": torch.rand(B, 6, 3, S, S, generator=g),
            "GTinp": torch.rand(B, 5, 3, S, S, generator=g)}


# ------------------------------------------------------------------------------------------------ RCCL / threads / ABI
def test_nccl_world1_flat_allreduce_broadcast_and_barrier(tmp_path):
    """The RCCL path (backend "nccl") has otherwise only run under gloo: initialise a world-size-1 nccl group on the
    GPU, build the model with dist=True (bucketed parameter broadcast + flat gradient buffer), run one training step
    with the gradient all-reduce on its side stream, then barrier.  Results must equal the non-distributed step."""
    import torch.distributed as dist
    from bin_amd.models import create_model
    from bin_amd.weights import reference_state_dict
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29617")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        data = _batch(1, 64, 3)
        ref = create_model(_train_opt_r2(tmp_path, lr=1e-4))
        ref.netG.module.load_state_dict(reference_state_dict(0), strict=True)
        ref.feed_data(data)
        ref.optimize_parameters(1)
        m = create_model(_train_opt_r2(tmp_path, lr=1e-4, dist=True))
        m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
        m.grad_sync.force_collective = True                        # world size 1: still issue every collective
        m.broadcast_parameters(force=True)
        assert len(m.grad_sync._buckets) == 4                      # model1..model4 reduce during backward
        t = torch.ones(8, device="cuda")
        dist.all_reduce(t)                                         # RCCL really runs a collective on this GPU
        assert float(t.sum()) == 8.0
        m.feed_data(data)
        m.optimize_parameters(1)
        dist.barrier()
        torch.cuda.synchronize()
        assert float(m.loss) == float(ref.loss)
        for (k, a), (_, b) in zip(m.netG.module.named_parameters(), ref.netG.module.named_parameters()):
            assert torch.equal(a, b), k
    finally:
        dist.destroy_process_group()


REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _train_opt_r3(tmp_path, dist=False):
    return {"model": "bin", "gpu_ids": [0], "is_train": True, "dist": dist,
            "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2, "precision": "f16x3"},
            "path": {"pretrain_model_G": None, "strict_load": True, "models": str(tmp_path), "training_state": str(tmp_path)},
            "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None,
                      "lr_G": 1e-4, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000],
                      "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}


def _spawn(target, world, args, timeout=900):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(got, key=lambda g: g[0])


# ------------------------------------------------------------------------------------------------ world 2 on one GPU
def _w2_train_worker(rank, world, port, q, tmp):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BIN_AMD_DIST_BACKEND="gloo")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bin_amd.models import create_model
        from bin_amd.weights import reference_state_dict
        m = create_model(_train_opt_r3(os.path.join(tmp, str(rank)), dist=True))
        net = m.netG.module
        net.load_state_dict(reference_state_dict(0), strict=True)
        if rank == 1:                                   # rank 1 starts from different weights: the broadcast must fix it
            with torch.no_grad():
                for p in net.parameters():
                    p.mul_(1.01)
        m.broadcast_parameters()
        assert len(m.grad_sync._buckets) == 4           # model1..model4 are reduced DURING backward, on the side stream
        data = _batch(2, 64, 3)
        m.feed_data({k: v[rank:rank + 1] for k, v in data.items()})
        reduced_early = []
        orig = m.grad_sync._reduce_slice

        def spy(start, end, overlap):
            reduced_early.append((start, end, overlap, torch.cuda.current_stream().cuda_stream))
            return orig(start, end, overlap)
        m.grad_sync._reduce_slice = spy
        m.optimize_parameters(1)
        torch.cuda.synchronize()
        early = [r for r in reduced_early if r[2]]
        assert len(early) == 4, reduced_early           # four buckets went out from inside the backward pass ...
        assert m.grad_sync._stream is not None          # ... on the reducer's own side stream
        grads = m.grad_sync.flat.detach().cpu().numpy().copy()
        loss1 = float(m.loss.detach())
        m.optimize_parameters(2)
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy()
        q.put((rank, loss1, grads if rank == 0 else None, hashlib.sha256(flat.tobytes()).hexdigest(), dist.get_world_size()))
    finally:
        dist.destroy_process_group()


def test_world2_on_one_gpu_real_backward_with_overlapped_bucket_reduce(tmp_path):
    got = _spawn(_w2_train_worker, 2, (str(tmp_path),))
    assert [g[4] for g in got] == [2, 2]
    assert got[0][1] != got[1][1]                        # different samples -> different local losses
    assert got[0][3] == got[1][3]                        # bit-identical parameters on both ranks after two steps
    # single process, batch 2 (same two samples): its gradients are the mean over the samples = the ranks' average
    from bin_amd.models import create_model
    from bin_amd.weights import reference_state_dict
    m = create_model(_train_opt_r3(tmp_path / "single"))
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    m.feed_data(_batch(2, 64, 3))
    m.optimize_parameters(1)
    ref = torch.cat([p.grad.reshape(-1) for p in m.netG.module.parameters()]).cpu().numpy()
    avg = got[0][2]
    assert avg.shape == ref.shape
    o = 0
    worst = 0.0
    for name, p in m.netG.module.named_parameters():
        a, b = avg[o:o + p.numel()], ref[o:o + p.numel()]
        o += p.numel()
        scale = float(np.abs(b).max())
        if scale > 0:
            worst = max(worst, float(np.abs(a - b).max()) / scale)
            assert float(np.abs(a - b).max()) <= 2e-4 * scale + 1e-9, name
    assert abs(0.5 * (got[0][1] + got[1][1]) - float(m.loss)) <= 2e-6
    print(f"world-2 averaged gradients vs single-process batch 2: worst relative difference {worst:.2e}")


def _w2_folder_worker(rank, world, port, q, argv):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from bin_amd import test as run_test
    rc = run_test.main(list(argv) + ["--launcher", "pytorch", "--backend", "gloo", "--manifest"])
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()
    q.put((rank, rc))


def test_world2_folder_sharding_is_complete_and_duplicate_free(tmp_path):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_gpu_harness import _blur_tree, _yml
    from bin_amd import test as run_test
    from bin_amd.data import util as du
    from bin_amd.weights import reference_state_dict
    clips = (("c0", 0, 6), ("c1", 40, 4))               # 5 + 3 windows: the shard boundary falls INSIDE clip c0
    root = _blur_tree(str(tmp_path / "data"), clips=clips)
    weights = str(tmp_path / "w.pth")
    torch.save(reference_state_dict(0), weights)
    yml = _yml(tmp_path, weights)
    common = ["--input_path", os.path.join(root, "test_blur"), "--gt_path", os.path.join(root, "test"),
              "--opt", yml, "--precision", "f16x3", "--io_threads", "2"]
    out2 = str(tmp_path / "out2")
    got = _spawn(_w2_folder_worker, 2, (common + ["--output_path", out2],))
    assert [g[1] for g in got] == [0, 0]
    out1 = str(tmp_path / "out1")
    assert run_test.main(common + ["--output_path", out1, "--manifest"]) == 0
    res1 = os.path.join(out1, "60fps_test_results", "adobe_stage4")
    res2 = os.path.join(out2, "60fps_test_results", "adobe_stage4")
    man = [open(os.path.join(res2, f"written.rank{r}.txt")).read().split() for r in range(2)]
    single = open(os.path.join(res1, "written.rank0.txt")).read().split()
    assert man[0] and man[1]                                         # both ranks wrote something
    assert not set(man[0]) & set(man[1])                             # no file written twice
    assert sorted(man[0] + man[1]) == sorted(single)                 # together: exactly the single-process file set
    for rel in single:                                               # and the same images, bit for bit
        assert np.array_equal(du.imread_u8(os.path.join(res2, rel)), du.imread_u8(os.path.join(res1, rel))), rel
    log = open(os.path.join(res2, [f for f in os.listdir(res2) if f.endswith(".log")][0])).read()
    assert "ranks: 2" in log and "windows: 8" in log


def _bench_as_the_driver_launches_it(n, extra, port):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` (the driver's N > 1 command),
    with bench.py's test hook that puts every rank on the one GPU of the box over gloo (RCCL refuses two ranks per device)."""
    import json
    import subprocess
    env = dict(os.environ, BIN_AMD_BENCH_BACKEND="gloo", BIN_AMD_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", str(n)] + extra
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]                             # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_two_ranks_inference_line_is_whole_job_throughput():
    d = _bench_as_the_driver_launches_it(2, ["--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"], 29631)
    assert d["n_gpus"] == 2 and d["nccl_ranks"] == 2 and d["backend"] == "gloo" and d["scaling"] == "weak"
    assert d["steps"] == 2 and d["warmup"] == 1
    assert abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2 * 1e-3)) < 1e-2 * d["value"]       # all ranks' windows / max time
    assert d["roofline"]["frac"] <= 1.0 and d["power"]["samples"] >= 0


def test_bench_two_ranks_training_line_reduces_gradients_across_ranks():
    d = _bench_as_the_driver_launches_it(2, ["--mode", "train", "--batch", "1", "--steps", "2", "--warmup", "1"], 29633)
    assert d["n_gpus"] == 2 and d["nccl_ranks"] == 2 and d["backend"] == "gloo"
    assert abs(d["value"] - 2 * 1 * 2 / (d["ms_per_step"] * 2 * 1e-3)) < 1e-2 * d["value"]
    assert np.isfinite(d["loss"]) and d["roofline"]["dominant_kernel"]["launches"] > 0
