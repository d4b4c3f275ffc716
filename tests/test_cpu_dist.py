"""CPU, world_size 2, gloo: the N>1 paths — flat gradient all-reduce of the training wrapper (DP) and
window sharding of inference (no data-path collective, a gather of PSNR sums only)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_cpu_host import _cpu_model
        from conftest import load_golden
        g = load_golden("g3_train")
        # global batch of 2 = the golden sample twice -> rank r gets one copy; averaged grads must equal
        # the single-sample golden gradients, and both ranks must hold identical parameters afterwards
        m = _cpu_model(tmp, dist=True)
        m.feed_data({"LQs": torch.from_numpy(g["LQs"]), "GTenh": torch.from_numpy(g["GTenh"]),
                     "GTinp": torch.from_numpy(g["GTinp"])})
        if rank == 1:                       # perturb rank 1's data: grads differ before the all-reduce
            m.B1 = m.B1 * 0.5
        m.optimize_parameters(1)
        named = dict(m.netG.module.named_parameters())
        key = "model.model1_1.SFENet1.weight"
        q.put((rank, float(m.loss), named[key].grad.double().sum().item(), named[key].detach().double().sum().item()))
        from bin_amd.harness import shard_windows
        q.put((rank, "shard", shard_windows(1287, rank, world)))
    finally:
        dist.destroy_process_group()


def test_dp_grad_allreduce_and_sharding(tmp_path):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    train = sorted([g for g in got if g[1] != "shard"])
    shards = sorted([g for g in got if g[1] == "shard"])
    assert train[0][1] != train[1][1]                      # different local losses (different data)
    assert abs(train[0][2] - train[1][2]) < 1e-9           # identical averaged gradients
    assert abs(train[0][3] - train[1][3]) < 1e-12          # identical parameters after the step
    (a0, a1), (b0, b1) = shards[0][2], shards[1][2]
    assert a0 == 0 and a1 == b0 and b1 == 1287 and abs((a1 - a0) - (b1 - b0)) <= 1


def test_shard_windows_covers_everything():
    from bin_amd.harness import shard_windows
    for n in (0, 1, 7, 8, 1287):
        for world in (1, 2, 3, 8):
            spans = [shard_windows(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_flat_grad_views_accumulate_in_place():
    """FlatGradAllReduce.attach(): every .grad is a slice of one flat buffer and autograd accumulates into it in place
    (so the DP all-reduce needs no per-parameter copies)."""
    from bin_amd.models.bin_model import FlatGradAllReduce
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 1))
    sync = FlatGradAllReduce(net.parameters())
    sync.attach()
    ptrs = [p.grad.data_ptr() for p in net.parameters()]
    x = torch.rand(2, 3, 8, 8)
    net(x).sum().backward()
    net(x * 2).sum().backward()                         # second backward accumulates
    assert [p.grad.data_ptr() for p in net.parameters()] == ptrs
    ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert torch.equal(ref, sync.flat) and float(sync.flat.abs().sum()) > 0
    sync.attach()                                       # next step: zeroed, same storage
    assert float(sync.flat.abs().sum()) == 0 and [p.grad.data_ptr() for p in net.parameters()] == ptrs


def test_interval_subtraction():
    from bin_amd.models.bin_model import _subtract
    assert _subtract([(0, 10)], []) == [(0, 10)]
    assert _subtract([(0, 10)], [(3, 5)]) == [(0, 3), (5, 10)]
    assert _subtract([(0, 10), (20, 30)], [(8, 22), (25, 26)]) == [(0, 8), (22, 25), (26, 30)]
    assert _subtract([(0, 10)], [(0, 10)]) == []
    assert _subtract([(5, 7)], [(0, 100)]) == []


def _stray_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bin_amd.models.bin_model import FlatGradAllReduce
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2), torch.nn.Linear(2, 1))
        params = list(net.parameters())
        sync = FlatGradAllReduce(params)
        sync.attach()
        local = [torch.full_like(p, float(rank + 1) * (i + 1)) for i, p in enumerate(params)]
        for p, g in zip(params, local):
            p.grad.copy_(g)
        # an "early bucket": the first two layers' slice is reduced before the rest of the backward ...
        end = sum(p.numel() for p in params[:4])
        sync._reduce_slice(0, end, overlap=False)
        # ... and afterwards somebody REPLACES one of its gradients by a fresh (local, unreduced) tensor
        params[1].grad = local[1].clone() * 10
        sync()
        want = [(g * 10 if i == 1 else g) for i, g in enumerate(local)]
        # mean over ranks of rank-local values: local_r = (r + 1) * c  ->  mean = 1.5 * c for world 2
        ok = all(torch.allclose(p.grad, w / float(rank + 1) * 1.5) for p, w in zip(params, want))
        views = sync._views_intact()
        q.put((rank, ok, views))
    finally:
        dist.destroy_process_group()


def test_replaced_grad_after_an_early_bucket_reduce_is_averaged_once():
    """a .grad replaced after its bucket was already all-reduced must be reduced exactly once more (that
    parameter only), the rest of the bucket must NOT be summed a second time."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stray_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, True, True), (1, True, True)], got
