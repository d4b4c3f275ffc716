"""Collectives of the one-process-per-GPU layer, independent of where the tensor lives.

On the GPU box the default process group is "nccl" (= RCCL over xGMI) and device tensors go to it as they are.  RCCL
refuses two ranks on one device, so world > 1 on ONE GPU (tests, debugging on a single-GPU box) runs over "gloo"; a
device tensor is then staged through page-locked host memory on the CALLER'S CURRENT STREAM — copy out, wait for the
stream (the data a collective reduces must be complete: that wait is what a nccl kernel's stream order provides), reduce
on the host, copy back on the same stream — so the stream semantics the callers rely on (side-stream bucket reduces in
`FlatGradAllReduce`) are the same under both backends.
"""
import os

import torch
import torch.distributed as dist


def default_backend():
    """BIN_AMD_DIST_BACKEND, else nccl when a HIP device is present, else gloo."""
    return os.environ.get("BIN_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")


def _staged(t):
    return t.is_cuda and dist.get_backend() == "gloo"


def _via_host(t, fn):
    stream = torch.cuda.current_stream(t.device)
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    stream.synchronize()
    fn(host)
    t.copy_(host, non_blocking=True)
    stream.synchronize()                    # `host` dies with this frame: the copy back must have read it
    return t


def all_reduce(t, op=None):
    op = dist.ReduceOp.SUM if op is None else op
    if _staged(t):
        return _via_host(t, lambda h: dist.all_reduce(h, op=op))
    dist.all_reduce(t, op=op)
    return t


def broadcast(t, src=0):
    if _staged(t):
        return _via_host(t, lambda h: dist.broadcast(h, src=src))
    dist.broadcast(t, src=src)
    return t
