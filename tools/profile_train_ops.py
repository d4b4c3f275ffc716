#!/usr/bin/env python3
"""Where do the small ATen launches of a training step come from?  (rocprofv3 shows ~400 fills, ~340 buffer copies and ~530
relayout launches per step, 4-5 us each.)  One optimize_parameters() under torch.profiler, aggregated by op and input shape,
plus the Python stacks of the most frequent ones."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

from bin_amd.models import create_model  # noqa: E402
from bin_amd.weights import reference_state_dict  # noqa: E402

tmp = tempfile.mkdtemp()
opt = {"model": "bin", "gpu_ids": [0], "is_train": True, "dist": False,
       "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2, "precision": "f16x3", "backward_precision": None},
       "path": {"pretrain_model_G": None, "strict_load": True, "models": tmp, "training_state": tmp},
       "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None, "lr_G": 1e-4,
                 "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000], "restarts": None,
                 "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}
m = create_model(opt)
m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
g = torch.Generator().manual_seed(7)
B, S = 8, 256
m.feed_data({"LQs": torch.rand(B, 6, 3, S, S, generator=g), "GTenh": torch.rand(B, 6, 3, S, S, generator=g),
             "GTinp": torch.rand(B, 5, 3, S, S, generator=g)})
for i in range(2):
    m.optimize_parameters(i + 1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    m.optimize_parameters(3)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = sorted(ka, key=lambda e: -e.count)
print("count  device_us_total  op  shapes")
for e in rows[:45]:
    dt = getattr(e, "device_time_total", None)
    if dt is None:
        dt = getattr(e, "cuda_time_total", 0)
    print(f"{e.count:6d} {dt:12.1f}  {e.key[:60]:60s} {str(e.input_shapes)[:90]}")
print("---- fills / zeros / copies / adds by input shape")
for e in rows:
    if any(k in e.key for k in ("fill_", "zero_", "copy_", "aten::add", "aten::zeros", "aten::cat", "aten::clone")):
        print(f"{e.count:6d}  {e.key[:40]:40s} {str(e.input_shapes)[:110]}")
print("---- stacks of the most frequent small ops")
ks = prof.key_averages(group_by_stack_n=6)
for e in sorted(ks, key=lambda e: -e.count)[:80]:
    if any(k in e.key for k in ("fill_", "zero_", "copy_", "aten::add", "aten::mul", "aten::cat", "aten::slice", "aten::neg", "aten::empty")):
        print(e.count, e.key, [s.split("/")[-1] for s in e.stack if "bin_amd" in s or "torch/optim" in s or "autograd" in s][:5])
