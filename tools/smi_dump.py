"""Dump what amdsmi exposes on this box (raw gpu_metrics table, violation status, clock domains), idle and under a sustained
dense-block conv load — the field inventory bin_amd/utils/smi.py is written against.  usage: smi_dump.py [seconds]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bin_amd import ops
from bin_amd.utils import smi

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
s = smi.Sampler(0, 0.02)
print("source:", None if s.src is None else s.src.name, s.error)
src = s.src


def raw(tag):
    if src is None or src.name != "amdsmi":
        return
    m, h = src.m, src.h
    try:
        g = m.amdsmi_get_gpu_metrics_info(h)
        keep = {k: v for k, v in g.items() if not k.startswith(("pcie", "xgmi", "vcn", "jpeg", "common_header")) and "vclk" not in k and "dclk" not in k}
        print(f"[{tag}] gpu_metrics:", json.dumps(keep, default=str)[:6000])
    except Exception as e:
        print(f"[{tag}] gpu_metrics failed: {e}")
    t0 = time.perf_counter()
    try:
        v = m.amdsmi_get_violation_status(h)
        print(f"[{tag}] violation_status ({(time.perf_counter() - t0) * 1e3:.0f} ms):", json.dumps(v, default=str)[:3000])
    except Exception as e:
        print(f"[{tag}] violation_status failed: {e}")
    for name in ("GFX", "MEM", "DF", "SOC", "SYS"):
        try:
            print(f"[{tag}] clock_info {name}:", m.amdsmi_get_clock_info(h, getattr(m.AmdSmiClkType, name)))
        except Exception as e:
            print(f"[{tag}] clock_info {name} failed: {e}")
    try:
        print(f"[{tag}] power_info:", m.amdsmi_get_power_info(h), "cap:", m.amdsmi_get_power_cap_info(h), "energy:", m.amdsmi_get_energy_count(h))
    except Exception as e:
        print(f"[{tag}] power failed: {e}")
    t0 = time.perf_counter()
    for _ in range(20):
        src.read()
    print(f"[{tag}] one Sampler.read() = {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms")


raw("idle")
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
x = ops.nchw_to_planes((torch.rand(1, 224, 384, 672, generator=g) - 0.3).to(dev), 3)
cw = ops.ConvWeights(((torch.rand(32, 160, 3, 3, generator=g) - 0.5) / 38).to(dev), torch.zeros(32).to(dev), nterms=3)
out = ops.CP.empty(2, 1, 384, 672, 3, dev)
f = lambda: ops.conv2d(x, cw, relu=True, out=out, cin_chunks=10)
for _ in range(50):
    f()
torch.cuda.synchronize()
s.start()
t0 = time.time()
dumped = False
while time.time() - t0 < secs:
    for _ in range(200):
        f()
    if not dumped and time.time() - t0 > secs / 2:
        raw("load")
        dumped = True
    torch.cuda.synchronize()
print("summary under load:", json.dumps(s.stop(), indent=1))
