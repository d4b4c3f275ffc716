"""Where do single-launch stalls of the training step come from?  (one `rdb_tail_x3_kernel` launch of 17-22 ms in
the rocprofv3 traces of rounds 4 and 5.)

  python tools/stall_hunt.py steps [--steps 50] [--mode train|infer]     un-profiled: one HIP event per step boundary on the
                                                                         launch stream, per-step ms -> min / median / p99 / max
  python tools/stall_hunt.py trace DIR                                   DIR = output of
        rocprofv3 --kernel-trace --hip-trace --memory-allocation-trace --output-format csv -d DIR -- python bench.py --mode train ...
     every launch longer than 4x the median of its kernel, with what the host was inside of (HIP API calls overlapping the
     launch's interval) and which other kernels were in flight."""
import csv
import glob
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_step(mode, steps, warmup=3):
    import torch
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if mode == "train":
        step = bench.make_train_step(batch=8, precision="f16x3")
    else:
        step = bench.make_infer_step(precision="f16x3")
    for i in range(warmup):
        step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    host = []
    ev[0].record()
    t0 = time.perf_counter()
    for i in range(steps):
        step()
        ev[i + 1].record()
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    return {"mode": mode, "steps": steps, "wall_ms_per_step": round(wall, 3), **bench.step_spread(ms),
            "per_step_ms": [round(v, 3) for v in ms]}


def trace(d):
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    ht = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)
    mt = glob.glob(d + "/**/*memory_allocation_trace.csv", recursive=True)
    K = []
    with open(kt[0]) as f:
        for i, r in enumerate(csv.DictReader(f)):
            K.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], i, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    t_first = min(k[0] for k in K)
    by = {}
    for k in K:
        by.setdefault(k[2], []).append(k)
    Hs = []
    if ht:
        with open(ht[0]) as f:
            for r in csv.DictReader(f):
                Hs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "?")))
    Ms = []
    if mt:
        with open(mt[0]) as f:
            for r in csv.DictReader(f):
                Ms.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Operation", "?"), r.get("Allocation_Size", "?")))
    out = []
    for name, ls in by.items():
        if len(ls) < 8:
            continue
        med = statistics.median(e - s for s, e, *_ in ls)
        order = sorted(ls)
        for j, (s, e, _, idx, q, st) in enumerate(order):
            if e - s > 4 * med and e - s > 300e3:
                apis = sorted(((min(e, he) - max(s, hs)), fn, th, (he - hs)) for hs, he, fn, th in Hs if hs < e and he > s)[::-1][:6]
                mem = [(op, sz, (me - ms_) / 1e3) for ms_, me, op, sz in Ms if ms_ < e and me > s][:6]
                others = sorted(((min(e, oe) - max(s, os_)), on) for os_, oe, on, oi, *_ in K if oi != idx and os_ < e and oe > s)[::-1][:5]
                out.append({"kernel": name, "launch_of_this_kernel": j + 1, "of": len(ls), "dispatch": idx + 1, "queue": q, "stream": st,
                            "us": round((e - s) / 1e3, 1), "median_us": round(med / 1e3, 1), "ms_after_first_dispatch": round((s - t_first) / 1e6, 1),
                            "host_api_overlap": [{"fn": fn, "thread": th, "overlap_us": round(o / 1e3, 1), "call_us": round(c / 1e3, 1)} for o, fn, th, c in apis],
                            "memory_ops": mem,
                            "kernels_in_flight": [{"kernel": on, "overlap_us": round(o / 1e3, 1)} for o, on in others]})
    out.sort(key=lambda r: -r["us"])
    return {"dir": d, "dispatches": len(K), "hip_calls": len(Hs), "stalls": out[:20]}


if __name__ == "__main__":
    if sys.argv[1] == "steps":
        import argparse
        ap = argparse.ArgumentParser()
        ap.add_argument("cmd")
        ap.add_argument("--steps", type=int, default=50)
        ap.add_argument("--mode", default="train")
        a = ap.parse_args()
        print(json.dumps(per_step(a.mode, a.steps)))
    else:
        print(json.dumps(trace(sys.argv[2]), indent=1))
