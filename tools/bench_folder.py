"""End-to-end folder run (PNG decode -> device -> net -> PNG encode) on a synthetic 720p clip at several IO thread counts:
what bench.py's `harness` leg reports, as a sweep.  usage: python tools/bench_folder.py [--frames 81] [--precision f16x3]"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=81)
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--threads", default="4,8,12,16")
    args = ap.parse_args()
    import bench
    for t in [int(x) for x in args.threads.split(",")]:
        h = bench.harness_bench(args.precision, args.frames, None, io_threads=t)
        print(json.dumps({k: h[k] for k in ("io_threads", "frames_per_s", "steady_state_frames_per_s", "wall_s", "windows",
                                            "net_and_glue_ms_per_window")}), flush=True)


if __name__ == "__main__":
    main()
