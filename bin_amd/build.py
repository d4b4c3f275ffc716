"""Build recipe for libbinhip.so (hand-written HIP for gfx950, flat C ABI — include/binhip.h).

`hipcc --offload-arch=gfx950` cross-compiles without a GPU, so this runs in the build container; the
resulting .so stays in-tree (git-ignored) and travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["binhip_conv.hip", "binhip_fused.hip", "binhip_wgrad.hip", "binhip_misc.hip", "binhip_plan.hip"]
LIB_PATH = os.path.join(CSRC, "libbinhip.so")


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "binhip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    """Compile every HIP source for gfx950 into bin_amd/csrc/libbinhip.so."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
               os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB_PATH)
