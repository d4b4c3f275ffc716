"""Build recipe for libbinhip.so (hand-written HIP for gfx950, flat C ABI — include/binhip.h).

`hipcc --offload-arch=gfx950` cross-compiles without a GPU, so this runs in the build container; the
resulting .so stays in-tree (git-ignored) and travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["binhip_conv.hip", "binhip_conv_x3.hip", "binhip_fused.hip", "binhip_fused_x3.hip", "binhip_wgrad.hip", "binhip_misc.hip", "binhip_plan.hip"]
LIB_PATH = os.path.join(CSRC, "libbinhip.so")


HEADER = os.path.join(os.path.dirname(HERE), "include", "binhip.h")
TUNING_SYMBOLS = ("binhip_set_variant", "binhip_set_tail_depth", "binhip_wgrad_set_debug")


def abi_symbols():
    """The entry points include/binhip.h declares (every BINHIP_API declaration), in header order."""
    import re
    with open(HEADER) as f:
        return re.findall(r"(?m)^BINHIP_API\s+[\w\s\*]+?\b(binhip_\w+)\s*\(", f.read())


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "binhip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, defines=(), out=None):
    """Compile every HIP source for gfx950 into bin_amd/csrc/libbinhip.so.

    `defines` / `out`: side builds for tools/ (e.g. defines=("BINHIP_TUNING=1",), out="tools/_abl/libbinhip_tuning.so":
    the kernel-variant / ablation switches, which the product library does not contain).  The sources are compiled in
    parallel (one hipcc per file)."""
    lib_path = out or LIB_PATH
    if out is None and not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = CSRC if out is None else os.path.dirname(os.path.abspath(out))
    os.makedirs(objdir, exist_ok=True)
    tag = "" if out is None else "." + os.path.splitext(os.path.basename(out))[0]
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", tag + ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"] + [f"-D{d}" for d in defines] + \
              ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    # Dynamic symbols = exactly the entry points include/binhip.h declares (sources are compiled -fvisibility=hidden; the
    # version script also makes the host-side kernel handles hipcc emits with default visibility local).
    vmap = os.path.join(objdir, "binhip_exports" + tag + ".map")
    names = abi_symbols() + (list(TUNING_SYMBOLS) if any(d.startswith("BINHIP_TUNING") for d in defines) else []) + \
        (["binhip_set_timeline"] if any(d.startswith("BINHIP_TIMELINE") for d in defines) else [])
    with open(vmap, "w") as f:
        f.write("{\n  global:\n" + "".join(f"    {n};\n" for n in names) + "  local: *;\n};\n")
    # -z defs: a kernel template the host pass silently failed to instantiate shows up as an undefined symbol HERE, not at dlopen
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,defs", f"-Wl,--version-script={vmap}",
           "-o", lib_path] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return lib_path


TUNING_LIB = os.path.join(os.path.dirname(HERE), "tools", "_abl", "libbinhip_tuning.so")


def build_tuning_library(verbose=True):
    """Side build with the kernel-variant / ablation switches (BINHIP_TUNING): load it with BIN_AMD_LIB=<path>."""
    return build_library(force=True, verbose=verbose, defines=("BINHIP_TUNING=1",), out=TUNING_LIB)


if __name__ == "__main__":
    if "--tuning" in sys.argv:
        print(build_tuning_library())
    else:
        build_library(force="--force" in sys.argv)
        print(LIB_PATH)
