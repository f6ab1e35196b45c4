"""Build libmneslam_hip.so for gfx950 with hipcc (in-tree, next to the sources).

    python -m mneslam_amd.build            # build if sources are newer than the library
    python -m mneslam_amd.build --force

hipcc cross-compiles without a GPU.  The library is the ONLY compute backend of the package; there
is no CPU fallback (mneslam_amd/_lib.py raises when it is missing).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libmneslam_hip.so")
SOURCES = ["capi.hip", "render.hip", "wgrad.hip", "adam.hip", "sampler.hip", "tile_adam.hip", "gridenc.hip"]
HEADERS = ["mne_device.h", "mne_launch.h", "mne_platform.h", "mlp_mfma.h"]

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-munsafe-fp-atomics", "-fgpu-rdc-off-placeholder"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libmneslam_hip.so")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "mneslam_hip.h")]
    if not force and not _stale(LIB, deps):
        return LIB
    flags = [f for f in HIPCC_FLAGS if f != "-fgpu-rdc-off-placeholder"]
    objs = []
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "mneslam_hip.h")]
    for s in srcs:
        obj = s[:-4] + ".o"
        if force or _stale(obj, [s] + hdrs):
            cmd = [_hipcc(), *flags, "-I", INCLUDE, "-I", CSRC, "-c", s, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
