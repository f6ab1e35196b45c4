"""Build libmneslam_hip.so for gfx950 with hipcc (in-tree, next to the sources).

    python -m mneslam_amd.build            # build if sources are newer than the library
    python -m mneslam_amd.build --force
    python -m mneslam_amd.build --fuzz     # + the layout-fuzz variants the GPU tests load (mneslam_amd/_fuzz/)

hipcc cross-compiles without a GPU.  The library is the ONLY compute backend of the package; there
is no CPU fallback (mneslam_amd/_lib.py raises when it is missing).

Variants (``build_variant``): the same sources with extra -D switches, objects and library in their own directory.
``FUZZ_VARIANTS`` are the kernel-argument layout perturbations tests/test_layout_fuzz_gpu.py runs the 2x64 + colour-plane
cases against: a kernel never reads the padding, so its results must not depend on it.
"""
import json
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

from . import isa_check

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libmneslam_hip.so")
FUZZ_DIR = os.path.join(HERE, "_fuzz")
SOURCES = ["capi.hip", "render.hip", "wgrad.hip", "adam.hip", "sampler.hip", "tile_adam.hip", "gridenc.hip", "pose.hip", "encodings.hip"]
HEADERS = ["mne_device.h", "mne_launch.h", "mne_platform.h", "mlp_mfma.h"]

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics"]

# name -> extra defines.  Only translation units that see RenderArgs change with MNE_ARGS_PAD; everything is rebuilt
# anyway so that a variant is one self-contained library.
FUZZ_VARIANTS = {"pad8": ["-DMNE_ARGS_PAD=8"], "pad16": ["-DMNE_ARGS_PAD=16"]}


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


def _deps():
    return [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "mneslam_hip.h")]


def isa_report_path(lib):
    return os.path.join(os.path.dirname(lib), "isa_report.json")


def _compile_one(src, obj, extra, verbose):
    """hipcc -c with -save-temps: the device assembly of the very compile that produced the object is kept next to it
    (<name>.gfx950.s) for the ISA checks; every other temporary is removed."""
    name = os.path.basename(src)[:-4]
    tmp = os.path.join(os.path.dirname(obj), "_tmp_" + name)
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    cmd = [_hipcc(), *HIPCC_FLAGS, *extra, "-I", INCLUDE, "-I", CSRC, "-save-temps=obj", "-c", src, "-o", os.path.join(tmp, name + ".o")]
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.run(cmd, check=True, stderr=subprocess.PIPE, text=True)
    except subprocess.CalledProcessError as e:
        sys.stderr.write(e.stderr)
        raise
    asm = [f for f in os.listdir(tmp) if f.endswith("-gfx950.s")]
    shutil.move(os.path.join(tmp, name + ".o"), obj)
    if asm:
        shutil.move(os.path.join(tmp, asm[0]), obj[:-2] + ".gfx950.s")
    shutil.rmtree(tmp, ignore_errors=True)


def check_isa(obj_dir, lib, strict=True):
    """Run the ISA checks (isa_check.py) over the device assembly of every translation unit of a build; writes
    isa_report.json next to the library; raises if a kernel carries the spill-in-front-of-exec-restore defect."""
    report = {"kernels": [], "hazards": []}
    for s in SOURCES:
        path = os.path.join(obj_dir, s[:-4] + ".gfx950.s")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: the build keeps the device assembly for the ISA checks")
        text = open(path).read()
        for r in isa_check.resources(text):
            report["kernels"].append(dict(r, source=s))
        for kernel, (line, block, spill, restore, _) in isa_check.definite_hazards(text):
            report["hazards"].append({"source": s, "kernel": kernel, "line": line, "block": block, "spill": spill, "before": restore})
    with open(isa_report_path(lib), "w") as f:
        json.dump(report, f, indent=1)
    if report["hazards"] and strict and os.environ.get("MNE_ALLOW_SPILL_HAZARD", "0") != "1":
        msg = "; ".join(f"{h['kernel']} ({h['source']}:{h['line']}: {h['spill']} <before> {h['before']})" for h in report["hazards"])
        raise RuntimeError("compiler defect in the generated ISA (VGPR spill store in front of an exec restore, DESIGN.md section 9): " + msg)
    return report


def _compile_all(obj_dir, lib, extra, force, verbose, jobs):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = _deps()
    if not force and not _stale(lib, srcs + hdrs) and os.path.exists(isa_report_path(lib)):
        return lib
    os.makedirs(obj_dir, exist_ok=True)
    todo, objs = [], []
    for s in srcs:
        obj = os.path.join(obj_dir, os.path.basename(s)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [s] + hdrs) or not os.path.exists(obj[:-2] + ".gfx950.s"):
            todo.append((s, obj))
    if todo:
        with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
            list(ex.map(lambda so: _compile_one(so[0], so[1], extra, verbose), todo))
    check_isa(obj_dir, lib)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


def build(force=False, verbose=False, jobs=None):
    return _compile_all(CSRC, LIB, [], force, verbose, jobs or (os.cpu_count() or 4))


def variant_path(name):
    return os.path.join(FUZZ_DIR, name, "libmneslam_hip.so")


def build_variant(name, defines, force=False, verbose=False, jobs=None):
    d = os.path.join(FUZZ_DIR, name)
    return _compile_all(d, variant_path(name), list(defines), force, verbose, jobs or (os.cpu_count() or 4))


def build_fuzz(force=False, verbose=False):
    return [build_variant(n, d, force, verbose) for n, d in FUZZ_VARIANTS.items()]


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--fuzz" in sys.argv:
        for p in build_fuzz(force="--force" in sys.argv, verbose=True):
            print(p)
