"""Build the TEST-ONLY host emulation of the kernels (see hip_emu.h) into tests/hostemu/_build/.
Used only by the CPU test-suite; never loaded by the mneslam_amd package."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(REPO, "mneslam_amd", "csrc")
OUT = os.path.join(HERE, "_build")
# MNE_EMU_SANITIZE=address: AddressSanitizer build in its own directory (run the tests with
# LD_PRELOAD=<clang resource dir>/lib/linux/libclang_rt.asan-x86_64.so ASAN_OPTIONS=detect_leaks=0): out-of-bounds LDS or
# global accesses of a kernel show up as heap-buffer-overflow reports with the kernel's source line
SAN = os.environ.get("MNE_EMU_SANITIZE", "")
if SAN:
    OUT = os.path.join(HERE, "_build_" + SAN)
LIB = os.path.join(OUT, "libmneslam_emu.so")
SOURCES = ["capi.hip", "render.hip", "wgrad.hip", "adam.hip", "sampler.hip", "tile_adam.hip", "gridenc.hip", "pose.hip", "encodings.hip"]


def _cxx():
    for cand in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("clang++ not found (the emulator needs ext_vector_type)")


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps += [os.path.join(HERE, "hip_emu.h"), os.path.join(REPO, "include", "mneslam_hip.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    objs = []
    for s in SOURCES:
        obj = os.path.join(OUT, s[:-4] + ".o")
        cmd = [_cxx(), "-x", "c++", "-std=c++20", "-O1", "-g", "-fPIC", "-pthread", "-DMNE_HOST_EMU",
               "-ffp-contract=off", "-Wno-unknown-pragmas", "-Wno-pass-failed",
               *((["-fsanitize=" + SAN, "-fno-omit-frame-pointer", "-DMNE_EMU_OS_THREADS"]) if SAN else []),
               "-I", HERE, "-I", CSRC, "-I", os.path.join(REPO, "include"),
               "-c", os.path.join(CSRC, s), "-o", obj]
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([_cxx(), "-shared", "-pthread", *((["-fsanitize=" + SAN, "-shared-libsan"]) if SAN else []), *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
