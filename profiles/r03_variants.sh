#!/bin/bash
# usage: r03_variants.sh OUTNAME variant...   (variant = main | name under mneslam_amd/_fuzz); 200-step office0 line per library
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export PYTHONPATH=$PWD
for v in "$@"; do
  lib=mneslam_amd/_fuzz/$v/libmneslam_hip.so
  [ $v = main ] && lib=mneslam_amd/libmneslam_hip.so
  echo "== $v" | tee -a $OUT/out.txt
  MNE_LIB_OVERRIDE=$PWD/$lib python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $OUT/out.txt
import os, sys, json, io, contextlib
from mneslam_amd import _lib
_lib.LIB_PATH = os.environ["MNE_LIB_OVERRIDE"]
extra = os.environ.get("BENCH_ARGS", "").split()
sys.argv = ["bench.py", "--steps", "200", "--warmup", "20", "--cpu-iters", "0"] + extra
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().split("\n")[-1])
r = d["roofline"]
print("  %.1f it/s %.4f ms | %s %.3f |" % (d["value"], d["ms_per_step"], r["kernel"][:16], r["avg_launch_ms"]), {k[:18]: round(v, 3) for k, v in r["other_kernels_avg_ms"].items()})
PY
done
