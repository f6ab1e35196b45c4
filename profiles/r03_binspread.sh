#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r03_binabl
mkdir -p $OUT
export PYTHONPATH=$PWD
for v in main x_bin_nostore x_bin_noatomic x_bin_neither x_bin_spread64_nostore; do
  lib=mneslam_amd/_fuzz/$v/libmneslam_hip.so
  [ $v = main ] && lib=mneslam_amd/libmneslam_hip.so
  echo "== $v" | tee -a $OUT/out.txt
  MNE_LIB_OVERRIDE=$PWD/$lib python - <<'PY' 2>&1 | tail -3 | tee -a $OUT/out.txt
import os, sys, json, subprocess
from mneslam_amd import _lib
_lib.LIB_PATH = os.environ["MNE_LIB_OVERRIDE"]
sys.argv = ["bench.py", "--steps", "60", "--warmup", "20", "--cpu-iters", "0"]
import io, contextlib
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().split("\n")[-1])
r = d["roofline"]
print("%.1f it/s %.4f ms | tile_adam %.3f |" % (d["value"], d["ms_per_step"], r["avg_launch_ms"]), {k[:14]: round(v, 3) for k, v in r["other_kernels_avg_ms"].items()})
PY
done
