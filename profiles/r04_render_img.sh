#!/bin/bash
# N1: render_img (bench.py --mode render_img) -- the line, then the kernel table
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r04_render_img; mkdir -p $OUT
for v in main "$@"; do
  echo "== $v"; timeout 300 python profiles/r03_variant_bench.py $v --mode render_img --steps 40 --warmup 10 2>/dev/null | tail -1 | cut -c1-420
done
cd /tmp; rm -rf /tmp/ks_r; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_r -o k -- python $REPO/bench.py --mode render_img --steps 40 --warmup 10 > $OUT/ks.log 2>&1
python $REPO/profiles/summarize_rocprof_db.py $(find /tmp/ks_r -name '*.db' | head -1) 2>&1 | head -14 | cut -c1-170 | tee $OUT/kernel_stats.txt
