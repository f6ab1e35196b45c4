#!/bin/bash
# N1 (render_img) on variant builds of the library: the --mode render_img line per variant, then the kernel table of the main build
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r05_render_img; mkdir -p $OUT
for v in main "$@"; do
  echo "== $v" | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $v --mode render_img --steps 80 --warmup 20 --pretrain 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print({k:d.get(k) for k in ('value','unit','ms_per_step')}, 'frac', r.get('frac'), 'decoded', d.get('config',{}).get('decoded_samples_per_pair'), 'L1', d.get('config',{}).get('depth_l1_vs_gt'))" | tee -a $OUT/lines.txt
done
cd /tmp; rm -rf /tmp/ks_r; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_r -o k -- python $REPO/bench.py --mode render_img --steps 80 --warmup 20 --pretrain 100 > $OUT/ks.log 2>&1
python $REPO/profiles/summarize_rocprof_db.py $(find /tmp/ks_r -name '*.db' | head -1) 2>&1 | head -14 | cut -c1-170 | tee $OUT/kernel_stats.txt
# HBM traffic of the render kernels (separate counter passes, --kernel-trace only)
if [ -n "$PMC" ]; then
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA"; do
    i=$((i+1)); rm -rf /tmp/pmc_ri_$i
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_ri_$i -o p -- python $REPO/bench.py --mode render_img --steps 80 --warmup 20 --pretrain 100 > /dev/null 2> $OUT/pmc_$i.err
  done
  db() { find /tmp/pmc_ri_$1 -name '*.db' 2>/dev/null | head -1; }
  python $REPO/profiles/pmc_summary.py $(db 1) $(db 2) $(db 3) $OUT/pmc_traffic.json $OUT/pmc_traffic.txt "render_img frame pair" > /dev/null
  head -12 $OUT/pmc_traffic.txt | cut -c1-170
fi
