#!/bin/bash
# N1 (render_img): Z-order ray walk on / off, XCD task numbering on / off, one-set forward decode (SEQF) on a colour-plane scene
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r05_render_img2; mkdir -p $OUT
line() {  # label, variant, env, extra args
  echo -n "$1: " | tee -a $OUT/lines.txt
  env $3 timeout 300 python profiles/r03_variant_bench.py $2 --mode render_img --steps 80 --warmup 20 --pretrain 100 $4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print(round(d['ms_per_step'],2), 'ms/pair frac', round(r.get('frac',0),3), 'decoded', d.get('config',{}).get('decoded_samples_per_pair'), 'L1', d.get('config',{}).get('depth_l1_vs_gt'))" | tee -a $OUT/lines.txt
}
line "office0 zorder+xcd" main X=1 ""
line "office0 rowmajor+xcd" main MNE_RENDER_PATCH_ORDER=0 ""
line "office0 zorder, no xcd map" xcd0 X=1 ""
line "office0 zorder+xcd (again)" main X=1 ""
line "scannet zorder seqf" main X=1 "--config scannet"
line "scannet zorder seqf0" seqf0 X=1 "--config scannet"
line "scannet rowmajor seqf" main MNE_RENDER_PATCH_ORDER=0 "--config scannet"
cd /tmp; rm -rf /tmp/ks_r; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_r -o k -- python $REPO/bench.py --mode render_img --steps 80 --warmup 20 --pretrain 100 > $OUT/ks.log 2>&1
python $REPO/profiles/summarize_rocprof_db.py $(find /tmp/ks_r -name '*.db' | head -1) 2>&1 | head -8 | cut -c1-170 | tee $OUT/kernel_stats.txt
if [ -n "$PMC" ]; then
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA"; do
    i=$((i+1)); rm -rf /tmp/pmc_ri_$i
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_ri_$i -o p -- python $REPO/bench.py --mode render_img --steps 80 --warmup 20 --pretrain 100 > /dev/null 2> $OUT/pmc_$i.err
  done
  db() { find /tmp/pmc_ri_$1 -name '*.db' 2>/dev/null | head -1; }
  python $REPO/profiles/pmc_summary.py $(db 1) $(db 2) $(db 3) $OUT/pmc_traffic.json $OUT/pmc_traffic.txt "render_img frame pair" > /dev/null
  head -6 $OUT/pmc_traffic.txt | cut -c1-170
fi
