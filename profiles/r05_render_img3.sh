#!/bin/bash
# N1 (render_img): the software-pipelined frame decode (decode_pipe_kernel) -- off, 8 waves, 12 waves per CU
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r05_render_img3; mkdir -p $OUT
line() {  # label, variant, extra args
  echo -n "$1: " | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $2 --mode render_img --steps 80 --warmup 20 --pretrain 100 $3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print(round(d['ms_per_step'],2), 'ms/pair frac', round(r.get('frac',0),3), 'decoded', d.get('config',{}).get('decoded_samples_per_pair'), 'L1', d.get('config',{}).get('depth_l1_vs_gt'))" | tee -a $OUT/lines.txt
}
for v in "$@"; do line "office0 $v" $v ""; done
KS=${KS:-main}
cd /tmp; rm -rf /tmp/ks_r; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_r -o k -- python $REPO/profiles/r03_variant_bench.py $KS --mode render_img --steps 80 --warmup 20 --pretrain 100 > $OUT/ks.log 2>&1
python $REPO/profiles/summarize_rocprof_db.py $(find /tmp/ks_r -name '*.db' | head -1) 2>&1 | head -8 | cut -c1-170 | tee $OUT/kernel_stats_$KS.txt
