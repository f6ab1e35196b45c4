#!/bin/bash
# frame kernels (decode_frame_kernel + ray_frame_kernel) on the record; training decode at 12 waves (spills) as a side question
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r05_frame_final; mkdir -p $OUT
timeout 200 python -m pytest tests/test_hip_parity_gpu.py -x -q -k "render_maps or forward or hash_scene or frame" 2>&1 | tail -2 | tee $OUT/tests.txt
line() {
  echo -n "$1: " | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $2 --mode render_img --steps 80 --warmup 20 --pretrain 100 $3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print(round(d['ms_per_step'],2), 'ms/pair frac', round(r.get('frac',0),3), 'decoded', d.get('config',{}).get('decoded_samples_per_pair'), 'L1', d.get('config',{}).get('depth_l1_vs_gt'))" | tee -a $OUT/lines.txt
}
line "office0 main" main ""
line "office0 main (again)" main ""
line "scannet main" main "--config scannet"
for v in main dec12 main dec12; do
  echo -n "office0 training $v: " | tee -a $OUT/lines.txt
  timeout 300 python profiles/r03_variant_bench.py $v --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')" | tee -a $OUT/lines.txt
done
cd /tmp; rm -rf /tmp/ks_r; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_r -o k -- python $REPO/bench.py --mode render_img --steps 80 --warmup 20 --pretrain 100 > $OUT/ks.log 2>&1
python $REPO/profiles/summarize_rocprof_db.py $(find /tmp/ks_r -name '*.db' | head -1) 2>&1 | head -8 | cut -c1-170 | tee $OUT/kernel_stats.txt
