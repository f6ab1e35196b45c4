#!/bin/bash
# INS Indoor: heavy_bwd_kernel on a second stream beside the deferred pass (default) vs behind it on the caller's stream (MNE_HEAVY_OVERLAP=0)
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_heavy; mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms;', r['kernel'][:22], round(r['avg_launch_ms']*1000,1), 'us', {k[:14]: round(v*1000,1) for k,v in r['other_kernels_avg_ms'].items()})"; }
for cfg in indoor indoor_fp16; do
for v in 0 1 0 1; do
  echo -n "$cfg overlap=$v: " | tee -a $OUT/lines.txt
  MNE_HEAVY_OVERLAP=$v timeout 300 python bench.py --config $cfg --no-variants --cpu-iters 0 --steps 300 --warmup 50 2>/dev/null | tail -1 | line | tee -a $OUT/lines.txt
done; done
for v in 0 1; do
rm -rf /tmp/pf; MNE_HEAVY_OVERLAP=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o t -- python bench.py --config indoor --no-variants --steps 100 --warmup 20 --cpu-iters 0 > /dev/null 2>&1
python profiles/timeline.py $(find /tmp/pf -name '*.db' | head -1) > $OUT/timeline_indoor_overlap$v.txt 2>&1; cat $OUT/timeline_indoor_overlap$v.txt
done
