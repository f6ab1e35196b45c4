#!/bin/bash
# second look at MNE_ADAM_NT on another box, in the forms the records use: 200 steps and the driver's 20 steps
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r05_adam_nt2; mkdir -p $OUT
for v in nt0 main nt0 main; do
  for form in "--steps 200 --warmup 20" "--steps 20 --warmup 5"; do
    echo -n "$v $form: " | tee -a $OUT/lines.txt
    timeout 60 python profiles/r03_variant_bench.py $v --no-variants --cpu-iters 0 $form 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s; tile_adam', round(r['avg_launch_ms']*1000,1), 'us')" | tee -a $OUT/lines.txt
  done
done
