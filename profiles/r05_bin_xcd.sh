#!/bin/bash
# TIMING EXPERIMENT (results of the variant are wrong by construction): bin_kernel with one list cursor per (list, XCD) -- what would
# the appends cost if the same-address queue on the hot lists were divided by eight?  Read bin_kernel's time only.
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
for c in office0 scannet indoor; do
  for v in main binxcd; do
  if [ $v = binxcd ]; then export MNE_BIN_XCD_EXPERIMENT=1; else unset MNE_BIN_XCD_EXPERIMENT; fi
  python profiles/r03_variant_bench.py $v --config $c --steps 100 --warmup 20 --cpu-iters 0 --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
o = dict(r['other_kernels_avg_ms']); o[r['kernel']] = r['avg_launch_ms']
print('$c $v | ' + ' '.join('%s=%.3f' % (k.split(' ')[0], v) for k, v in o.items() if 'bin' in k or 'ray_kernel' in k))"
done; done
