#!/bin/bash
# NS-b: half-precision plane storage: parity tests + bench lines (office0, indoor; eager and graph)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/fp16_planes.txt; : > $out
timeout 900 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "fp16 or graph_replay" 2>&1 | tail -4
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  %-52s ms/step %.4f  it/s %.1f psnr %.2f | %s %.3f | %s' % (d['config']['workload'], d['ms_per_step'], d['value'], d['psnr_last_iter'], r['kernel'][:12], r['avg_launch_ms'], ['%s %.3f' % (k[:10], v) for k, v in r['other_kernels_avg_ms'].items()]))"; }
for c in office0 indoor apartment scannet; do for ps in fp32 fp16; do
  timeout 300 python bench.py --config $c --plane-storage $ps --steps 150 --warmup 20 --cpu-iters 0 2>/dev/null | line >> $out
done; done
echo "graph (MNE_GRAPH=1):" >> $out
for ps in fp32 fp16; do MNE_GRAPH=1 timeout 300 python bench.py --config indoor --plane-storage $ps --steps 150 --warmup 20 --cpu-iters 0 2>/dev/null | line >> $out; done
cat $out
