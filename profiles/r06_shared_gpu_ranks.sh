#!/bin/bash
# The multi-agent paths of bench.py executed as REAL processes on the HIP library, N ranks sharing the box's one GPU over gloo (MNE_SHARE_GPUS=1):
# functional evidence for the process-per-agent data path (barriers, timing rule, decoder all-reduce, overlap-rectangle exchange with one and two
# neighbours, all-agent plane reduction) -- not a scaling measurement, and RCCL itself is not exercised.
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_shared; mkdir -p $OUT; export MNE_SHARE_GPUS=1 MNE_SIDE_RECORD_LIMIT_S=1500
show() { python -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{\"metric\"')]
if not l: print('NO LINE'); sys.exit()
d=json.loads(l[-1]); c=d['config']
print(d['n_gpus'], 'ranks', round(d['value'],1), 'it/s', c['workload'], '|', c['collective_backend'], '| ranks_seen', c['ranks_seen'], '|', c['parallelism'][:110])
print('   data:', d['data'][:150])
for r in (d.get('per_rank') or []): print('   rank', r['rank'], round(r['it_per_s'],1), 'it/s; plane params', r['plane_params'], '; psnr', round(r['psnr_last_iter'],2), 'L1', round(r['depth_l1_last_iter'],4), '; exchange B/iter', r['overlap_exchange_bytes_per_iter'])
for k,v in (d.get('variants') or {}).items():
    if not isinstance(v, dict): print('   side record', k, v); continue
    print('   side record', k, {x: (round(v[x],1) if isinstance(v[x],float) else v[x]) for x in v if x in ('value','workload','error','baseline_config','exchange_bytes_per_iter_all_ranks')})
    for r in (v.get('per_rank') or []): print('      rank', r['rank'], round(r['it_per_s'],1), 'it/s psnr', round(r['psnr_last_iter'],2), 'L1', round(r['depth_l1_last_iter'],4), 'exchange B/iter', r['overlap_exchange_bytes_per_iter'])
"; }
echo "== 2 independent agents + side records (shared decoder; configs[2] as worded: apartment split 2-way)" | tee $OUT/lines.txt
timeout 900 python bench.py --gpus 2 --steps 60 --warmup 10 --cpu-iters 0 2>$OUT/g2.err | tee $OUT/g2.json | show | tee -a $OUT/lines.txt
echo "== configs[3] as worded: ScanNet scene0000 split 4-way (interior agents: two neighbours)" | tee -a $OUT/lines.txt
timeout 900 python bench.py --gpus 4 --split --config scannet --steps 30 --warmup 5 --cpu-iters 0 --no-variants 2>$OUT/g4.err | tee $OUT/g4.json | show | tee -a $OUT/lines.txt
echo "== configs[4] geometry: INS Indoor split 8-way" | tee -a $OUT/lines.txt
timeout 900 python bench.py --gpus 8 --split --config indoor --steps 10 --warmup 3 --cpu-iters 0 --no-variants 2>$OUT/g8.err | tee $OUT/g8.json | show | tee -a $OUT/lines.txt
tail -q -n 3 $OUT/g2.err $OUT/g4.err $OUT/g8.err | grep -v "amdgpu.ids\|hostname" | tail -12
