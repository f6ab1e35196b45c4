mkdir -p gpurun_out
for f in 0 1 2 16 24 28; do
  echo "FLAGS=$f" >> gpurun_out/ablate.log
  MNE_DBG_FLAGS=$f python bench.py --steps 40 --warmup 10 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('  ms/step %.3f  dom=%s %.3f ms  others=%s' % (d['ms_per_step'], r['kernel'][:14], r['avg_launch_ms'], r['other_kernels_avg_ms']))
" >> gpurun_out/ablate.log
done
cat gpurun_out/ablate.log
