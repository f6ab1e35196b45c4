# needs an ablation build as the in-tree library: bash profiles/build_variants.sh capi.hip "abl:-DMNE_ABLATION" && cp profiles/_variants/lib_abl.so mneslam_amd/libmneslam_hip.so
mkdir -p gpurun_out; rm -f gpurun_out/ablate_tile.log
for f in 0 32 64; do
  echo "FLAGS=$f" >> gpurun_out/ablate_tile.log
  MNE_DBG_FLAGS=$f python bench.py --steps 40 --warmup 10 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('  ms/step %.3f  dom=%s %.3f ms  others=%s' % (d['ms_per_step'], r['kernel'][:14], r['avg_launch_ms'], r['other_kernels_avg_ms']))
" >> gpurun_out/ablate_tile.log
done
cat gpurun_out/ablate_tile.log
