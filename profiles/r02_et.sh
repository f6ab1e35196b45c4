#!/bin/bash
# early ray termination + ray_kernel: parity tests, bench, kernel table, timeline; A/B without early termination and gather depth
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/pytest_gpu.log 2>&1; tail -22 gpurun_out/pytest_gpu.log
run() {  # tag, env...
  tag=$1; shift
  env "$@" python bench.py --steps 200 --warmup 30 --cpu-iters 0 2>gpurun_out/bench_$tag.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$tag ms/step %.4f it/s %.1f psnr %.2f | %s %.3f | %s | contrib %d' % (d['ms_per_step'], d['value'], d['psnr_last_iter'], r['kernel'][:12], r['avg_launch_ms'], list(r['other_kernels_avg_ms'].values()), r['contributing_samples_last_iter']))"
  rm -rf /tmp/p_$tag; env "$@" timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_$tag -o t -- python bench.py --steps 60 --warmup 10 --cpu-iters 0 > /dev/null 2>&1
  db=$(find /tmp/p_$tag -name '*.db' | head -1)
  python profiles/summarize_rocprof_db.py $db 70 2>&1 | head -20 | cut -c1-150 > gpurun_out/stats_$tag.txt
  python profiles/timeline.py $db 30 20 > gpurun_out/timeline_$tag.txt 2>&1
  head -9 gpurun_out/stats_$tag.txt
}
run et X=0
cat gpurun_out/timeline_et.txt
run noet MNE_NO_EARLY_TERMINATION=1
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
cp profiles/_variants/lib_g24.so mneslam_amd/libmneslam_hip.so
run et_g24 X=0
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
