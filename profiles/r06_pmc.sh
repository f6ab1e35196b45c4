#!/bin/bash
# Round 6 counter passes + kernel tables: the default bench workload (FETCH / WRITE / SQ passes) and the variants VERDICT r03 asked
# evidence for (2x64, ScanNet colour planes, INS Indoor, fp16 plane storage) -- round 5: the SQ (matrix-pipe busy) pass for EVERY workload -- and the hash grid.
# Separate --pmc passes, --kernel-trace only (gpurun refuses anything else); variants and CPU baseline off in every traced run.
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06_pmc; mkdir -p $OUT
cd /tmp
run() {   # tag, label, bench args
  tag=$1; label=$2; shift 2
  B="python $REPO/bench.py --no-variants --cpu-iters 0 $*"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA"; do
    i=$((i+1)); rm -rf /tmp/pmc_${tag}_$i
    
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_${tag}_$i -o p -- $B --steps 10 --warmup 3 > /dev/null 2> $OUT/pmc_${tag}_$i.err
  done
  db() { find /tmp/pmc_${tag}_$1 -name '*.db' 2>/dev/null | head -1; }
  sq=$(db 3); [ -z "$sq" ] && sq=-
  python $REPO/profiles/pmc_summary.py $(db 1) $(db 2) $sq $OUT/pmc_traffic_$tag.json $OUT/pmc_traffic_$tag.txt "$label" > /dev/null
  rm -rf /tmp/ks_$tag; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o k -- $B --steps 100 --warmup 20 > $OUT/ks_$tag.log 2>&1
  d=$(find /tmp/ks_$tag -name '*.db' | head -1)
  python $REPO/profiles/summarize_rocprof_db.py $d > $OUT/kernel_stats_$tag.txt 2>&1
  python $REPO/profiles/timeline.py $d 12 40 > $OUT/timeline_$tag.txt 2>&1
  echo "== $tag: $(tail -1 $OUT/ks_$tag.log | cut -c1-160)"; head -9 $OUT/pmc_traffic_$tag.txt | tail -7 | cut -c1-140
}
run office0 replica_office0_triplane_asWired_2048x128 --config office0
run office0_2x64 replica_office0_triplane_asWired_2048x128_2x64 --config office0 --hidden 64
run scannet scannet_scene0000_colorplanes_2048x117 --config scannet
run indoor ins_indoor_agent0_triplane_2048x1045 --config indoor
run indoor_fp16 ins_indoor_agent0_triplane_fp16planes_2048x1045 --config indoor_fp16
run office0_fp16 replica_office0_triplane_fp16planes_2048x128 --config office0_fp16
# driver-form kernel table of the default workload
rm -rf /tmp/ks_drv; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_drv -o k -- python $REPO/bench.py --no-variants --cpu-iters 0 --steps 20 --warmup 5 > $OUT/ks_driver_form.log 2>&1
python $REPO/profiles/summarize_rocprof_db.py $(find /tmp/ks_drv -name '*.db' | head -1) > $OUT/kernel_stats_driver_form.txt 2>&1
cd $REPO
timeout 900 python profiles/quality_trajectory.py 300 2>&1 | grep -v amdgpu.ids > $OUT/quality_trajectory.txt; tail -3 $OUT/quality_trajectory.txt
# the hash-grid iteration (configs[1] literal): traffic passes + kernel table
cd /tmp
B="python $REPO/bench.py --config office0_hash --no-variants --cpu-iters 0"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmc_h$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_h$i -o p -- $B --steps 10 --warmup 3 > /dev/null 2> $OUT/pmc_hash_$i.err
done
python $REPO/profiles/pmc_traffic.py $(find /tmp/pmc_h1 -name '*.db' | head -1) $(find /tmp/pmc_h2 -name '*.db' | head -1) $OUT/hash_pmc_traffic.json $OUT/hash_pmc_traffic.txt > /dev/null; head -9 $OUT/hash_pmc_traffic.txt | cut -c1-150
rm -rf /tmp/ks_hash; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_hash -o k -- $B --steps 100 --warmup 20 > $OUT/ks_hash.log 2>&1
d=$(find /tmp/ks_hash -name '*.db' | head -1)
python $REPO/profiles/summarize_rocprof_db.py $d > $OUT/hash_kernel_stats.txt 2>&1
python $REPO/profiles/timeline.py $d 12 40 hash_gather_kernel > $OUT/hash_timeline.txt 2>&1
cd $REPO
# CPU baseline at 16 / 32 / 64 / 128 host threads
timeout 600 python profiles/r05_cpu_threads.py 3 2>/dev/null > $OUT/cpu_threads.txt; cat $OUT/cpu_threads.txt
# configs[4] as worded, eager vs captured graph: kernel timelines (why does the replay lose?)
cd /tmp
for g in "" "--graph two_stream"; do
  tag=indoor_fp16$( [ -n "$g" ] && echo _graph ); rm -rf /tmp/ks_$tag
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o k -- python $REPO/bench.py --config indoor_fp16 $g --no-variants --cpu-iters 0 --steps 100 --warmup 20 > $OUT/ks_$tag.log 2>&1
  d=$(find /tmp/ks_$tag -name '*.db' | head -1)
  python $REPO/profiles/timeline.py $d 12 40 > $OUT/timeline_$tag.txt 2>&1
  python $REPO/profiles/summarize_rocprof_db.py $d > $OUT/kernel_stats_$tag.txt 2>&1
  echo "== $tag: $(tail -1 $OUT/ks_$tag.log | cut -c1-120)"
done
cd $REPO
