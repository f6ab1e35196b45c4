#!/bin/bash
# kernel tables of the tree under _ab/ and of this tree, same box
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
for t in _ab .; do
  tag=$(echo $t | tr -d './'); tag=${tag:-cur}
  rm -rf /tmp/ks_$tag
  (cd $REPO/$t; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o k -- python bench.py --steps 200 --warmup 20 --cpu-iters 0 $([ $t = . ] && echo --no-variants) > /dev/null 2>&1)
  d=$(find /tmp/ks_$tag -name '*.db' | head -1)
  python $REPO/profiles/summarize_rocprof_db.py $d 60 > $REPO/gpurun_out/r03_ab_stats_$tag.txt 2>&1
  python $REPO/profiles/timeline.py $d 12 40 > $REPO/gpurun_out/r03_ab_timeline_$tag.txt 2>&1
  echo "== $t"; head -24 $REPO/gpurun_out/r03_ab_stats_$tag.txt | cut -c1-150
done
