cd /root/repo; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
for c in indoor scannet; do
rm -rf /tmp/ks_i; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_i -o k -- python $REPO/bench.py --config $c --no-variants --cpu-iters 0 --steps 100 --warmup 20 > /tmp/ks_i.log 2>&1
echo "== $c"; python $REPO/profiles/timeline.py $(find /tmp/ks_i -name '*.db' | head -1) 12 40 | head -24
done
