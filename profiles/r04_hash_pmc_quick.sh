cd /root/repo; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
B="python $REPO/bench.py --config office0_hash --no-variants --cpu-iters 0"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmc_h$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_h$i -o p -- $B --steps 10 --warmup 3 > /dev/null 2> /tmp/pmc_$i.err
done
python $REPO/profiles/pmc_traffic.py $(find /tmp/pmc_h1 -name '*.db' | head -1) $(find /tmp/pmc_h2 -name '*.db' | head -1) /tmp/x.json /tmp/x.txt > /dev/null; head -9 /tmp/x.txt | cut -c1-150
