cd /root/repo; export TMPDIR=/tmp
P='import json,sys
d=json.loads(sys.stdin.read().strip().split("\n")[-1]); r=d["roofline"]
print(round(d["value"],1), round(d["ms_per_step"],4), r["kernel"][:24], round(r["avg_launch_ms"],4), {k[:14]: round(v,3) for k,v in r.get("other_kernels_avg_ms",{}).items()})'
for rep in 1 2; do
 for v in main sl1024; do
  for pr in 0 -1; do
    echo -n "$v prio=$pr   "; MNE_SIDE_PRIORITY=$pr timeout 300 python profiles/r03_variant_bench.py $v --config office0_hash --steps 300 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "$P"
  done
 done
done
