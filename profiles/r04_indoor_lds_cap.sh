cd /root/repo; export TMPDIR=/tmp
P='import json,sys
d=json.loads(sys.stdin.read().strip().split("\n")[-1]); r=d["roofline"]
print(round(d["value"],1), round(d["ms_per_step"],4), {k[:14]: round(v,3) for k,v in r.get("other_kernels_avg_ms",{}).items()})'
for rep in 1 2; do
for cap in 256 128 192 384 512 2048; do
  echo -n "cap=$cap  "; MNE_HOT_LDS_SAMPLES=$cap timeout 300 python bench.py --config indoor --steps 200 --warmup 30 --cpu-iters 0 --no-variants 2>/dev/null | python -c "$P"
done; done
