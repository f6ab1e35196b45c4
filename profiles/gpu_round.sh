# one GPU call: parity tests, bench, kernel trace.  usage: bash profiles/gpu_round.sh <tag> [pytest-filter]
tag=${1:-x}
mkdir -p gpurun_out
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q ${2:+-k "$2"} > gpurun_out/pytest_$tag.log 2>&1
tail -5 gpurun_out/pytest_$tag.log
timeout 300 python bench.py --steps 200 --warmup 30 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
tail -1 gpurun_out/bench_$tag.json
rm -rf /tmp/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o trace -- python bench.py --steps 60 --warmup 10 --cpu-iters 0 > /dev/null 2> gpurun_out/prof_$tag.err
db=$(find /tmp/prof_$tag -name '*.db' | head -1)
python profiles/summarize_rocprof_db.py $db > gpurun_out/kernel_stats_$tag.txt 2>&1
head -30 gpurun_out/kernel_stats_$tag.txt
