#!/bin/bash
# end-of-round GPU pass: parity tests, smoke, default bench (+torchrun form), 2x64 decoder sanity, kernel trace
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 2500 gpurun_out/bench_default.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 100 --warmup 20 --cpu-iters 0 2>&1 | tail -1 | cut -c1-300
timeout 300 python bench.py --hidden 64 --steps 100 --warmup 20 --cpu-iters 0 2>&1 | tail -1 | cut -c1-400
timeout 300 python bench.py --scatter atomics --steps 100 --warmup 20 --cpu-iters 0 2>&1 | tail -1 | cut -c1-300
timeout 300 python bench.py --path autograd --steps 50 --warmup 10 --cpu-iters 0 2>&1 | tail -1 | cut -c1-300
rm -rf /tmp/pf; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o t -- python bench.py --steps 100 --warmup 20 --cpu-iters 0 > /dev/null 2>&1
python profiles/summarize_rocprof_db.py $(find /tmp/pf -name '*.db' | head -1) 120 > gpurun_out/kernel_stats_final.txt 2>&1; head -24 gpurun_out/kernel_stats_final.txt | cut -c1-160
