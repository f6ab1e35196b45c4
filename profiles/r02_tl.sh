#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/pv; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pv -o t -- python bench.py --steps 60 --warmup 10 --cpu-iters 0 > /dev/null 2>&1
db=$(find /tmp/pv -name '*.db' | head -1)
python profiles/summarize_rocprof_db.py $db 70 2>&1 | head -22 | cut -c1-150 > gpurun_out/stats_cur.txt
python profiles/timeline.py $db 30 20 > gpurun_out/timeline_cur.txt 2>&1
cat gpurun_out/stats_cur.txt gpurun_out/timeline_cur.txt
python - <<PY
import torch, bench
from mneslam_amd import configs
ag = bench.Agent(configs.bench_office0(), torch.device("cuda"), seed=0, n_keyframes=20, path="fused")
import struct
for it in range(60):
    ag.step(); 
    if it % 10 == 9:
        torch.cuda.synchronize()
        ws = ag.fused.ws
        n = int(ws[-16:-12].view(torch.int32).item())
        print("iter", it, "deferred rays", n, "contrib", int(ag.fused.tape_rows.item()))
PY
