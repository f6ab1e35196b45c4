"""How many independent agents does ONE MI355X carry?  N agents as N host threads of one process (each with its own streams, planes, decoder, optimizer: the
reference's decomposition, only co-located), every agent stepping the default workload with prefetch; aggregate iterations per second.  One agent leaves HBM idle
during its latency-bound front end and the CUs half idle during its HBM-bound plane update; do two agents fill each other's gaps?
python profiles/r06_agents_per_gpu.py [workload] [steps]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mneslam_amd import configs

name = sys.argv[1] if len(sys.argv) > 1 else "office0"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dev = torch.device("cuda")
for n in (1, 2, 3, 4):
    agents = []
    for k in range(n):
        ag = bench.Agent(configs.WORKLOADS[name][0](), dev, seed=k, n_keyframes=20, path="fused")
        agents.append(ag)
    streams = [torch.cuda.Stream(dev) for _ in range(n)]
    barrier = threading.Barrier(n + 1)
    def run(k):
        with torch.cuda.stream(streams[k]):
            for _ in range(50):
                agents[k].step(prefetch=True)
            torch.cuda.synchronize()
            barrier.wait()
            for i in range(steps):
                agents[k].step(prefetch=i + 1 < steps)
            streams[k].synchronize()
            agents[k].fused.synchronize()
            torch.cuda.current_stream().synchronize()
        barrier.wait()
    th = [threading.Thread(target=run, args=(k,)) for k in range(n)]
    for t in th: t.start()
    barrier.wait(); t0 = time.perf_counter()
    barrier.wait(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    for t in th: t.join()
    q = [a.quality() for a in agents]
    print(f"{name}: {n} agent(s) on one GPU: {n * steps / dt:8.1f} it/s aggregate, {steps / dt:8.1f} per agent; psnr {[round(x[0], 1) for x in q]}; mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    del agents
    torch.cuda.empty_cache()
