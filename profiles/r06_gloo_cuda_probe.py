"""Do torch.distributed's gloo collectives / point-to-point operations accept device tensors when two ranks SHARE one GPU?  (What a
one-GPU box can execute of the multi-agent data path: RCCL refuses two ranks on one device.)
python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 profiles/r06_gloo_cuda_probe.py"""
import os
import torch
import torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
t = torch.full((1024,), float(rank + 1), device=dev)
try:
    dist.all_reduce(t)
    print(rank, "all_reduce on a device tensor:", float(t[0]))
except Exception as e:
    print(rank, "all_reduce failed:", type(e).__name__, str(e)[:200])
s = torch.full((4096,), float(rank + 10), device=dev)
r = torch.zeros_like(s)
try:
    ops = [dist.P2POp(dist.isend, s, 1 - rank), dist.P2POp(dist.irecv, r, 1 - rank)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    print(rank, "batch_isend_irecv on device tensors:", float(r[0]))
except Exception as e:
    print(rank, "batch_isend_irecv failed:", type(e).__name__, str(e)[:200])
try:
    out = [None, None]
    dist.all_gather_object(out, {"rank": rank})
    print(rank, "all_gather_object:", out)
    dist.barrier()
    print(rank, "barrier ok")
except Exception as e:
    print(rank, "object collective failed:", type(e).__name__, str(e)[:200])
dist.destroy_process_group()
