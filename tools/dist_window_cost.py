"""Measurement tool: what the collective side of bench.py's timed window costs with the RCCL backend (one rank is all a 1-GPU box
offers: launch, stream hand-over and synchronisation costs of the collective path, not its wire time): torch.distributed.barrier(),
the [world,18] all_reduce + host copy of dist.run_and_select, and the whole window with / without the closing barrier."""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from diffdope_amd import dist as ddist, workloads as wl
w = wl.build("cfg2", dev)
eng, params = wl.engine_for(w, wl.bench_lr_schedule(400, "adam"), optimizer="adam")
eng.run(20); torch.cuda.synchronize()
for _ in range(8):
    eng.rewind(20); eng.run(100)
torch.cuda.synchronize()
def med(f, n=25):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(ts))
x = torch.zeros((1, 18), device=dev)
print(f"torch.distributed.barrier()                      {med(lambda: dist.barrier()):8.1f} us")
print(f"all_reduce [1,18] + .cpu()                        {med(lambda: (dist.all_reduce(x), x.cpu())):8.1f} us")
print(f"all_reduce [1,18] + torch.cuda.synchronize()      {med(lambda: (dist.all_reduce(x), torch.cuda.synchronize())):8.1f} us")
def win(closing_barrier, select=True):
    eng.rewind(20)
    if select: ddist.run_and_select(eng, 20)
    else: eng.run(20)
    if closing_barrier: dist.barrier()
    torch.cuda.synchronize()
print(f"run(20) + synchronize                             {med(lambda: win(False, False)):8.1f} us")
print(f"run_and_select(20) [all_reduce inside] + sync     {med(lambda: win(False)):8.1f} us")
print(f"run_and_select(20) + dist.barrier() + sync        {med(lambda: win(True)):8.1f} us")
dist.destroy_process_group()
