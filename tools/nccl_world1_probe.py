#!/usr/bin/env python3
"""Does an asynchronous all-reduce at world size 1 block the HOST?  Enqueue ~20 ms of GPU work, then
time the host side of dist.all_reduce(async_op=True) on a 57 MB bucket and of work.wait().  (A host
that cannot run ahead of the GPU shows up as idle gaps in the --force-dp step.)"""
import os
import time
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.randn(14_251_156, device="cuda")
a = torch.randn(8192, 8192, device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    for _ in range(6):
        b = a @ a                                   # ~7 ms each on the fp32 pipe
    t1 = time.perf_counter()
    w = dist.all_reduce(x, async_op=True)
    t2 = time.perf_counter()
    w.wait()
    t3 = time.perf_counter()
    y = x * 2
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    print(f"enqueue matmuls {1e3*(t1-t0):.2f} ms | all_reduce(async) call {1e3*(t2-t1):.3f} ms | work.wait() "
          f"{1e3*(t3-t2):.3f} ms | next op enqueue {1e3*(t4-t3):.3f} ms | drain {1e3*(t5-t4):.2f} ms", flush=True)
dist.destroy_process_group()
