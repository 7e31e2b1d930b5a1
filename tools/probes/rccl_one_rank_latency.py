#!/usr/bin/env python3
"""Floor of the small-message all-reduce cost over RCCL: ONE rank (the only GPU a test box has), the two message sizes of config 3
(actor 33 KB, critic 116 KB), asynchronous handles waited for by stream order as PPOLearner.update does.  A one-rank all-reduce moves no
data between GPUs: what is timed is the library's launch path (enqueue + its kernel), i.e. a lower bound of the per-message latency L in
docs/KERNEL_NOTES.md section 6."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
for n in (8141, 28873 + 8):
    buf = torch.zeros(n, device="cuda")
    for _ in range(20):
        dist.all_reduce(buf, async_op=True).wait()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(200):
        dist.all_reduce(buf, async_op=True).wait()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{4 * n / 1024:.0f} KB: {1e3 * e0.elapsed_time(e1) / 200:.1f} us per all-reduce on the stream, {1e6 * (t1 - t0) / 200:.1f} us of host time per call")
dist.destroy_process_group()
