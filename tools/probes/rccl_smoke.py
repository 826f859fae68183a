#!/usr/bin/env python3
"""RCCL sanity on the ranks available (run under torch.distributed.run): init 'nccl' with device_id, broadcast,
all-reduce of a 12.6 MB float32 bucket (the DSVI gradient bucket size), barrier; prints the all-reduce time."""
import os
import time
import torch
import torch.distributed as dist

rank, world, lr = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
dev = torch.device('cuda', lr % torch.cuda.device_count())
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
t = torch.full((3_160_000,), float(rank + 1), device=dev)
dist.broadcast(t, src=0)
assert float(t[0]) == 1.0
for _ in range(3):
    dist.all_reduce(t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    dist.all_reduce(t)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
dist.barrier()
if rank == 0:
    print(f'world {world}: all_reduce of {t.numel() * 4 / 1e6:.1f} MB in {dt * 1e6:.0f} us; value {float(t[0]):.3g}')
dist.destroy_process_group()
