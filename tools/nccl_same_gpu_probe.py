"""Can two ranks of the RCCL backend share ONE GPU here?  (probe for testing tiled.DistComm on a one-GPU box)"""
import os, sys, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world)
t = torch.full((4,), float(rank), device="cuda:0")
if rank == 0:
    dist.send(t, 1)
else:
    r = torch.empty_like(t); dist.recv(r, 0); print("rank1 got", r.tolist())
dist.barrier(); torch.cuda.synchronize()
print("rank", rank, "ok")
dist.destroy_process_group()
