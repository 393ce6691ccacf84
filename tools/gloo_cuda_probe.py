"""does torch.distributed's gloo backend move CUDA (HIP) tensors on this build?  two ranks sharing GPU 0"""
import os, sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def w(rank, world):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    t = torch.tensor([float(rank + 1)], device=dev, dtype=torch.float64)
    try:
        dist.all_reduce(t, op=dist.ReduceOp.MAX); print(rank, "all_reduce cuda", t.item())
        b = torch.full((1000,), float(rank), device=dev); dist.broadcast(b, 0); print(rank, "broadcast cuda", b[0].item())
        own = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(own, torch.tensor([float(rank)], dtype=torch.float64, device=dev)); print(rank, "all_gather cuda", [x.item() for x in own])
        dist.barrier(); print(rank, "barrier ok")
    except Exception as e:
        print(rank, "FAILED", repr(e)[:300])
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(w, args=(2,), nprocs=2)
