"""dev probe: gloo all-reduce time of CUDA tensor slices of several sizes (2 ranks on one GPU)."""
import os, time, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
flat = torch.zeros(70_000_000, device="cuda")
for lo, hi in ((13_000_000, 70_000_000), (6_000_000, 28_000_000), (0, 6_000_000), (0, 29_000_000), (6_000_008, 28_000_000)):
    for rep in range(3):
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        w = dist.all_reduce(flat[lo:hi], async_op=True)
        w.wait(); torch.cuda.synchronize()
        if rank == 0:
            print("range %9d..%9d  %.1f ms" % (lo, hi, (time.perf_counter() - t0) * 1e3), flush=True)
# two in flight
for rep in range(2):
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    w1 = dist.all_reduce(flat[28_000_000:70_000_000], async_op=True)
    w2 = dist.all_reduce(flat[6_000_000:28_000_000], async_op=True)
    w3 = dist.all_reduce(flat[0:6_000_000], async_op=True)
    for w in (w1, w2, w3):
        w.wait()
    torch.cuda.synchronize()
    if rank == 0:
        print("three in flight: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
dist.destroy_process_group()
