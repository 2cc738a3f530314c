import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29611"
    dist.init_process_group("gloo", rank=rank, world_size=2)
    torch.cuda.set_device(0)
    t = torch.full((1024,), float(rank + 1), device="cuda")
    try:
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(rank, "gloo cuda all_reduce ok", t[0].item(), flush=True)
    except Exception as e:
        print(rank, "gloo cuda all_reduce FAILED:", repr(e)[:200], flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, nprocs=2)
