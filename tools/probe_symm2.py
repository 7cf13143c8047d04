"""two processes on ONE GPU: can torch symmetric memory map the peer's buffer (IPC) under a gloo / nccl-less group?"""
import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world, port):
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import torch.distributed._symmetric_memory as sm
    try:
        t = sm.empty((1024, 100), dtype=torch.float32, device=torch.device("cuda", 0))
        t.fill_(float(rank + 1))
        h = sm.rendezvous(t, dist.group.WORLD)
        print(rank, "rendezvous ok", h.world_size, [hex(p) for p in h.buffer_ptrs], flush=True)
        peer = h.get_buffer(1 - rank, (1024, 100), torch.float32)
        torch.cuda.synchronize(); h.barrier(); torch.cuda.synchronize()
        print(rank, "peer value before:", float(peer[3, 3]), flush=True)
        peer[10:20].fill_(100.0 + rank)            # write into the peer's memory
        torch.cuda.synchronize(); h.barrier(); torch.cuda.synchronize()
        print(rank, "my rows after peer wrote:", float(t[10, 0]), float(t[0, 0]), flush=True)
    except Exception:
        import traceback; traceback.print_exc()
    dist.destroy_process_group()

if __name__ == "__main__":
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
