import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
import torch.distributed._symmetric_memory as sm
try:
    t = sm.empty((1024, 100), dtype=torch.float32, device=torch.device("cuda",0))
    h = sm.rendezvous(t, dist.group.WORLD)
    print("rendezvous ok", type(h), "world", h.world_size, "rank", h.rank)
    b = h.get_buffer(0, (1024,100), torch.float32)
    b.fill_(3.0); torch.cuda.synchronize()
    print("peer buffer aliasing:", float(t[5,5]))
    h.barrier(); torch.cuda.synchronize(); print("barrier ok")
except Exception as e:
    import traceback; traceback.print_exc()
# IPC handle export of an ordinary tensor
try:
    x = torch.ones(1000, device="cuda")
    info = x.untyped_storage()._share_cuda_()
    print("share_cuda ok:", len(info))
except Exception as e:
    print("share_cuda failed:", e)
print("peer access api:", torch.cuda.device_count())
dist.destroy_process_group()
