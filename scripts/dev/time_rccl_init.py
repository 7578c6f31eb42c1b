import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
t0 = time.time()
import torch, torch.distributed as dist
t1 = time.time()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t2 = time.time()
x = torch.ones(1000, device="cuda")
dist.all_reduce(x); torch.cuda.synchronize()
t3 = time.time()
dist.all_reduce(x); torch.cuda.synchronize()
t4 = time.time()
print("import torch %.1f s, init_process_group %.1f s, first all_reduce %.1f s, second %.4f s; env NCCL_SOCKET_IFNAME=%s NCCL_IB_DISABLE=%s" % (
    t1 - t0, t2 - t1, t3 - t2, t4 - t3, os.environ.get("NCCL_SOCKET_IFNAME"), os.environ.get("NCCL_IB_DISABLE")), flush=True)
dist.destroy_process_group()
