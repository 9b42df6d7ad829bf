import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
t=torch.zeros(1,device="cuda:0")
for name,fn in [("barrier",lambda: dist.barrier()),("barrier dev",lambda: dist.barrier(device_ids=[0])),("allreduce",lambda: dist.all_reduce(t))]:
    fn(); torch.cuda.synchronize()
    for _ in range(3):
        t0=time.perf_counter(); fn(); torch.cuda.synchronize(); print(name, "%.3f ms"%((time.perf_counter()-t0)*1e3))
dist.destroy_process_group()
