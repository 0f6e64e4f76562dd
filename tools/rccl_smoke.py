import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29511")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
x = torch.ones(1<<20, device="cuda"); dist.all_reduce(x); torch.cuda.synchronize(); print("rccl allreduce ok", float(x.sum()))
dist.destroy_process_group()
