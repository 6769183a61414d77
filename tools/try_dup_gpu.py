import os, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda") * (rank + 1)
dist.all_reduce(t)
print(rank, t.tolist(), flush=True)
dist.destroy_process_group()
