import os, torch, torch.distributed as dist
rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
try:
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    t = torch.ones(4, device='cuda') * (rank + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print('rank', rank, 'allreduce ok', t.tolist(), flush=True)
except Exception as e:
    print('rank', rank, 'FAILED', type(e).__name__, str(e)[:300], flush=True)
