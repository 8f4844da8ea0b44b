"""Probe: does the gloo backend all-reduce GPU tensors on this build (2 ranks on cuda:0)?  Bounded by timeouts."""
import datetime, os, socket, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=30))
    torch.cuda.set_device(0)
    x = torch.full((1000,), float(rank + 1), device='cuda')
    print('rank', rank, 'before', flush=True)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print('rank', rank, 'after', x[0].item(), flush=True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == '__main__':
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
    print('ok', flush=True)
