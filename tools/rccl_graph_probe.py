"""Can an RCCL all-reduce be captured into a hipGraph by torch on this stack?  One rank (all a one-GPU box hosts).
python tools/rccl_graph_probe.py"""
import os, sys, time
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29547')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
x = torch.ones(1 << 20, device=dev)
dist.all_reduce(x); torch.cuda.synchronize()          # communicator set-up outside the capture
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        y = x * 2.0
        dist.all_reduce(y)
        z = y + 1.0
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print('captured and replayed: z[0] = %.1f (expect 3.0)' % z[0].item(), file=sys.stderr)
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    print('replay of mul + all_reduce + add: %.1f us' % ((time.perf_counter() - t0) / 200 * 1e6), file=sys.stderr)
except Exception as e:
    print('capture failed: %r' % (e,), file=sys.stderr)
dist.barrier(); dist.destroy_process_group()
