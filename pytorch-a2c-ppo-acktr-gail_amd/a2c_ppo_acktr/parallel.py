"""Data-parallel sharding of the update path: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).  New in this build -- the reference is single device (main.py:39).

Scheme: the rollout storage is replicated; every rank draws the SAME minibatch permutation
(CPU generator, same seed) and takes a contiguous slice of each minibatch.  Per-sample loss terms are
scaled by 1/B_global, so a SUM all-reduce of the flat gradient bucket (one 3.47 MB message for NatureCNN)
yields exactly the full-minibatch gradient; clip-norm + optimiser then run replicated on identical data,
so parameters never need a broadcast.
"""
import torch
import torch.distributed as dist


class DataParallel(object):
    def __init__(self, group=None, force_active=False):
        """force_active: take the data-parallel code path (sharding arithmetic, collectives) even with one rank --
        lets a one-GPU box drive the real RCCL calls."""
        self.group = group
        self._force = force_active
        if dist.is_available() and dist.is_initialized():
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        # gloo is a host transport: device buckets are staged through host memory here, synchronously, instead of
        # through gloo's own device path (which was seen to hang when two ranks share one GPU, the only way a one-GPU
        # box can rehearse world_size 2).  RCCL ("nccl") reduces the device bucket in place, stream-ordered.
        self._host_staged = dist.is_available() and dist.is_initialized() and dist.get_backend(group) == 'gloo'
        self._piecewise = None          # a PiecewiseGraph being captured: collectives split it instead of running

    @property
    def active(self):
        return self.world > 1 or self._force

    def bounds(self, n):
        """Contiguous balanced split of n items: [lo, hi) of this rank."""
        return (n * self.rank) // self.world, (n * (self.rank + 1)) // self.world

    def shard(self, idx):
        """This rank's slice of a minibatch index tensor."""
        if not self.active:
            return idx
        lo, hi = self.bounds(idx.numel())
        return idx[lo:hi]

    def all_reduce_sum_(self, flat):
        """In-place SUM all-reduce of one flat bucket (gradients, KFAC factors, loss accumulators)."""
        if self.active:
            if self._piecewise is not None:             # capture records kernels only: the collective becomes an eager
                self._piecewise.split(lambda: self.all_reduce_sum_(flat))      # step between two graph pieces
                return flat
            if self._host_staged and flat.is_cuda:
                host = flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                flat.copy_(host)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def all_reduce_sum_many_(self, flats):
        """Consecutive buckets (the two KFAC factor buckets): one eager step between two graph pieces, not an empty piece."""
        if self.active and self._piecewise is not None:
            self._piecewise.split(lambda: [self.all_reduce_sum_(f) for f in flats])
            return
        for f in flats:
            self.all_reduce_sum_(f)

    def all_reduce_sum_async_(self, flat):
        """Start the SUM all-reduce of a bucket and return a handle whose .wait() orders the current stream after it.
        RCCL runs it on its own stream, so kernels launched meanwhile overlap with the transfer; the host-staged
        gloo form (CPU tests, one-GPU rehearsals) simply finishes it here."""
        if not self.active:
            return _Done()
        if self._host_staged and flat.is_cuda:
            self.all_reduce_sum_(flat)
            return _Done()
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def broadcast_(self, t, src):
        if self.active:
            if self._host_staged and t.is_cuda:
                host = t.cpu()
                dist.broadcast(host, src=src, group=self.group)
                t.copy_(host)
            else:
                dist.broadcast(t, src=src, group=self.group)
        return t


class PiecewiseGraph(object):
    """An update captured as hipGraph pieces with eager steps (the collectives) between them.  RCCL calls cannot be part
    of a torch-captured graph here (they run on RCCL's own stream behind event edges the capture does not own), and the
    host-staged gloo form synchronises; so under data parallelism the launch-bound stretches between two collectives are
    captured one graph each, all from ONE memory pool (a later piece reads what an earlier one wrote), and replay() runs
    graph, collective, graph, ... in capture order on the current stream.  Capture records only: nothing has run and no
    collective has been issued when capture() returns."""

    def __init__(self, device):
        self.device = device
        self.items = []                 # torch.cuda.CUDAGraph | callable
        self._pool = None
        self._cur = None
        self._stream = None

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        g.capture_begin(pool=self._pool)
        self._cur = g

    def _end(self):
        self._cur.capture_end()
        self.items.append(self._cur)
        self._cur = None

    def split(self, fn):
        """Called from inside capture(): close the current piece, queue `fn` as an eager step, open the next piece."""
        self._end()
        self.items.append(fn)
        self._begin()

    def capture(self, dp, body):
        import gc
        torch.cuda.synchronize(self.device)
        gc.collect()                                    # nothing may be freed by the collector mid-capture (see
        gc_was_on = gc.isenabled()                      # ops.graph_capture): collect now, then keep it off
        gc.disable()
        cur = torch.cuda.current_stream(self.device)
        self._stream = torch.cuda.Stream(device=self.device)
        self._stream.wait_stream(cur)
        dp._piecewise = self
        try:
            with torch.cuda.stream(self._stream):
                self._begin()
                try:
                    body()
                finally:
                    self._end()
        finally:
            dp._piecewise = None
            if gc_was_on:
                gc.enable()
        cur.wait_stream(self._stream)
        return self

    def replay(self):
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()


def bounded_sync(device, seconds, what):
    """Wait for everything queued on `device`'s current stream, but not for ever: a collective captured into a hipGraph
    that never completes with N ranks (a peer missing, a mismatched replay order) shows no error, the stream just never
    drains.  After `seconds` the process says what it was waiting for and exits with status 70, so that a launcher
    (torch.distributed.run, bench.py's own) records a failure instead of hanging until its own limit."""
    import os
    import sys
    import time
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    t0 = time.monotonic()
    while not ev.query():
        if time.monotonic() - t0 > seconds:
            sys.stderr.write('a2c_ppo_acktr: %s did not complete within %.0f s on %s (rank %s of %s); giving up\n'
                             % (what, seconds, device, os.environ.get('RANK', '0'), os.environ.get('WORLD_SIZE', '1')))
            sys.stderr.flush()
            os._exit(70)
        time.sleep(0.0005)


def quiesce_collectives(device, seconds=0.35):
    """Call before capturing RCCL collectives into a hipGraph (ACKTR_DP_GRAPH=full).  torch's process-group watchdog polls
    the end events of the eager collectives it has not yet retired (every 100 ms); on HIP, querying an event whose stream
    has MEANWHILE entered capture -- RCCL's stream joins ours through the capture's event edges -- fails with "operation
    not permitted on an event last recorded in a capturing stream" and the watchdog takes the process down (seen once in
    three runs of tests/test_dp_gpu.py).  So: let the device finish, then give the watchdog time for two polls, after which
    it holds nothing recorded on that stream; collectives issued during capture are not handed to it."""
    import time
    torch.cuda.synchronize(device)
    time.sleep(seconds)


class _Done(object):
    def wait(self):
        return True


def assign_eig_owners(sizes, world):
    """Layer-wise sharding of the KFAC eigendecompositions (kfac.py:205-214) over ranks: greedy
    longest-processing-time on an n^3 cost model.  Deterministic, identical on every rank.
    Returns a list owner[i] in [0, world)."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    load = [0.0] * world
    owner = [0] * len(sizes)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += float(sizes[i]) ** 3
    return owner
