"""Data parallelism for the train step: one process per GPU, RCCL all-reduce over xGMI.

The reference uses single-process nn.DataParallel (lib/core/base.py:108): every step it broadcasts
all 76 M parameters, gathers outputs to GPU 0 and reduces gradients there.  Here each rank owns a
full replica and a contiguous flat gradient buffer (optim.FlatAdam); gradients are summed with a few
large all-reduces over slices of that buffer, launched from post-accumulate hooks as soon as every
parameter of a bucket has its gradient, so they overlap with the rest of backward.  BatchNorm
statistics stay per-rank, which is what DataParallel replicas do (no SyncBN in the reference).
xGMI is point-to-point (7 links x ~153 GB/s per GPU): ring all-reduce is per-link bound, so buckets
are large (default 64 MiB) -- latency, not bandwidth, is what small buckets would pay for.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK/WORLD_SIZE/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = os.environ.get("P2M_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        ndev = torch.cuda.device_count()
        if local >= ndev:
            if backend == "nccl":                                      # "nccl" IS RCCL on ROCm
                raise RuntimeError(f"LOCAL_RANK {local} but only {ndev} GPU(s): RCCL needs one GPU per rank "
                                   "(P2M_DIST_BACKEND=gloo allows several ranks per GPU for functional tests)")
            local = local % ndev
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class BucketedAllReduce:
    """Sum-all-reduce of a flat gradient buffer in buckets, overlapped with backward.

    params/offsets describe where each parameter's gradient lives inside `flat_grad`
    (as laid out by optim.FlatAdam; works for any flat buffer, CPU+gloo included)."""

    def __init__(self, params, offsets, flat_grad, bucket_bytes=64 << 20, group=None, always=False):
        """always: run the collectives even in a 1-rank group (functional tests of the RCCL path on one GPU)."""
        self.flat_grad, self.group = flat_grad, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (always and dist.is_initialized())
        self.buckets = []          # (start, end, [param indices])
        cur_start, cur_idx = None, []
        limit = max(1, bucket_bytes // 4)
        order = sorted(range(len(params)), key=lambda i: offsets[i])
        for i in order:
            if cur_start is None:
                cur_start = offsets[i]
            cur_idx.append(i)
            end = offsets[i] + params[i].numel()
            if end - cur_start >= limit:
                self.buckets.append((cur_start, end, cur_idx))
                cur_start, cur_idx = None, []
        if cur_idx:
            self.buckets.append((cur_start, offsets[cur_idx[-1]] + params[cur_idx[-1]].numel(), cur_idx))
        self._bucket_of = {}
        for b, (_, _, idx) in enumerate(self.buckets):
            for i in idx:
                self._bucket_of[i] = b
        self._index_of = {id(p): i for i, p in enumerate(params)}
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._hooks = []
        self._comm = None
        self.launch_log = []       # per step: (bucket, gradients reported so far when it was launched) - for the tests
        self._seen = 0
        if self.active:
            for i, p in enumerate(params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def reset(self):
        self._pending = [len(idx) for (_, _, idx) in self.buckets]
        self._handles = []
        self._seen = 0
        self._streams = [set() for _ in self.buckets]      # streams that carry gradient-writing kernels of each bucket
        self._done = set()

    def _arrived(self, i):
        # once per parameter and step: torch fires the post-accumulate hook of a parameter even when the backward
        # returned None for it (in-place accumulation), i.e. AFTER notify() has already reported it
        if i in self._done:
            return
        self._done.add(i)
        b = self._bucket_of[i]
        self._pending[b] -= 1
        self._seen += 1
        if self.flat_grad.is_cuda:
            self._streams[b].add(torch.cuda.current_stream(self.flat_grad.device))
        if self._pending[b] == 0:
            self._launch(b)

    def _make_hook(self, i):
        def hook(_param):
            self._arrived(i)
        return hook

    def notify(self, params):
        """For gradients that are accumulated IN PLACE into their slices of the flat buffer (Pose2Mesh.
        accumulate_grads_in_place: autograd never sees them, so no hook fires): the backward reports a layer's parameters
        right after enqueueing the kernels that write their gradients, ON THE STREAM those kernels run on (the caller is
        inside that stream's context).  A bucket whose last gradient arrives this way is launched at once - the
        collective is ordered after the current stream's work - instead of waiting for finish()."""
        if not self.active:
            return
        for p in params:
            i = self._index_of.get(id(p))
            if i is not None:
                self._arrived(i)

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        self.launch_log.append((b, self._seen))
        if not self.flat_grad.is_cuda:
            self._handles.append(dist.all_reduce(self.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))
            return
        # The gradients of one bucket are written by kernels on SEVERAL streams (weight gradients on the backward's side
        # stream, BatchNorm gradients and autograd's accumulations on the main stream).  The collective is issued from a
        # communication stream that waits for all of them - not from whichever stream reported last, which would order it
        # after that stream only, and not from the main stream after a wait_stream(side), which would stall the backward.
        dev = self.flat_grad.device
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        for st in self._streams[b] | {cur}:
            self._comm.wait_stream(st)
        with torch.cuda.stream(self._comm):
            self._handles.append(dist.all_reduce(self.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))

    def finish(self):
        """Call after backward: launches buckets whose hooks never fired (unused parameters),
        waits for all of them, and returns the scale (1/world) still to be applied to the sum."""
        if self.active:
            for b, left in enumerate(self._pending):
                if left > 0:
                    self._pending[b] = 0
                    self._launch(b)
            for h in self._handles:
                h.wait()           # nccl (RCCL): the CURRENT stream waits for the collective's stream; gloo: blocks the host
            if self._comm is not None:
                torch.cuda.current_stream(self.flat_grad.device).wait_stream(self._comm)
        self.reset()
        return 1.0 / self.world
