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

    def __init__(self, params, offsets, flat_grad, bucket_bytes=64 << 20, group=None):
        self.flat_grad, self.group = flat_grad, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
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
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._hooks = []
        if self.world > 1:
            for i, p in enumerate(params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def reset(self):
        self._pending = [len(idx) for (_, _, idx) in self.buckets]
        self._handles = []

    def _make_hook(self, i):
        def hook(_param):
            b = self._bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        self._handles.append(dist.all_reduce(self.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                             async_op=True))

    def finish(self):
        """Call after backward: launches buckets whose hooks never fired (unused parameters),
        waits for all of them, and returns the scale (1/world) still to be applied to the sum."""
        if self.world > 1:
            for b, left in enumerate(self._pending):
                if left > 0:
                    self._launch(b)
            for h in self._handles:
                h.wait()
        self.reset()
        return 1.0 / self.world
