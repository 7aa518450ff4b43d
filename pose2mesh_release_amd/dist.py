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


def ranks_seen(device=None):
    """Diagnostics for the first multi-GPU run: every rank contributes (rank, local device index) through an all-gather on
    the job's backend - RCCL when it is "nccl" - and gets the full list back.  A communicator that silently spans fewer
    ranks, or two ranks on one GPU, shows up here."""
    if not dist.is_initialized():
        return {"backend": None, "world": 1, "ranks": [0]}
    world, rank = dist.get_world_size(), dist.get_rank()
    backend = dist.get_backend()
    if device is not None:
        device = torch.device(device)                  # "cuda:1", torch.device("cuda") ... all accepted
    if device is None and torch.cuda.is_available():
        device = torch.device("cuda", torch.cuda.current_device())
    if device is not None and device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    local = device.index if (device is not None and device.type == "cuda") else -1
    # the exchange itself runs where the backend can run it: RCCL on the GPU, gloo on the host
    mine = torch.tensor([rank, local], dtype=torch.int64, device=device if backend == "nccl" else torch.device("cpu"))
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    info = {"backend": backend, "world": world, "ranks": [int(t[0]) for t in out],
            "local_devices": [int(t[1]) for t in out]}
    if backend == "nccl":
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            info["rccl_version"] = None
    return info


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
        # optional timing (bench.py --gpus N, tests): HIP events around every collective on the communication stream and
        # around finish()'s wait on the main stream -> timing_report().  Off by default: with it on, the communication
        # stream waits for each collective (one more stream dependency per bucket, no host synchronisation).
        self.timing = False
        self._t_ref = self._t_bwd_end = self._t_done = None
        self._t_buckets = []
        self._report = None
        if self.active:
            for i, p in enumerate(params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def reset(self):
        self._pending = [len(idx) for (_, _, idx) in self.buckets]
        self._handles = []
        self._seen = 0
        self.last_launch_log, self.launch_log = getattr(self, "launch_log", []), []   # one step's entries, never more
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
            if self.timing:
                ev_s, ev_e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev_s.record(self._comm)
            h = dist.all_reduce(self.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._handles.append(h)
            if self.timing:
                h.wait()           # stream wait: the communication stream (not the host) waits for the collective
                ev_e.record(self._comm)
                self._t_buckets.append((b, (e - s) * 4, self._seen, ev_s, ev_e))

    def mark_step_start(self):
        """timing: the reference point of timing_report() - call at the start of the step, on the main stream."""
        if self.timing and self.flat_grad.is_cuda:
            self._t_ref = torch.cuda.Event(enable_timing=True)
            self._t_ref.record(torch.cuda.current_stream(self.flat_grad.device))
            self._t_buckets = []

    def timing_report(self):
        """After a step with `timing` on (and mark_step_start() at its start): synchronises the device and returns
        {buckets: [{bucket, bytes, grads_seen_at_launch, start_ms, end_ms}], backward_end_ms, step_end_ms,
        allreduce_ms_total, exposed_ms, hidden_frac}: times relative to mark_step_start(); exposed = what the main stream
        waited for collectives after its own backward work was done; hidden_frac = 1 - exposed / total."""
        rep = self._report
        if rep is None or rep["ref"] is None:
            return None
        torch.cuda.synchronize(self.flat_grad.device)
        ref = rep["ref"]
        bl = [{"bucket": b, "bytes": nb, "grads_seen_at_launch": seen, "start_ms": round(ref.elapsed_time(es), 4),
               "end_ms": round(ref.elapsed_time(ee), 4)} for b, nb, seen, es, ee in rep["buckets"]]
        total = sum(x["end_ms"] - x["start_ms"] for x in bl)
        bwd_end, done = ref.elapsed_time(rep["bwd_end"]), ref.elapsed_time(rep["done"])
        exposed = max(0.0, done - bwd_end)
        return {"buckets": bl, "backward_end_ms": round(bwd_end, 4), "step_end_ms": round(done, 4),
                "allreduce_ms_total": round(total, 4), "exposed_ms": round(exposed, 4),
                "hidden_frac": round(max(0.0, 1.0 - exposed / total), 4) if total > 0 else None}

    def finish(self):
        """Call after backward: launches buckets whose hooks never fired (unused parameters),
        waits for all of them, and returns the scale (1/world) still to be applied to the sum."""
        if self.active:
            for b, left in enumerate(self._pending):
                if left > 0:
                    self._pending[b] = 0
                    self._launch(b)
            timed = self.timing and self.flat_grad.is_cuda
            if timed:
                cur = torch.cuda.current_stream(self.flat_grad.device)
                self._t_bwd_end = torch.cuda.Event(enable_timing=True)
                self._t_bwd_end.record(cur)           # the main stream's own backward work ends here
            for h in self._handles:
                h.wait()           # nccl (RCCL): the CURRENT stream waits for the collective's stream; gloo: blocks the host
            if self._comm is not None:
                torch.cuda.current_stream(self.flat_grad.device).wait_stream(self._comm)
            if timed:
                self._t_done = torch.cuda.Event(enable_timing=True)
                self._t_done.record(cur)
                self._report = {"ref": self._t_ref, "bwd_end": self._t_bwd_end, "done": self._t_done,
                                "buckets": list(self._t_buckets)}
                self._t_buckets = []
        self.reset()
        return 1.0 / self.world
