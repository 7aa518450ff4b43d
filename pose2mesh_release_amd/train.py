"""The whole train step as ONE captured hipGraph.

One step of the reference's loop (lib/core/base.py:122-148: zero_grad, forward, losses, backward, optimizer step) is
~400 kernel launches; enqueueing them from Python costs 7-8 ms, which is the step time itself below batch ~100
(DESIGN.md section 6 "Host side").  Everything in that step has static shapes and lives on the device -- the flat
optimizers own parameters, gradients and moments, the step-dependent scalars (lr, Adam's bias corrections) are read from
device memory -- so the step can be captured once and replayed:

    step = GraphedTrainStep(model, opt, lambda: total_loss(model(pose2d_static), ...))     # loss_fn does forward + loss
    for batch in loader:
        pose2d_static.copy_(batch["pose2d"]); ...                   # refresh the static input tensors
        loss = step()                                                # a device scalar (overwritten by the next call)

The first `warmup` calls run eagerly (they are real training steps: graphs, caches and the allocator settle), the next
one captures, later ones replay.  Results are bitwise those of the eager loop (tests/test_gpu_train.py).  Changing
param_groups[0]["lr"] (MultiStepLR) needs no re-capture.  Not supported: gradient all-reduce inside the step, BatchNorm
momentum=None (its update reads a host counter), anything in loss_fn that synchronises with the host.
"""
import torch

from . import ops as _ops
from . import optim as _optim


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_fn, warmup=3, grad_scale=1.0):
        if not isinstance(optimizer, _optim._FlatOptimizer):
            raise TypeError("GraphedTrainStep needs optim.FlatAdam / optim.FlatRMSprop (device-resident optimizer state)")
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        # at least one eager step on the capture stream: graph handles, per-stream scratch buffers and the optimizer's
        # device scalars must exist before the capture (hipMalloc / synchronous copies are illegal inside one)
        self.warmup, self.grad_scale = max(1, int(warmup)), float(grad_scale)
        self.device = optimizer.flat_param.device
        self.stream = torch.cuda.Stream(device=self.device)
        self.graph, self.loss, self.calls = None, None, 0

    def _eager(self):
        self.opt.zero_grad()
        loss = self.loss_fn()
        loss.backward()
        self.opt.step(self.grad_scale)
        return loss.detach()

    def __call__(self):
        with torch.cuda.device(self.device):
            return self._call()

    def _call(self):
        cur = torch.cuda.current_stream(self.device)
        self.calls += 1
        if self.graph is None and self.calls <= self.warmup:
            # on the capture stream, so that per-stream scratch and the allocator's pools are the ones the capture will see
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                loss = self._eager()
            cur.wait_stream(self.stream)
            return loss
        if self.graph is None:
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self.opt.prepare_step(self.grad_scale)         # the scalars must exist before the capture reads them
                self.opt.step_count -= 1                        # (this call's real prepare_step follows below)
                self.opt.zero_grad()
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with _ops.capture_guard(), torch.cuda.graph(g, stream=self.stream):   # (no finalizer may run inside the capture)
                self.opt.zero_grad()
                loss = self.loss_fn()
                loss.backward()
                self.opt.step_captured()
                self.loss = loss.detach()
            self.graph = g
        self.opt.prepare_step(self.grad_scale)                  # host: step counter, lr, bias corrections -> device
        self.graph.replay()
        # the replay updated parameters and BatchNorm running statistics by raw pointer: derived operands cached by
        # ops.WeightCache (packed / split weights, eval-mode BatchNorm coefficients) are stale for any EAGER forward that
        # follows (the reference's train-epoch / validation-epoch loop); the captured kernels rebuild their own copies
        _ops.bump_weight_epoch()
        return self.loss
