"""Flat-buffer optimizers for MI355X: all parameters live in ONE fp32 buffer, all gradients in another, so that
(a) the optimizer step is a single fused launch (p2m_adam_step / p2m_rmsprop_step) instead of ~140 per-tensor kernels
and (b) the data-parallel gradient exchange is a handful of large RCCL all-reduces over contiguous slices (dist.py).

Both classes ARE torch.optim.Optimizer subclasses with one param group whose `lr` is read at every step(), so the
reference's schedulers and helpers run against them unchanged: `optim.lr_scheduler.MultiStepLR` (lib/funcs_utils.py:
101-104, the 'step' scheduler of every yaml recipe), `lr_check` / `lr_warmup` (funcs_utils.py:17-31, called at
lib/core/base.py:118), and `optimizer.state_dict()` / `load_state_dict()` in the torch layout that main/train.py:51-58
checkpoints.

  FlatAdam     torch.optim.Adam(params, lr)       lib/funcs_utils.py:92-96   (defaults betas (0.9, 0.999), eps 1e-8)
  FlatRMSprop  torch.optim.RMSprop(params, lr)    lib/funcs_utils.py:87-91   (defaults alpha 0.99, eps 1e-8) -- what
                                                  asset/yaml/*.yml select (`optimizer: 'rmsprop'`)
"""
import ctypes

import torch

from . import _lib, ops
from ._lib import check

_vp = ctypes.c_void_p
_HP_SLOTS = 16          # pinned staging slots of prepare_step (how far the host may run ahead of the GPU without waiting)


class _FlatOptimizer(torch.optim.Optimizer):
    _state_names = ()

    def __init__(self, params, defaults):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        dev = params[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 for p in params):
            raise ValueError(f"{type(self).__name__} needs fp32 parameters on one GPU")
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("one parameter group only (the reference builds its optimizers from model.parameters())")
        self.params = params
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.numel = n
        self.flat_param = torch.zeros(n, device=dev)
        self.flat_grad = torch.zeros(n, device=dev)
        self._bufs = {name: torch.zeros(n, device=dev) for name in self._state_names}
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat_param[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
        self.step_count = 0
        self._hp_host = self._hp_dev = None
        self._hp_events, self._hp_slot = [], 0

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value):
        self.param_groups[0]["lr"] = value

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):      # re-attach in case autograd replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    def _launch(self, grad_scale):
        raise NotImplementedError

    @torch.no_grad()
    def step(self, grad_scale=1.0, closure=None):
        """grad_scale multiplies the gradient first (1/world_size after a sum all-reduce)."""
        if callable(grad_scale):                          # torch's signature: step(closure)
            closure, grad_scale = grad_scale, 1.0
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.step_count += 1
        with torch.cuda.device(self.flat_param.device):
            self._launch(float(grad_scale))
        ops.bump_weight_epoch()       # the parameters changed behind torch's back: derived weight operands are stale
        return loss

    # ---- capturable form (train.GraphedTrainStep): the step-dependent scalars live in device memory --------------------
    def _hp_values(self, grad_scale):
        """{lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale} for the step that is about to run."""
        return [float(self.param_groups[0]["lr"]), 1.0, 1.0, float(grad_scale)]

    def prepare_step(self, grad_scale=1.0):
        """Host half of a captured step: counts the step and refreshes the four device-resident scalars (one small async
        copy from pinned memory on the current stream).  Call it right before replaying the graph that holds
        step_captured()."""
        self.step_count += 1
        with torch.cuda.device(self.flat_param.device):
            if self._hp_dev is None:
                # a RING of pinned staging slots: with graph replay the host runs many steps ahead of the GPU, and an
                # async H2D copy reads its pinned source when the GPU gets there -- one slot would be overwritten by
                # step t+k's values before step t's copy has run (wrong bias corrections / lr, nondeterministically)
                self._hp_host = torch.empty((_HP_SLOTS, 4), dtype=torch.float32).pin_memory()
                self._hp_dev = torch.empty(4, dtype=torch.float32, device=self.flat_param.device)
                self._hp_events = [None] * _HP_SLOTS
            k = self._hp_slot
            self._hp_slot = (k + 1) % _HP_SLOTS
            if self._hp_events[k] is not None:
                self._hp_events[k].synchronize()      # the copy that last read this slot has completed
            self._hp_host[k].copy_(torch.tensor(self._hp_values(grad_scale), dtype=torch.float32))
            self._hp_dev.copy_(self._hp_host[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._hp_events[k] = ev

    def _launch_captured(self):
        raise NotImplementedError

    @torch.no_grad()
    def step_captured(self):
        """Device half of a captured step: the update launch, reading lr / bias corrections / grad_scale from device
        memory (prepare_step) -- identical from step to step, hence capturable."""
        if self._hp_dev is None:
            raise RuntimeError("prepare_step() must run once before the capture")
        with torch.cuda.device(self.flat_param.device):
            self._launch_captured()
        ops.bump_weight_epoch()

    # torch-shaped state dict so reference checkpoints (main/train.py:51-58) round-trip
    def state_dict(self):
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            st = {"step": torch.tensor(float(self.step_count))}
            for name in self._state_names:
                st[name] = self._bufs[name][o:o + n].view_as(p).clone()
            state[i] = st
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        for k, v in g.items():
            if k != "params":
                self.param_groups[0][k] = tuple(v) if isinstance(v, list) else v
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            st = sd["state"].get(i)
            if st is None:
                continue
            n = p.numel()
            for name in self._state_names:
                self._bufs[name][o:o + n].copy_(st[name].reshape(-1))
            self.step_count = int(st["step"])


class FlatAdam(_FlatOptimizer):
    _state_names = ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))

    @property
    def exp_avg(self):
        return self._bufs["exp_avg"]

    @property
    def exp_avg_sq(self):
        return self._bufs["exp_avg_sq"]

    def _hp_values(self, grad_scale):
        # exactly what p2m_adam_step derives on the host: the betas as C floats, pow / sqrt in double, one rounding to float
        import ctypes
        import math
        g, t = self.param_groups[0], self.step_count
        b1, b2 = ctypes.c_float(g["betas"][0]).value, ctypes.c_float(g["betas"][1]).value
        return [float(g["lr"]), 1.0 - math.pow(b1, t), math.sqrt(1.0 - math.pow(b2, t)), float(grad_scale)]

    def _launch_captured(self):
        g = self.param_groups[0]
        check(_lib.hip().p2m_adam_step_dev(_vp(self.flat_param.data_ptr()), _vp(self.flat_grad.data_ptr()),
                                           _vp(self.exp_avg.data_ptr()), _vp(self.exp_avg_sq.data_ptr()), self.numel,
                                           _vp(self._hp_dev.data_ptr()), g["betas"][0], g["betas"][1], g["eps"],
                                           _vp(torch.cuda.current_stream().cuda_stream)), "p2m_adam_step_dev")

    def _launch(self, grad_scale):
        g = self.param_groups[0]
        check(_lib.hip().p2m_adam_step(_vp(self.flat_param.data_ptr()), _vp(self.flat_grad.data_ptr()),
                                       _vp(self.exp_avg.data_ptr()), _vp(self.exp_avg_sq.data_ptr()),
                                       self.numel, self.step_count, float(g["lr"]), g["betas"][0], g["betas"][1],
                                       g["eps"], grad_scale, _vp(torch.cuda.current_stream().cuda_stream)),
              "p2m_adam_step")


class FlatRMSprop(_FlatOptimizer):
    _state_names = ("square_avg",)

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8):
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=0, momentum=0, centered=False))

    @property
    def square_avg(self):
        return self._bufs["square_avg"]

    def _launch_captured(self):
        g = self.param_groups[0]
        check(_lib.hip().p2m_rmsprop_step_dev(_vp(self.flat_param.data_ptr()), _vp(self.flat_grad.data_ptr()),
                                              _vp(self.square_avg.data_ptr()), self.numel, _vp(self._hp_dev.data_ptr()),
                                              g["alpha"], g["eps"], _vp(torch.cuda.current_stream().cuda_stream)),
              "p2m_rmsprop_step_dev")

    def _launch(self, grad_scale):
        g = self.param_groups[0]
        check(_lib.hip().p2m_rmsprop_step(_vp(self.flat_param.data_ptr()), _vp(self.flat_grad.data_ptr()),
                                          _vp(self.square_avg.data_ptr()), self.numel, float(g["lr"]), g["alpha"],
                                          g["eps"], grad_scale, _vp(torch.cuda.current_stream().cuda_stream)),
              "p2m_rmsprop_step")


def get_optimizer(model, name="rmsprop", lr=1e-3):
    """lib/funcs_utils.py:77-98 for the two optimizers the reference's recipes use."""
    if name == "rmsprop":
        return FlatRMSprop(model.parameters(), lr=lr)
    if name == "adam":
        return FlatAdam(model.parameters(), lr=lr)
    raise ValueError(f"optimizer {name!r}: only 'rmsprop' (the yaml recipes) and 'adam' have fused kernels")
