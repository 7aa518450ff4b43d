"""Flat-buffer optimizer for MI355X: all parameters live in ONE fp32 buffer, all gradients in
another, so that (a) the Adam step is a single fused launch (p2m_adam_step) instead of ~140
per-tensor kernels and (b) the data-parallel gradient exchange is a handful of large RCCL
all-reduces over contiguous slices (see dist.py).  Semantics = torch.optim.Adam as the reference
builds it (lib/funcs_utils.py:92-96: lr only, default betas/eps, no weight decay)."""
import ctypes

import torch

from . import _lib
from ._lib import check

_vp = ctypes.c_void_p


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("FlatAdam needs fp32 parameters on one GPU")
        self.lr, self.betas, self.eps = lr, betas, eps
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.numel = n
        self.flat_param = torch.zeros(n, device=dev)
        self.flat_grad = torch.zeros(n, device=dev)
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat_param[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
        self.step_count = 0

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):      # re-attach in case autograd replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    def step(self, grad_scale=1.0):
        self.step_count += 1
        with torch.cuda.device(self.flat_param.device):
            check(_lib.hip().p2m_adam_step(_vp(self.flat_param.data_ptr()), _vp(self.flat_grad.data_ptr()),
                                           _vp(self.exp_avg.data_ptr()), _vp(self.exp_avg_sq.data_ptr()),
                                           self.numel, self.step_count, self.lr, self.betas[0], self.betas[1],
                                           self.eps, float(grad_scale),
                                           _vp(torch.cuda.current_stream().cuda_stream)), "p2m_adam_step")

    # torch.optim.Adam-shaped state dict so reference checkpoints (main/train.py:51-58) round-trip
    def state_dict(self):
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[o:o + n].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(p).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            st = sd["state"].get(i)
            if st is None:
                continue
            n = p.numel()
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(st["step"])
