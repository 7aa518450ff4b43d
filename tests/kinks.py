"""Test infrastructure: ReLU-kink-resolved gradient comparison against the float64 oracle.

Why.  In train mode the network is (conv -> BatchNorm -> ReLU) x 14-20.  An element whose pre-activation lies within
fp32 rounding of 0 gets mask 0 under one evaluation order and mask 1 under another; the forward value is unaffected
(|y| ~ 1e-6) but the backward loses / gains that element's WHOLE gradient path, which moves every upstream gradient by
~1e-3 relative.  The reference's own fp32 CPU arithmetic does this too: the fp32 oracle differs from the float64 oracle
by 2-4e-3 on all upstream tensors (1.3e-2 on bn.8.bias) for human36 B=3, and by 1e-6 when no element happens to flip
(tools/probes/grad_diag.py).  A plain tolerance therefore either hides real errors (2e-2) or is
flaky (1e-4).

What this does instead: it ACCOUNTS for the flips.  The implementation under test reports, for every ReLU layer, the
mask it actually used (the module's `_tap` test hook hands out the raw conv output and the BatchNorm scale / shift; the
mask fmaf(y, scale, shift) > 0 is reproduced exactly in float64, because an fp32 x fp32 product is exact in double and
one rounding never changes a sign).  The float64 oracle is then run with THOSE masks.  What must hold:
  * every element whose mask differs from the oracle's own (y64 > 0) is a genuine kink element: |y64| <= 1e-4
    (a wrong mask from a real bug would sit at O(1));
  * the number of such elements is reported (typically 0-10 of millions);
  * with the masks aligned, every gradient tensor agrees with float64 to fp32 round-off (1e-5 rel-L2).
"""
import torch

import meshnet_oracle as mo


class _MaskedReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x * mask

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


class _FProxy:
    """torch.nn.functional with relu replaced (only inside the oracle module, only during one call)."""

    def __init__(self, masks, stats):
        self._masks, self._stats, self._n = masks, stats, 0

    def __getattr__(self, name):
        return getattr(torch.nn.functional, name)

    def relu(self, x, inplace=False):
        lid = self._n
        self._n += 1
        own = x.detach() > 0
        # masks may arrive bit-unpacked from a child process (padded to a multiple of 8 elements)
        m = own if self._masks is None else self._masks[lid].flatten()[:x.numel()].reshape(x.shape)
        diff = m != own
        n = int(diff.sum())
        self._stats["n_relu_elements"] += x.numel()
        self._stats["n_flips"] += n
        if n:
            self._stats["max_abs_preact_at_flip"] = max(self._stats["max_abs_preact_at_flip"],
                                                        float(x.detach().abs()[diff].max()))
            self._stats["flips_per_layer"][lid] = n
        return _MaskedReLU.apply(x, m)


def hip_masks(tap):
    """tap: list of (conv index, y_raw [M, F], scale [F], shift [F]) from Pose2Mesh._tap -> list of bool CPU tensors in
    layer order.  Exactly the kernels' fmaf(y, scale, shift) > 0 (k_bn_act_fwd / k_bn_bwd_*), see the module docstring."""
    out = []
    for ci, y, sc, sh in sorted(tap, key=lambda t: t[0]):
        out.append(((y.double() * sc.double() + sh.double()) > 0).cpu())
    return out


def masked_oracle_gradients(sd32, graph_L_torch32, x32, mano, w32, masks, dtype=torch.float64):
    """Float64 oracle forward + backward with the ReLU masks forced to `masks` (None: its own).
    Returns (out, grads dict incl. '__input__', stats)."""
    sd = {k: (v.to(dtype).clone() if v.dtype.is_floating_point else v.clone()) for k, v in sd32.items()}
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k]
    for k in names:
        sd[k].requires_grad_(True)
    x = x32.to(dtype).clone().requires_grad_(True)
    stats = {"n_relu_elements": 0, "n_flips": 0, "max_abs_preact_at_flip": 0.0, "flips_per_layer": {}}
    old_F = mo.F
    mo.F = _FProxy(masks, stats)
    try:
        out = mo.meshnet_forward(sd, [g.to(dtype) for g in graph_L_torch32], x, mano, True)
    finally:
        mo.F = old_F
    (out * w32.to(dtype)).sum().backward()
    grads = {k: sd[k].grad for k in names}
    grads["__input__"] = x.grad
    return out.detach(), grads, stats
