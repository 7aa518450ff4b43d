"""Shared test helpers: deterministic weights (numpy PCG64: identical on every machine), synthetic
graphs, oracle runners."""
import numpy as np
import torch

import meshnet_oracle as mo  # oracle/ is on sys.path via conftest


def numpy_state(sd_like, seed, bn_random=True):
    """Fill a MeshNet-shaped state dict from a numpy Generator (platform independent).
    Conv/fc weights ~ U(-s,s) with the reference's scales; BN affine and running stats randomised
    so eval-mode BN is not a near-identity."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, v in sd_like.items():
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_var"):
            out[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32)) if bn_random else torch.ones(shp)
        elif k.endswith("running_mean"):
            out[k] = torch.from_numpy(rng.normal(0, 0.1, shp).astype(np.float32)) if bn_random else torch.zeros(shp)
        elif ".bn." in "." + k and k.endswith("weight"):
            out[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        elif ".bn." in "." + k and k.endswith("bias"):
            out[k] = torch.from_numpy(rng.normal(0, 0.1, shp).astype(np.float32))
        elif k.endswith("weight"):
            s = np.sqrt(2.0 / (shp[0] + shp[1]))
            out[k] = torch.from_numpy(rng.uniform(-s, s, shp).astype(np.float32))
        else:  # linear bias
            out[k] = torch.from_numpy(rng.normal(0, 0.05, shp).astype(np.float32))
    return out


def oracle_graphs(graph_L):
    """graph_L as build_coarse_graphs returns it -> trimmed list of torch sparse matrices."""
    return [mo.scipy_to_torch_coo(L) for L in mo.trim_graph_list(graph_L)]


def meshnet_input(B, J, seed=7):
    g = torch.Generator().manual_seed(seed)
    p2 = torch.randn(B, J, 2, generator=g)
    p2 = (p2 - p2.mean(1, keepdim=True)) / p2.std(1, keepdim=True, unbiased=False)
    p3 = torch.randn(B, J, 3, generator=g) * 0.3
    return torch.cat((p2, p3), dim=2)


def oracle_run(sd, gl_torch, x, mano, training, grad_seed=None):
    """Runs the CPU oracle; returns (out, grads dict or None, updated state)."""
    sd = {k: v.clone() for k, v in sd.items()}
    x = x.clone()
    if grad_seed is None:
        with torch.no_grad():
            out = mo.meshnet_forward(sd, gl_torch, x, mano, training)
        return out, None, sd
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    x.requires_grad_(True)
    out = mo.meshnet_forward(sd, gl_torch, x, mano, training)
    g = torch.Generator().manual_seed(grad_seed)
    w = torch.randn(out.shape, generator=g)
    (out * w).sum().backward()
    grads = {k: v.grad for k, v in sd.items() if v.requires_grad}
    grads["__input__"] = x.grad
    return out.detach(), grads, {k: v.detach() for k, v in sd.items()}
