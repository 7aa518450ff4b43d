"""Shared test helpers: deterministic weights (numpy PCG64: identical on every machine), synthetic
graphs, oracle runners."""
import numpy as np
import torch

import meshnet_oracle as mo  # oracle/ is on sys.path via conftest


def numpy_state(sd_like, seed, bn_random=True):
    """Fill a MeshNet-shaped state dict from a numpy Generator (platform independent).
    Conv/fc weights ~ U(-s,s) with the reference's scales; BN affine and running stats randomised
    so eval-mode BN is not a near-identity."""
    import zlib
    out = {}
    for k, v in sd_like.items():
        rng = np.random.default_rng([seed, zlib.crc32(k.encode())])   # per-key stream: key order is irrelevant
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_var"):
            out[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32)) if bn_random else torch.ones(shp)
        elif k.endswith("running_mean"):
            out[k] = torch.from_numpy(rng.normal(0, 0.1, shp).astype(np.float32)) if bn_random else torch.zeros(shp)
        elif k.endswith("weight") and len(shp) == 1:          # BatchNorm affine (bn.N.* / batch_normN.*)
            out[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        elif k.endswith("bias") and (".bn." in "." + k or "batch_norm" in k):
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


# ---------------------------------------------------------------------------------------------
# golden fixtures (tests/golden/*.npz, produced by the REAL reference -- see make_golden.py)
# ---------------------------------------------------------------------------------------------
import os as _os

import scipy.sparse as _sp

GOLDEN = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(_os.path.join(GOLDEN, name), allow_pickle=False)


def golden_graphs(joint_set):
    """graph_L exactly as the reference's build_coarse_graphs returned it (list of scipy CSR, float64)."""
    z = golden(f"graphs_{joint_set}.npz")
    out = []
    for i in range(int(z["num_levels"])):
        n = int(z[f"L{i}_n"])
        out.append(_sp.csr_matrix((z[f"L{i}_data"], z[f"L{i}_indices"], z[f"L{i}_indptr"]), shape=(n, n)))
    return out, z["perm0"], z["perm_reverse"]


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().flatten(), torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_vertex_l2(a, b):
    """BASELINE.json's parity metric: max over (sample, vertex) of ||v_new - v_ref||_2."""
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).norm(dim=-1).max())


def golden_regressor(name="loss_human36.npz"):
    """The reference's own joint regressor data/Human36M/J_regressor_h36m_correct.npy ((17, 6890), 107 nnz), as embedded
    (CSR triplets, fp32) in the fixtures that tests/golden/make_golden.py generated from it."""
    z = golden(name)
    R = np.zeros(tuple(int(v) for v in z["jreg_shape"]), dtype=np.float32)
    R[z["jreg_rows"], z["jreg_cols"]] = z["jreg_vals"]
    return R


def loss_case(joint_set, B=None, seed=17, jreg=None):
    """Deterministic inputs of the train-step epilogue + losses (lib/core/base.py:122-143) for one joint set:
    numpy PCG64 draws, mask shapes as the reference dataloaders emit them ([B, nv, 1] / [B, J, 1], constant per
    sample: data/Human36M/dataset.py:392-394).  Used by make_golden.py (real reference) and by the tests."""
    from pose2mesh_release_amd import synth
    gL, _, rev = golden_graphs(joint_set)
    J = int(gL[-1].shape[0])
    nv = 778 if joint_set == "mano" else 6890
    V0 = int(gL[0].shape[0])
    if B is None:
        B = 4 if joint_set == "mano" else 2
    _, faces = synth.hull_mesh(nv, 0)
    rng = np.random.default_rng(seed)

    def t(shape, scale):
        return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))
    c = {"faces": faces, "perm_reverse": np.asarray(rev), "nv": nv, "V0": V0, "J": J, "B": B,
         "J_regressor": torch.from_numpy(synth.synthetic_regressor(J, nv) if jreg is None else np.asarray(jreg, np.float32)),
         "cam_mesh": t((B, V0, 3), 0.3), "lift_pose": t((B, J, 3), 300.0), "gt_mesh": t((B, nv, 3), 0.3),
         "gt_reg3dpose": t((B, J, 3), 300.0), "gt_lift3dpose": t((B, J, 3), 300.0)}
    vm = (rng.random(B) > 0.3).astype(np.float32)
    vr = (rng.random(B) > 0.3).astype(np.float32)
    vm[0], vr[0] = 1.0, 1.0
    if B > 1:
        vm[1] = 0.0
    c["val_mesh"] = torch.from_numpy(np.repeat(vm[:, None, None], nv, axis=1).copy())
    c["val_reg3dpose"] = torch.from_numpy(np.repeat(vr[:, None, None], J, axis=1).copy())
    c["val_lift3dpose"] = torch.ones(B, J, 1)
    return c
