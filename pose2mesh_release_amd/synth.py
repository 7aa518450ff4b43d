"""Synthetic stand-ins for the licensed SMPL/MANO assets (none ship with the reference):
sphere-hull meshes with SMPL's / MANO's exact vertex, edge and face counts, the reference's joint
skeletons, and benchmark inputs normalised like the datasets do."""
import numpy as np

H36M_SKELETON = ((0, 7), (7, 8), (8, 9), (9, 10), (8, 11), (11, 12), (12, 13), (8, 14), (14, 15), (15, 16),
                 (0, 1), (1, 2), (2, 3), (0, 4), (4, 5), (5, 6))                 # data/Human36M/dataset.py:56-59
H36M_FLIP = ((1, 4), (2, 5), (3, 6), (14, 11), (15, 12), (16, 13))
COCO_SKELETON = ((1, 2), (0, 1), (0, 2), (2, 4), (1, 3), (6, 8), (8, 10), (5, 7), (7, 9), (12, 14), (14, 16),
                 (11, 13), (13, 15), (17, 11), (17, 12), (17, 18), (18, 5), (18, 6), (18, 0))   # demo/run.py:88-91
COCO_FLIP = ((1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16))
MANO_SKELETON = ((0, 1), (0, 5), (0, 9), (0, 13), (0, 17), (1, 2), (2, 3), (3, 4), (5, 6), (6, 7), (7, 8), (9, 10),
                 (10, 11), (11, 12), (13, 14), (14, 15), (15, 16), (17, 18), (18, 19), (19, 20))   # demo/run.py:110
MANO_HORI = ((1, 5), (5, 9), (9, 13), (13, 17), (2, 6), (6, 10), (10, 14), (14, 18), (3, 7), (7, 11), (11, 15),
             (15, 19), (4, 8), (8, 12), (12, 16), (16, 20))                      # data/FreiHAND/dataset.py:38-40

JOINT_SETS = {
    "human36": (17, H36M_SKELETON, H36M_FLIP, 9),
    "coco": (19, COCO_SKELETON, COCO_FLIP, 9),
    "mano": (21, MANO_SKELETON, MANO_HORI, 6),
}


def hull_mesh(num_vertex, seed=0):
    """Convex hull of `num_vertex` random unit vectors: a closed triangle mesh with V=num_vertex,
    F=2V-4, E=3V-6 (6890 -> 13776 faces / 20664 edges, exactly SMPL's counts)."""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(seed)
    p = rng.standard_normal((num_vertex, 3))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    return p.astype(np.float32), ConvexHull(p).simplices.astype(np.int64)


def make_graphs(joint_set="human36", num_vertex=None, seed=0):
    """(faces, graph_L list as build_coarse_graphs returns it, perm_reverse, num_joint)."""
    from . import graph_utils
    J, skel, flip, levels = JOINT_SETS[joint_set]
    if num_vertex is None:
        num_vertex = 778 if joint_set == "mano" else 6890
    _, faces = hull_mesh(num_vertex, seed)
    _, graph_L, _, perm_rev = graph_utils.build_coarse_graphs(faces, J, skel, flip, levels=levels)
    return faces, graph_L, perm_rev, J


def pose2d_batch(B, J, seed=123):
    """N(0,1) joints, then per-sample per-axis standardisation over joints
    (data/Human36M/dataset.py:387-388)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, J, 2, generator=g)
    x = (x - x.mean(dim=1, keepdim=True)) / x.std(dim=1, keepdim=True, unbiased=False)
    return x


def synthetic_regressor(J, nv, seed=5):
    """Sparse row-stochastic joint regressor like data/Human36M/J_regressor_h36m_correct.npy ((17, 6890), 107 nnz):
    6 vertices per joint, weights summing to 1."""
    rng = np.random.default_rng(seed)
    R = np.zeros((J, nv), dtype=np.float32)
    for j in range(J):
        idx = rng.choice(nv, size=6, replace=False)
        w = rng.random(6).astype(np.float32)
        R[j, idx] = w / w.sum()
    return R
