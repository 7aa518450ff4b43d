"""TEST INFRASTRUCTURE ONLY -- loader for the *real* reference implementation.

Imports hongsukchoi/Pose2Mesh_RELEASE from /root/reference (read-only) so that
 (a) the CPU restatement in oracle/meshnet_oracle.py / oracle/coarsen_oracle.py can be
     validated against the real code, and
 (b) tests/golden/make_golden.py can generate the committed fixtures.

/root/reference does NOT exist on the GPU box; nothing under `-m gpu`, smoke() or bench.py
calls this module.  Three shims are needed:
  1. fake `core.config` exposing the 3 cfg keys the model files read
     (lib/models/meshnet.py:21, lib/models/pose2mesh_net.py:13, lib/models/posenet.py:89);
     the real lib/core/config.py needs easydict and mkdirs under the read-only tree.
  2. empty `cv2` module (lib/funcs_utils.py:6 imports it; never used on this path).
  3. Tensor.cuda -> identity while the reference forward runs (lib/models/meshnet.py:81
     hard-codes .cuda(); lib/core/loss.py:68,97 too).
load_loss() additionally imports lib/core/loss.py by path (the faked `core` package has no __path__), and
load_aug() imports lib/aug_utils.py / lib/coord_utils.py with a numpy `cv2.getAffineTransform` (exact 3-point
solve in float64, what OpenCV computes) and cfg.MODEL.input_shape = (384, 288) (lib/core/config.py:52).
"""
import contextlib
import io
import os
import sys
import types

REF_ROOT = os.environ.get("P2M_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "lib", "models"))


class _Cfg:  # attribute bag
    pass


_loaded = {}


def load(target_joint_set=None):
    """Returns a namespace with the reference modules: coarsening, graph_utils, meshnet,
    pose2mesh_net, posenet, cheby (graph_conv_cheby) and the fake cfg."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    if "ns" in _loaded:
        if target_joint_set is not None:
            _loaded["ns"].cfg.DATASET.target_joint_set = target_joint_set
        return _loaded["ns"]
    lib = os.path.join(REF_ROOT, "lib")
    if lib not in sys.path:
        sys.path.insert(0, lib)
    cfg = _Cfg()
    cfg.DATASET = _Cfg()
    cfg.DATASET.target_joint_set = target_joint_set or "human36"
    cfg.MODEL = _Cfg()
    cfg.MODEL.posenet_pretrained = False
    cfg.MODEL.posenet_path = ""
    cfg.MODEL.input_shape = (384, 288)            # lib/core/config.py:52
    cfg.MODEL.normal_loss_weight = 1e-1           # lib/core/config.py:53-55
    cfg.MODEL.edge_loss_weight = 20
    cfg.MODEL.joint_loss_weight = 1e-3
    core = types.ModuleType("core")
    core.__path__ = []
    core_config = types.ModuleType("core.config")
    core_config.cfg = cfg
    core.config = core_config
    sys.modules.setdefault("core", core)
    sys.modules.setdefault("core.config", core_config)
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with contextlib.redirect_stdout(io.StringIO()):
            import coarsening  # noqa
            import graph_utils  # noqa
            from models import meshnet, pose2mesh_net, posenet  # noqa
            from models.backbones import cheby_graph_conv  # noqa
    ns = types.SimpleNamespace(coarsening=coarsening, graph_utils=graph_utils, meshnet=meshnet,
                               pose2mesh_net=pose2mesh_net, posenet=posenet,
                               cheby=cheby_graph_conv, cfg=cfg)
    _loaded["ns"] = ns
    return ns


@contextlib.contextmanager
def cpu_cuda_shim():
    """Make Tensor.cuda() a no-op so the reference forward (meshnet.py:81) runs on CPU."""
    import torch
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = orig


def build_coarse_graphs(faces, joint_num, skeleton, flip_pairs, levels):
    ns = load()
    with contextlib.redirect_stdout(io.StringIO()):
        return ns.graph_utils.build_coarse_graphs(faces, joint_num, skeleton, flip_pairs, levels=levels)


def load_loss():
    """The real lib/core/loss.py module (CoordLoss, NormalVectorLoss, EdgeLengthLoss, get_loss)."""
    load()
    if "loss" in _loaded:
        return _loaded["loss"]
    import importlib.util
    import warnings
    spec = importlib.util.spec_from_file_location("core.loss", os.path.join(REF_ROOT, "lib", "core", "loss.py"))
    mod = importlib.util.module_from_spec(spec)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with contextlib.redirect_stdout(io.StringIO()):
            spec.loader.exec_module(mod)
    _loaded["loss"] = mod
    return mod


def _get_affine_transform_numpy(src, dst):
    """cv2.getAffineTransform: the 2x3 matrix M with M [x y 1]^T = dst for three point pairs (float64 result)."""
    import numpy as np
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    A = np.concatenate([src, np.ones((3, 1))], axis=1)          # 3 x 3
    return np.linalg.solve(A, dst).T                             # 2 x 3


def load_aug():
    """The real lib/aug_utils.py and lib/coord_utils.py (j2d_processing, get_bbox, process_bbox).  OpenCV is not
    installed here: the faked cv2 module gets a numpy getAffineTransform (the only cv2 call on this path,
    lib/aug_utils.py:170-173)."""
    load()
    if "aug" in _loaded:
        return _loaded["aug"]
    sys.modules["cv2"].getAffineTransform = _get_affine_transform_numpy
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with contextlib.redirect_stdout(io.StringIO()):
            import aug_utils  # noqa
            import coord_utils  # noqa
    ns = types.SimpleNamespace(aug_utils=aug_utils, coord_utils=coord_utils)
    _loaded["aug"] = ns
    return ns
