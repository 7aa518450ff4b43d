"""Training losses with the reference's names and values (lib/core/loss.py), written for the GPU:
the face index tensor is built once per device instead of being re-created and uploaded on every
call (loss.py:68,97 do `torch.LongTensor(self.face).cuda()` per forward).  Stock torch ops --
SURVEY.md section 8f ranks fused loss kernels as a later row."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class CoordLoss(nn.Module):
    """Masked mean L1 (loss.py:10-23)."""

    def __init__(self, has_valid=False):
        super().__init__()
        self.has_valid = has_valid

    def forward(self, pred, target, target_valid=None):
        if self.has_valid:
            pred, target = pred * target_valid, target * target_valid
        return (pred - target).abs().mean()


class _FaceLoss(nn.Module):
    def __init__(self, face):
        super().__init__()
        self.face = face
        self._cache = {}

    def _faces(self, device):
        f = self._cache.get(device)
        if f is None:
            f = torch.as_tensor(self.face, dtype=torch.long, device=device)
            self._cache[device] = f
        return f


class NormalVectorLoss(_FaceLoss):
    """|cos| between predicted edge directions and the GT face normal (loss.py:62-88)."""

    def forward(self, coord_out, coord_gt):
        f = self._faces(coord_out.device)
        a, b, c = coord_out[:, f[:, 0]], coord_out[:, f[:, 1]], coord_out[:, f[:, 2]]
        e = [F.normalize(b - a, p=2, dim=2), F.normalize(c - a, p=2, dim=2), F.normalize(c - b, p=2, dim=2)]
        ga, gb, gc = coord_gt[:, f[:, 0]], coord_gt[:, f[:, 1]], coord_gt[:, f[:, 2]]
        n = F.normalize(torch.cross(F.normalize(gb - ga, p=2, dim=2), F.normalize(gc - ga, p=2, dim=2), dim=2),
                        p=2, dim=2)
        cos = [(v * n).sum(2, keepdim=True).abs() for v in e]
        return torch.cat(cos, 1).mean()


class EdgeLengthLoss(_FaceLoss):
    """|edge length difference| over the three edges of every face (loss.py:91-114)."""

    def forward(self, coord_out, coord_gt):
        f = self._faces(coord_out.device)

        def lengths(x):
            a, b, c = x[:, f[:, 0]], x[:, f[:, 1]], x[:, f[:, 2]]
            return [((a - b) ** 2).sum(2, keepdim=True).sqrt(), ((a - c) ** 2).sum(2, keepdim=True).sqrt(),
                    ((b - c) ** 2).sum(2, keepdim=True).sqrt()]
        lo, lg = lengths(coord_out), lengths(coord_gt)
        return torch.cat([(o - g).abs() for o, g in zip(lo, lg)], 1).mean()


def get_loss(faces):
    """loss.py:117-120: (vertex L1, normal, edge, regressed-joint L1, lifted-joint L1)."""
    return (CoordLoss(has_valid=True), NormalVectorLoss(faces), EdgeLengthLoss(faces), CoordLoss(has_valid=True),
            CoordLoss(has_valid=True))


# ---------------------------------------------------------------------------------------------
# fused HIP path: the four mesh-side losses of the train step + their gradient in one C-ABI call
# ---------------------------------------------------------------------------------------------
import ctypes as _ct
import threading as _threading

import numpy as _np

from . import _lib as _libmod


def _mask(v, B, n, name):
    """The reference's masks are [B, n, 1] (data/*/dataset.py, consumed at lib/core/base.py:125); [B], [B,1], [B,1,1]
    and [B, n] broadcast the same way inside CoordLoss (loss.py:19).  Returns a contiguous [B, n] fp32 tensor."""
    if v is None:
        return None
    if v.dim() == 1:
        v = v.view(B, 1, 1)
    elif v.dim() == 2:
        v = v.unsqueeze(-1)
    if v.dim() != 3 or v.shape[0] != B or v.shape[1] not in (1, n) or v.shape[2] != 1:
        raise ValueError(f"{name}: expected a [B], [B,1,1] or [B,{n},1] mask, got {tuple(v.shape)} "
                         f"(per-coordinate masks are not used by the reference and not supported)")
    return v.expand(B, n, 1).reshape(B, n).contiguous().float()


class _MeshLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cam_mesh, mod, gt_mesh, valid_mesh, gt_pose, valid_pose):
        B, V0, _ = cam_mesh.shape
        dev = cam_mesh.device
        if not cam_mesh.is_cuda:
            raise _libmod.P2MError("FusedMeshLoss (HIP) needs GPU tensors; use get_loss(faces) for stock torch modules")
        t = mod._tensors(dev)
        cam = cam_mesh.contiguous().float()
        need_grad = ctx.needs_input_grad[0]
        grad = torch.empty_like(cam) if need_grad else None
        losses = torch.empty(4, device=dev, dtype=torch.float32)
        ws = torch.empty(int(_libmod.hip().p2m_mesh_loss_workspace(B, mod.nv, mod.nf, mod.J)), device=dev,
                         dtype=torch.float32)

        def p(x):
            return None if x is None else _ct.c_void_p(x.data_ptr())
        vm, vp = _mask(valid_mesh, B, mod.nv, "valid_mesh"), _mask(valid_pose, B, mod.J, "valid_pose")
        gm, gp = gt_mesh.contiguous().float(), gt_pose.contiguous().float()
        if tuple(gm.shape) != (B, mod.nv, 3) or tuple(gp.shape) != (B, mod.J, 3):
            raise ValueError(f"gt_mesh / gt_pose must be [B,{mod.nv},3] / [B,{mod.J},3]")
        with torch.cuda.device(dev):
            _libmod.check(_libmod.hip().p2m_mesh_loss(
                p(cam), V0, p(t["perm"]), mod.nv, p(gm), p(vm), p(t["faces"]), mod.nf,
                p(t["vf_ptr"]), p(t["vf_idx"]), p(t["jr_ptr"]), p(t["jr_idx"]), p(t["jr_val"]), p(t["vj_ptr"]),
                p(t["vj_idx"]), p(t["vj_val"]), mod.J, p(gp), p(vp),
                mod.w_vertex, mod.w_normal, mod.w_edge, mod.w_joint, p(ws), p(losses), p(grad), B,
                _ct.c_void_p(torch.cuda.current_stream().cuda_stream)), "p2m_mesh_loss")
        ctx.grad = grad
        ctx.mark_non_differentiable(losses)
        return losses.sum(), losses

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        return ctx.grad * g_total, None, None, None, None, None


class _CoordLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, valid, w):
        if not pred.is_cuda:
            raise _libmod.P2MError("FusedCoordLoss (HIP) needs GPU tensors; use CoordLoss for the stock torch module")
        p_, t_ = pred.contiguous().float(), target.contiguous().float()
        if p_.shape != t_.shape:
            raise ValueError(f"pred {tuple(p_.shape)} and target {tuple(t_.shape)} differ")
        n = p_.numel()
        v, per = None, 1
        if valid is not None:
            B = p_.shape[0]
            vs = tuple(valid.shape)
            per_sample = vs == (B,) or (len(vs) == p_.dim() and vs[0] == B and all(d == 1 for d in vs[1:]))
            if per_sample:                                             # [B] / [B,1,1]: one value per sample (a [J,1] mask
                v, per = valid.reshape(B).contiguous().float(), n // B   # with J == B broadcasts per JOINT: generic path)
            elif p_.dim() == 3 and vs == (B, p_.shape[1], 1):          # the reference's [B, J, 1] masks
                v, per = valid.reshape(B * p_.shape[1]).contiguous().float(), p_.shape[2]
            else:                                                      # whatever `pred * target_valid` broadcasts to
                v = valid.expand_as(p_).contiguous().float().reshape(-1)     # (raises on incompatible shapes, as torch does)
        grad = torch.empty_like(p_) if ctx.needs_input_grad[0] else None
        loss = torch.empty(1, device=p_.device, dtype=torch.float32)

        def p(x):
            return None if x is None else _ct.c_void_p(x.data_ptr())
        with torch.cuda.device(p_.device):
            _libmod.check(_libmod.hip().p2m_coord_loss(p(p_), p(t_), p(v), per, n, float(w), p(loss), p(grad),
                                                       _ct.c_void_p(torch.cuda.current_stream().cuda_stream)),
                          "p2m_coord_loss")
        ctx.grad, ctx.shape = grad, pred.shape
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g).view(ctx.shape), None, None, None


class FusedCoordLoss(nn.Module):
    """weight * CoordLoss(has_valid=True)(pred, target, valid) (lib/core/loss.py:10-23; base.py:128,139 weights the lifted-pose
    term by cfg.TRAIN.joint_loss_weight) - value and gradient from ONE launch (p2m_coord_loss) instead of a dozen
    elementwise torch kernels in forward + autograd.  valid: None, per sample ([B], [B,1,1]) or the reference's [B, J, 1]."""

    def __init__(self, weight=1.0):
        super().__init__()
        self.weight = float(weight)

    def forward(self, pred, target, target_valid=None):
        return _CoordLossFn.apply(pred, target, target_valid, self.weight)


def _regressor_tables(jr, nv):
    """CSR (by joint) and CSC (by vertex) of a dense [J, nv] joint regressor."""
    import scipy.sparse as sp
    csr = sp.csr_matrix(jr)
    csr.sort_indices()
    csc = sp.csc_matrix(jr)
    csc.sort_indices()
    i32, f32 = _np.int32, _np.float32
    return {"jr_ptr": csr.indptr.astype(i32), "jr_idx": csr.indices.astype(i32), "jr_val": csr.data.astype(f32),
            "vj_ptr": csc.indptr.astype(i32), "vj_idx": csc.indices.astype(i32), "vj_val": csc.data.astype(f32)}


class FusedMeshLoss(nn.Module):
    """loss = CoordLoss(pred_mesh, gt_mesh, valid) + w_normal*NormalVectorLoss + w_edge*EdgeLengthLoss
            + w_joint*CoordLoss(J_regressor @ (pred_mesh*1000), gt_pose, valid)
    with pred_mesh = cam_mesh[:, graph_perm_reverse[:nv]] -- i.e. lib/core/base.py:130-143 minus the PoseNet
    term -- computed (value and gradient) by p2m_mesh_loss.  Returns (total, components[4]).
    Masks take the reference's shapes ([B, nv, 1] / [B, J, 1]) or any per-sample broadcast of them.
    Set w_edge=0 before cfg.TRAIN.edge_loss_start, as the reference's epoch switch does (a zero weight also switches
    the term's gradient off, so degenerate edges cannot inject NaNs)."""

    def __init__(self, faces, perm_reverse, joint_regressor, w_normal=1e-1, w_edge=20.0, w_joint=1e-3, w_vertex=1.0):
        super().__init__()
        faces = _np.asarray(faces, dtype=_np.int64)
        self.nf = int(faces.shape[0])
        self.nv = int(faces.max()) + 1
        jr = _np.asarray(joint_regressor, dtype=_np.float32)
        self.J = int(jr.shape[0])
        assert jr.shape[1] == self.nv
        corner = _np.arange(self.nf * 3)
        vert = faces.reshape(-1)
        order = _np.argsort(vert, kind="stable")
        self._host = {
            "perm": _np.ascontiguousarray(_np.asarray(perm_reverse)[:self.nv], dtype=_np.int32),
            "faces": _np.ascontiguousarray(faces, dtype=_np.int32),
            "vf_ptr": _np.concatenate([[0], _np.cumsum(_np.bincount(vert, minlength=self.nv))]).astype(_np.int32),
            "vf_idx": _np.ascontiguousarray(corner[order], dtype=_np.int32),
        }
        self._host.update(_regressor_tables(jr, self.nv))
        self.w_vertex, self.w_normal, self.w_edge, self.w_joint = float(w_vertex), float(w_normal), float(w_edge), \
            float(w_joint)
        self._dev = {}
        self._lock = _threading.Lock()

    def _tensors(self, device):
        t = self._dev.get(device)
        if t is None:
            with self._lock:
                t = self._dev.get(device)
                if t is None:
                    t = {k: torch.from_numpy(_np.ascontiguousarray(v)).to(device) for k, v in self._host.items()}
                    self._dev[device] = t
        return t

    def forward(self, cam_mesh, gt_mesh, gt_pose, valid_mesh=None, valid_pose=None):
        return _MeshLossFn.apply(cam_mesh, self, gt_mesh, valid_mesh, gt_pose, valid_pose)


class MeshEpilogue(nn.Module):
    """The Tester's / demo's post-model steps in one launch (lib/core/base.py:200-204, demo/run.py:169-171):
        mesh   = cam_mesh[:, graph_perm_reverse[:nv], :] * scale        (scale = 1000 in the Tester, 1 in the demo)
        joints = J_regressor @ mesh                                      (sparse: CSR mat-vec)
    forward(cam_mesh [B, V0, 3]) -> (mesh [B, nv, 3], joints [B, J, 3]).  Inference only (no autograd)."""

    def __init__(self, perm_reverse, nv, joint_regressor, scale=1000.0):
        super().__init__()
        self.nv, self.scale = int(nv), float(scale)
        jr = _np.asarray(joint_regressor, dtype=_np.float32)
        self.J = int(jr.shape[0])
        assert jr.shape[1] == self.nv
        self._host = {"perm": _np.ascontiguousarray(_np.asarray(perm_reverse)[:self.nv], dtype=_np.int32)}
        self._host.update(_regressor_tables(jr, self.nv))
        self._dev = {}
        self._lock = _threading.Lock()

    _tensors = FusedMeshLoss._tensors

    @torch.no_grad()
    def forward(self, cam_mesh):
        if not cam_mesh.is_cuda:
            raise _libmod.P2MError("MeshEpilogue (HIP) needs GPU tensors")
        B, V0, _ = cam_mesh.shape
        dev = cam_mesh.device
        t = self._tensors(dev)
        cam = cam_mesh.contiguous().float()
        mesh = torch.empty((B, self.nv, 3), device=dev, dtype=torch.float32)
        joints = torch.empty((B, self.J, 3), device=dev, dtype=torch.float32)

        def p(x):
            return _ct.c_void_p(x.data_ptr())
        with torch.cuda.device(dev):
            _libmod.check(_libmod.hip().p2m_mesh_epilogue(
                p(cam), V0, p(t["perm"]), self.nv, self.scale, p(t["jr_ptr"]), p(t["jr_idx"]), p(t["jr_val"]), self.J,
                p(mesh), p(joints), B, _ct.c_void_p(torch.cuda.current_stream().cuda_stream)), "p2m_mesh_epilogue")
        return mesh, joints
