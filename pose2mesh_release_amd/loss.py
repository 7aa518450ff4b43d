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

import numpy as _np

from . import _lib as _libmod


class _MeshLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cam_mesh, mod, gt_mesh, valid_mesh, gt_pose, valid_pose):
        B, V0, _ = cam_mesh.shape
        dev = cam_mesh.device
        t = mod._tensors(dev)
        cam = cam_mesh.contiguous().float()
        need_grad = ctx.needs_input_grad[0]
        grad = torch.empty_like(cam) if need_grad else None
        losses = torch.empty(4, device=dev, dtype=torch.float32)
        ws = torch.empty(int(_libmod.hip().p2m_mesh_loss_workspace(B, mod.nv, mod.nf, mod.J)), device=dev,
                         dtype=torch.float32)

        def p(x):
            return None if x is None else _ct.c_void_p(x.data_ptr())

        def flat(v):
            return None if v is None else v.reshape(B).contiguous().float()
        vm, vp = flat(valid_mesh), flat(valid_pose)
        with torch.cuda.device(dev):
            _libmod.check(_libmod.hip().p2m_mesh_loss(
                p(cam), V0, p(t["perm"]), mod.nv, p(gt_mesh.contiguous().float()), p(vm), p(t["faces"]), mod.nf,
                p(t["vf_ptr"]), p(t["vf_idx"]), p(t["jreg"]), mod.J, p(gt_pose.contiguous().float()), p(vp),
                mod.w_vertex, mod.w_normal, mod.w_edge, mod.w_joint, p(ws), p(losses), p(grad), B,
                _ct.c_void_p(torch.cuda.current_stream().cuda_stream)), "p2m_mesh_loss")
        ctx.grad = grad
        ctx.mark_non_differentiable(losses)
        return losses.sum(), losses

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        return ctx.grad * g_total, None, None, None, None, None


class FusedMeshLoss(nn.Module):
    """loss = CoordLoss(pred_mesh, gt_mesh, valid) + w_normal*NormalVectorLoss + w_edge*EdgeLengthLoss
            + w_joint*CoordLoss(J_regressor @ (pred_mesh*1000), gt_pose, valid)
    with pred_mesh = cam_mesh[:, graph_perm_reverse[:nv]] -- i.e. lib/core/base.py:130-143 minus the PoseNet
    term -- computed (value and gradient) by p2m_mesh_loss.  Returns (total, components[4]).
    Set w_edge=0 before cfg.TRAIN.edge_loss_start, as the reference's epoch switch does."""

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
            "jreg": _np.ascontiguousarray(jr),
        }
        self.w_vertex, self.w_normal, self.w_edge, self.w_joint = float(w_vertex), float(w_normal), float(w_edge), \
            float(w_joint)
        self._dev = {}

    def _tensors(self, device):
        t = self._dev.get(device)
        if t is None:
            t = {k: torch.from_numpy(v).to(device) for k, v in self._host.items()}
            self._dev[device] = t
        return t

    def forward(self, cam_mesh, gt_mesh, gt_pose, valid_mesh=None, valid_pose=None):
        return _MeshLossFn.apply(cam_mesh, self, gt_mesh, valid_mesh, gt_pose, valid_pose)
