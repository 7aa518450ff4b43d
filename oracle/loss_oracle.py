"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the mesh losses and the post-model epilogue of the train /
test step.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Restates, operator by operator (torch CPU fp32, same evaluation order):
  CoordLoss            lib/core/loss.py:10-23
  NormalVectorLoss     lib/core/loss.py:62-88
  EdgeLengthLoss       lib/core/loss.py:91-114
  get_loss             lib/core/loss.py:117-120
  train-step epilogue  lib/core/base.py:130-143  (perm-reverse gather, J-regression, the five weighted losses)
  test-step epilogue   lib/core/base.py:200-204  (gather, x1000, J-regression)

Parity pinning: tests/golden/loss_{mano,coco}.npz hold inputs, the five loss values and d(loss)/d(cam_mesh)
produced by the REAL lib/core/loss.py classes (imported through oracle/ref_loader.py by
tests/golden/make_golden.py); tests/test_oracle.py checks this restatement against them.
"""
import numpy as np
import torch
import torch.nn.functional as F


def coord_loss(pred, target, target_valid=None, has_valid=True):
    """loss.py:17-23: nn.L1Loss(reduction='mean') of the masked tensors."""
    if has_valid:
        pred, target = pred * target_valid, target * target_valid                     # :19
    return F.l1_loss(pred, target, reduction="mean")                                  # :21


def normal_vector_loss(coord_out, coord_gt, face):
    """loss.py:67-88."""
    face = torch.as_tensor(np.asarray(face), dtype=torch.long, device=coord_out.device)   # :68
    v1_out = F.normalize(coord_out[:, face[:, 1], :] - coord_out[:, face[:, 0], :], p=2, dim=2)   # :70-71
    v2_out = F.normalize(coord_out[:, face[:, 2], :] - coord_out[:, face[:, 0], :], p=2, dim=2)   # :72-73
    v3_out = F.normalize(coord_out[:, face[:, 2], :] - coord_out[:, face[:, 1], :], p=2, dim=2)   # :74-75
    v1_gt = F.normalize(coord_gt[:, face[:, 1], :] - coord_gt[:, face[:, 0], :], p=2, dim=2)      # :77-78
    v2_gt = F.normalize(coord_gt[:, face[:, 2], :] - coord_gt[:, face[:, 0], :], p=2, dim=2)      # :79-80
    normal_gt = F.normalize(torch.cross(v1_gt, v2_gt, dim=2), p=2, dim=2)                          # :81-82
    cos1 = torch.abs(torch.sum(v1_out * normal_gt, 2, keepdim=True))                               # :84
    cos2 = torch.abs(torch.sum(v2_out * normal_gt, 2, keepdim=True))
    cos3 = torch.abs(torch.sum(v3_out * normal_gt, 2, keepdim=True))
    return torch.cat((cos1, cos2, cos3), 1).mean()                                                 # :87-88


def edge_length_loss(coord_out, coord_gt, face):
    """loss.py:96-114."""
    face = torch.as_tensor(np.asarray(face), dtype=torch.long, device=coord_out.device)   # :97

    def d(x, a, b):
        return torch.sqrt(torch.sum((x[:, face[:, a], :] - x[:, face[:, b], :]) ** 2, 2, keepdim=True))
    d1_out, d2_out, d3_out = d(coord_out, 0, 1), d(coord_out, 0, 2), d(coord_out, 1, 2)   # :99-104
    d1_gt, d2_gt, d3_gt = d(coord_gt, 0, 1), d(coord_gt, 0, 2), d(coord_gt, 1, 2)         # :106-108
    diff = torch.cat((torch.abs(d1_out - d1_gt), torch.abs(d2_out - d2_gt), torch.abs(d3_out - d3_gt)), 1)
    return diff.mean()                                                                     # :110-114


def train_losses(cam_mesh, lift_pose, perm_reverse, nv, face, J_regressor, gt_mesh, gt_reg3dpose, gt_lift3dpose,
                 val_mesh, val_reg3dpose, val_lift3dpose, normal_weight=1e-1, edge_weight=20.0, joint_weight=1e-3,
                 with_edge=True):
    """lib/core/base.py:130-143 (loss weights: lib/core/config.py:53-55).  Returns (total, [l1, l2, l3, l4, l5])
    with l3 = 0 when the edge loss is not yet switched on (epoch <= edge_loss_start)."""
    perm = torch.as_tensor(np.asarray(perm_reverse)[:nv], dtype=torch.long, device=cam_mesh.device)
    pred_mesh = cam_mesh[:, perm, :]                                                   # :130
    pred_pose = torch.matmul(J_regressor[None, :, :], pred_mesh * 1000)                # :131
    l1 = coord_loss(pred_mesh, gt_mesh, val_mesh)                                      # :134
    l2 = normal_weight * normal_vector_loss(pred_mesh, gt_mesh, face)                  # :135
    l4 = joint_weight * coord_loss(pred_pose, gt_reg3dpose, val_reg3dpose)             # :136
    l5 = joint_weight * coord_loss(lift_pose, gt_lift3dpose, val_lift3dpose)           # :137
    l3 = torch.zeros((), dtype=cam_mesh.dtype)
    loss = l1 + l2 + l4 + l5                                                           # :139
    if with_edge:                                                                      # :141-143
        l3 = edge_weight * edge_length_loss(pred_mesh, gt_mesh, face)
        loss = loss + l3
    return loss, [l1, l2, l3, l4, l5]


def test_epilogue(cam_mesh, perm_reverse, nv, J_regressor):
    """lib/core/base.py:200-204 (also demo/run.py:170-171): mesh in SMPL/MANO vertex order in mm + regressed joints."""
    perm = torch.as_tensor(np.asarray(perm_reverse)[:nv], dtype=torch.long, device=cam_mesh.device)
    pred_mesh = cam_mesh[:, perm, :] * 1000                                            # :201-202
    pred_pose = torch.matmul(J_regressor[None, :, :], pred_mesh)                       # :204
    return pred_mesh, pred_pose
