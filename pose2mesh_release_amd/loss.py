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
