"""PoseNet -- the 2D->3D pose lifter in front of MeshNet (reference: lib/models/posenet.py).

A 4096-wide residual MLP.  Module/parameter names match the reference so `pose_lifter.*` checkpoints load unchanged, and
`pretrained=True` loads `cfg.MODEL.posenet_path` exactly as lib/models/posenet.py:74-75,89-92 does.
"""
import sys

import torch
import torch.nn as nn


def _reference_cfg():
    """The reference's global config (lib/core/config.py) when its scripts have loaded it, else None."""
    mod = sys.modules.get("core.config")
    return getattr(mod, "cfg", None) if mod is not None else None


def cfg_posenet_pretrained():
    """cfg.MODEL.posenet_pretrained (lib/models/pose2mesh_net.py:13 reads it at construction); False outside the
    reference's scripts (config.py:56 default)."""
    cfg = _reference_cfg()
    try:
        return bool(cfg.MODEL.posenet_pretrained) if cfg is not None else False
    except AttributeError:
        return False


def load_checkpoint(load_dir, epoch=0, pick_best=False):
    """lib/funcs_utils.py:122-128: the reference's own loader when that module is imported (so a patched loader is
    honoured), else the same behaviour restated: print, torch.load, ValueError("No checkpoint exists!") on any failure.
    map_location is 'cuda' as in the reference when a GPU is there (CPU-only hosts: 'cpu', where the reference fails)."""
    fu = sys.modules.get("funcs_utils")
    if fu is not None and hasattr(fu, "load_checkpoint"):
        return fu.load_checkpoint(load_dir=load_dir, epoch=epoch, pick_best=pick_best)
    try:
        print(f"Fetch model weight from {load_dir}")
        return torch.load(load_dir, map_location="cuda" if torch.cuda.is_available() else "cpu")
    except Exception as e:
        raise ValueError("No checkpoint exists!\n", e)


class Linear(nn.Module):
    """Pre-activation residual block: BN-ReLU-Dropout-Linear twice (posenet.py:11-38)."""

    def __init__(self, linear_size, p_dropout=0.5):
        super().__init__()
        self.l_size = linear_size
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(p_dropout)
        self.w1 = nn.Linear(linear_size, linear_size)
        self.batch_norm1 = nn.BatchNorm1d(linear_size)
        self.w2 = nn.Linear(linear_size, linear_size)
        self.batch_norm2 = nn.BatchNorm1d(linear_size)

    def forward(self, x):
        y = self.w1(self.dropout(self.relu(self.batch_norm1(x))))
        y = self.w2(self.dropout(self.relu(self.batch_norm2(y))))
        return x + y


class LinearModel(nn.Module):
    """posenet.py:41-92.  As in the reference, the outer batch_norm1/relu/dropout are created (their
    parameters are part of the state dict) but never applied in forward (posenet.py:77-87)."""

    def __init__(self, num_joint, linear_size=4096, num_stage=2, p_dropout=0.5, pretrained=False, posenet_path=None):
        super().__init__()
        self.linear_size, self.p_dropout, self.num_stage = linear_size, p_dropout, num_stage
        self.input_size, self.output_size = num_joint * 2, num_joint * 3
        self.w1 = nn.Linear(self.input_size, linear_size)
        self.batch_norm1 = nn.BatchNorm1d(linear_size)
        self.linear_stages = nn.ModuleList([Linear(linear_size, p_dropout) for _ in range(num_stage)])
        self.w2 = nn.Linear(linear_size, self.output_size)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(p_dropout)
        if pretrained:
            self._load_pretrained_model(posenet_path)

    def forward(self, x):
        y = self.w1(x)
        for stage in self.linear_stages:
            y = stage(y)
        return self.w2(y)

    def _load_pretrained_model(self, posenet_path=None):
        """lib/models/posenet.py:89-92: checkpoint['model_state_dict'] of cfg.MODEL.posenet_path (strict load)."""
        print("Loading pretrained posenet...")
        if posenet_path is None:
            cfg = _reference_cfg()
            if cfg is None:
                raise ValueError("pretrained=True needs cfg.MODEL.posenet_path (the reference's core.config is not "
                                 "loaded) or an explicit posenet_path=")
            posenet_path = cfg.MODEL.posenet_path
        checkpoint = load_checkpoint(load_dir=posenet_path, pick_best=True)
        self.load_state_dict(checkpoint["model_state_dict"])


def get_model(num_joint, hid_dim, num_layer, p_dropout, pretrained=False, posenet_path=None):
    """lib/models/posenet.py:95-98 (+ an explicit path for callers without the reference's cfg)."""
    return LinearModel(num_joint, hid_dim, num_layer, p_dropout, pretrained, posenet_path)
