"""PoseNet -- the 2D->3D pose lifter in front of MeshNet (reference: lib/models/posenet.py).

A 4096-wide residual MLP: dense GEMMs that PyTorch-ROCm already sends to hipBLASLt; it is kept as
stock torch modules on purpose (SURVEY.md section 8f ranks it "next", not on the graph path).
Module/parameter names match the reference so `pose_lifter.*` checkpoints load unchanged.
"""
import torch.nn as nn


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

    def __init__(self, num_joint, linear_size=4096, num_stage=2, p_dropout=0.5, pretrained=False):
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
            raise NotImplementedError("load PoseNet weights with load_state_dict(); the reference's "
                                      "cfg.MODEL.posenet_path lookup (posenet.py:89-92) is not replicated")

    def forward(self, x):
        y = self.w1(x)
        for stage in self.linear_stages:
            y = stage(y)
        return self.w2(y)


def get_model(num_joint, hid_dim, num_layer, p_dropout, pretrained=False):
    return LinearModel(num_joint, hid_dim, num_layer, p_dropout, pretrained)
