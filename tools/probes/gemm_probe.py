import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch
from pose2mesh_release_amd import ops
B, V, F = 256, 11776, 128
M = B * V
A = [torch.randn(M, F, device="cuda") for _ in range(3)]
W = torch.randn(3 * F, 128, device="cuda")
bias = torch.randn(128, device="cuda")
for _ in range(3):
    ops.gemm_planes(A, F, 0, W, bias, M, 128, 1, True)
torch.cuda.synchronize()
