import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch
from pose2mesh_release_amd import ops, synth
_, gL, _, J = synth.make_graphs("human36")
g = ops.DeviceGraph(gL[0], "cuda:0")
B, V, F = 256, 11776, 128
X = torch.randn(B * V, F, device="cuda")
W = torch.randn(3 * F, 128, device="cuda")
for _ in range(3):
    ops.cheb_gemm_fused(g, X, F, 0, W, None, None, 128, B)
torch.cuda.synchronize()
