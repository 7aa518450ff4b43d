import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch
from pose2mesh_release_amd import ops, synth
_, gL, _, J = synth.make_graphs("human36")
g = ops.DeviceGraph(gL[0], "cuda:0")
B, V, F = 256, 11776, 128
X = torch.randn(B * V, F, device="cuda")
for _ in range(3):
    T1, T2 = ops.cheb_basis_fwd(g, X, B, F, 0)
d = [torch.randn(B * V, F, device="cuda") for _ in range(2)]
for _ in range(3):
    dX = ops.cheb_basis_bwd(g, X, d[0], d[1], None, B, F, 0)
torch.cuda.synchronize()
