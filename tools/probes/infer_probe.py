"""BASELINE.json configs[1]: batch-64 synthetic Human3.6M 2D poses, full coarse-to-fine SMPL GCN, forward only (eval).
python tools/probes/infer_probe.py [B]"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch  # noqa: E402

from pose2mesh_release_amd import pose2mesh_net, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
faces, graph_L, perm_rev, J = synth.make_graphs("human36")
torch.manual_seed(123)
model = pose2mesh_net.get_model(J, graph_L).cuda().eval()
x = synth.pose2d_batch(B, J, 123).cuda()
with torch.no_grad():
    for _ in range(5):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        model(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
print(f"eval forward B={B} J={J}: {dt * 1e3:.2f} ms/batch, {B / dt:.0f} meshes/s")
