import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch
from pose2mesh_release_amd import ops
def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = []
for rows, F, res in ((256 * 7743, 128, False), (256 * 7743, 128, True), (256 * 4100, 128, False), (256 * 2200, 256, True)):
    y = torch.randn(rows, F, device="cuda"); r = torch.randn(rows, F, device="cuda") if res else None
    co = torch.randn(4, F, device="cuda")
    ms = bench(lambda: ops.bn_act_fwd(y, co, True, r, F, 0, rows, F))
    gb = rows * F * 4 * (3 if res else 2) / 1e9
    out.append(f"{ms*1e3:7.1f} us {gb/ms:6.2f} TB/s")
print(os.environ.get("P2M_HIP_LIB", "default")[-24:], " | ".join(out))
