"""Basis stage of the finest SMPL-like level (B=256, V=11776, 6890 real rows) for timing / PMC passes:
the full-row kernel, the real-row kernel (P2M_BASIS_TILED=0: k_basis_fwd with ids, 1: k_basis_tile) and, with
PROBE_BWD=1, the retired two-source k_basis_bwd.   python tools/probes/basis_probe.py [F] [shift]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch  # noqa: E402

from pose2mesh_release_amd import ops, synth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
shift = int(sys.argv[2]) if len(sys.argv) > 2 else 0
_, gL, _, J = synth.make_graphs("human36")
g = ops.DeviceGraph(gL[0], "cuda:0")
B, V = 256, g.V
X = torch.randn(B * (V >> shift), F, device="cuda")


def bench(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ms_full = bench(lambda: ops.cheb_basis_fwd(g, X, B, F, shift))
ms_real = bench(lambda: ops.cheb_basis_fwd_real(g, X, B, F, shift))
moved_full = 4.0 * B * V * F * (2 + 1.0 / (1 << shift))
moved_real = 4.0 * B * g.n_real * F * (2 + 1.0 / (1 << shift))
print(f"V={V} real={g.n_real} F={F} shift={shift}: all rows {ms_full:.3f} ms ({moved_full / ms_full / 1e6:.0f} GB/s) | "
      f"real rows {ms_real:.3f} ms ({moved_real / ms_real / 1e6:.0f} GB/s moved, "
      f"{12.0 * B * V * F / ms_real / 1e6:.0f} GB/s by the all-rows formula)")
if os.environ.get("PROBE_BWD") == "1":
    d = [torch.randn(B * V, F, device="cuda") for _ in range(2)]
    Xf = torch.randn(B * V, F, device="cuda")
    print(f"k_basis_bwd: {bench(lambda: ops.cheb_basis_bwd(g, Xf, d[0], d[1], None, B, F, 0)):.3f} ms")
