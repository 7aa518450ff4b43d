"""Time of the plane contraction alone (k_gemm_planes_ws, row-set mode, planes already in HBM) on one conv shape:
the probe of tools/planes_ablate.sh.   python tools/probes/planes_abl_probe.py [level Ka N B]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch  # noqa: E402

from pose2mesh_release_amd import ops, synth  # noqa: E402

lvl, Ka, N, B = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (0, 128, 128, 256)
_, gL, _, J = synth.make_graphs("human36")
g = ops.DeviceGraph(gL[lvl], "cuda:0")
X = torch.randn(B * g.V, Ka, device="cuda")
W = torch.randn(3 * Ka, N, device="cuda") / (3 * Ka) ** 0.5
Bx = ops.weight_split(W)
C = torch.empty(B * g.V, N, device="cuda")
T1, T2 = ops.cheb_basis_fwd_real(g, X, B, Ka, 0)
xa = ops.amax_of(X)


def run(stats):
    return ops.gemm_planes_rows(g, 1, B, [X, T1, T2], Ka, 0, True, W, None, None, C, N, stats, Bx=Bx, amax=xa,
                                amax_bits=g.plane_bits)


for stats in (True, False):
    run(stats)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        run(stats)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 6
    print(f"B={B} V={g.V} rows={g.n_real} Ka={Ka} N={N} stats={int(stats)}: {ms:.3f} ms, "
          f"{2.0 * B * g.n_real * 3 * Ka * N / ms / 1e9:.1f} TF, {B * g.n_real * (3 * Ka + N) * 4 / ms / 1e6:.0f} GB/s algorithmic"
          f"   [{ops.GEMM_ARITH}]", flush=True)
