"""Throughput of the weight-gradient contraction p2m_gemm_tn (flat mode) in both arithmetics on the layer shapes of
the SMPL network at batch 256: P = X^T [g | Lg | L2g].   python tools/probes/gemm_tn_probe.py [B]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch  # noqa: E402

from pose2mesh_release_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(96, 256, 256), (184, 256, 256), (368, 256, 128), (736, 128, 128), (1472, 128, 64), (2944, 64, 64),
          (5888, 64, 64), (11776, 64, 32)]           # (V, Fin, Fout)
MODES = os.environ.get("PROBE_MODES", "f32,bf16x3").split(",")


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


for V, Fin, Fout in SHAPES:
    M = B * V
    g = torch.Generator(device="cuda").manual_seed(V)
    X = torch.randn(M, Fin, device="cuda", generator=g)
    G = [torch.randn(M, Fout, device="cuda", generator=g) for _ in range(3)]
    rows = min(M, 2048)
    line = f"V={V:6d} M={M:8d} K={Fin:3d} N=3x{Fout:3d}"
    for mode in MODES:
        ops.GEMM_ARITH = mode
        P, Pdb, nch = ops.gemm_tn([X], Fin, 0, G, M, 3 * Fout)
        ms = bench(lambda: ops.gemm_tn([X], Fin, 0, G, M, 3 * Fout))
        ms_u = bench(lambda: ops.weight_grad_unpack(P, Pdb, nch, Fout, Fin, 3, layout=1))
        line += f" | {mode}: {ms:7.3f} ms {2.0 * M * Fin * 3 * Fout / ms / 1e9:6.1f} TF (chunks {nch}, unpack {ms_u:.3f} ms)"
    print(line, flush=True)
