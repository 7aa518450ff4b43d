"""Throughput and error of p2m_gemm_planes in its arithmetics (native f32 MFMA / 3 bf16 slices / 2 fp16 slices) on the layer shapes
of the SMPL network at batch 256.   python tools/probes/gemm_probe.py [B]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch  # noqa: E402

from pose2mesh_release_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(96, 256, 256), (184, 256, 256), (368, 256, 128), (736, 128, 128), (1472, 128, 64), (2944, 64, 64),
          (5888, 64, 64), (11776, 64, 32)]           # (V, Ka, N)
if os.environ.get("PROBE_SHAPE"):                     # e.g. PROBE_SHAPE=736,128,128 PROBE_MODES=f16x2 for a PMC pass
    SHAPES = [tuple(int(x) for x in os.environ["PROBE_SHAPE"].split(","))]
MODES = os.environ.get("PROBE_MODES", "f32,bf16x3,f16x2").split(",")
PLANES = int(os.environ.get("PROBE_PLANES", "3"))      # 1..3 A planes: time vs K gives the per-tile overhead


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


for V, Ka, N in SHAPES:
    M = B * V
    g = torch.Generator(device="cuda").manual_seed(V)
    A = [torch.randn(M, Ka, device="cuda", generator=g) for _ in range(PLANES)]
    W = torch.randn(PLANES * Ka, N, device="cuda", generator=g) / (PLANES * Ka) ** 0.5
    bias = torch.randn(N, device="cuda", generator=g)
    rows = min(M, 4096)
    ref = torch.cat([a[:rows] for a in A], 1).double() @ W.double() + bias.double()
    line = f"V={V:6d} M={M:8d} K={PLANES}x{Ka:3d} N={N:3d}"
    for mode in MODES:
        ops.GEMM_ARITH = mode
        Bx = ops.weight_split(W)

        def run():
            ops.gemm_planes(A, Ka, 0, W, bias, M, N, 1, True)

        (C,), _ = ops.gemm_planes(A, Ka, 0, W, bias, M, N, 1, True)
        err = (C[:rows].double() - ref).abs().max().item()
        ms = bench(run)
        line += f" | {mode}: {ms:7.3f} ms {2.0 * M * PLANES * Ka * N / ms / 1e9:6.1f} TF err {err:.2e}"
    print(line, flush=True)
