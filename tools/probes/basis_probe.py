"""Basis stage of the split SMPL-like levels at B=256 for timing / PMC passes: real-row kernel (k_basis_tile, or
k_basis_fwd with ids under P2M_BASIS_TILED=0) per (level, F, shift).   python tools/probes/basis_probe.py [all|finest] [B]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch  # noqa: E402

from pose2mesh_release_amd import ops, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "all"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
_, gL, _, J = synth.make_graphs("human36")
CASES = [(0, 128, 1), (0, 128, 0), (0, 64, 0), (1, 128, 1), (1, 128, 0), (2, 256, 1), (2, 128, 0), (3, 256, 1), (3, 256, 0),
         (4, 256, 1)]
if which == "finest":
    CASES = CASES[:2]


def bench(fn, n=6):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


graphs = {}
tot_ms, tot_bytes = 0.0, 0.0
for lvl, F, shift in CASES:
    g = graphs.setdefault(lvl, ops.DeviceGraph(gL[lvl], "cuda:0"))
    if not g.split:
        continue
    X = torch.randn(B * (g.V >> shift), F, device="cuda")
    ms = bench(lambda: ops.cheb_basis_fwd_real(g, X, B, F, shift))
    moved = 4.0 * B * g.n_real * F * (2 + 1.0 / (1 << shift))
    tot_ms += ms
    tot_bytes += moved
    print(f"V={g.V:6d} real={g.n_real:5d} F={F:3d} shift={shift}: {ms:7.3f} ms  {moved / ms / 1e6:7.0f} GB/s moved", flush=True)
    del X
    if g.pair and shift == 0:
        # the paired operator (S L g, S L2 g over the coarse vertices with a real child): same source rows, half the planes
        G = torch.randn(B * g.V, F, device="cuda")
        ms = bench(lambda: ops.cheb_basis_pair(g, G, B, F))
        moved = 4.0 * B * F * (g.V + 2.0 * g.n_pair_real)
        print(f"V={g.V:6d} pair={g.n_pair_real:5d} F={F:3d} paired : {ms:7.3f} ms  {moved / ms / 1e6:7.0f} GB/s moved", flush=True)
        del G
print(f"TOTAL {tot_ms:.3f} ms  {tot_bytes / tot_ms / 1e6:.0f} GB/s moved   [TILED={os.environ.get('P2M_BASIS_TILED', '1')}]")
