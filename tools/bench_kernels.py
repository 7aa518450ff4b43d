#!/usr/bin/env python3
"""Micro-benchmarks of the individual C-ABI kernels at the BASELINE shapes (B=256, SMPL-like hull levels).
usage: python tools/bench_kernels.py [basis] [gemm] [bn] [tn] ...   (default: all)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "oracle")]
import torch
from pose2mesh_release_amd import ops, synth

B = int(os.environ.get("B", "256"))
which = set(sys.argv[1:]) or {"basis", "gemm", "tn", "bn", "unpack"}


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


_, gL, _, J = synth.make_graphs("human36")
graphs = {L.shape[0]: ops.DeviceGraph(L, "cuda:0") for L in gL[:-2]}
dev = "cuda:0"
if "basis" in which:
    for V, F, sh in [(11776, 128, 1), (11776, 128, 0), (11776, 64, 0), (5888, 128, 1), (1472, 256, 1), (736, 256, 0), (184, 256, 1)]:
        g = graphs[V]
        X = torch.randn(B * (V >> sh), F, device=dev)
        ms = timeit(lambda: ops.cheb_basis_fwd(g, X, B, F, sh))
        by = 4.0 * B * V * F * (2 + 1.0 / (1 << sh))
        d = [torch.randn(B * V, F, device=dev) for _ in range(3)]
        ms2 = timeit(lambda: ops.cheb_basis_bwd(g, d[0], d[1], d[2], None, B, F, sh))
        by2 = 4.0 * B * V * F * (3 + 1.0 / (1 << sh))
        print(f"basis V={V:6d} F={F:3d} sh={sh}: fwd {ms:7.3f} ms {by/ms/1e6:7.1f} GB/s | bwd {ms2:7.3f} ms {by2/ms2/1e6:7.1f} GB/s", flush=True)
        del X, d
if "gemm" in which:
    for V, Fin, Fout in [(11776, 128, 128), (11776, 128, 64), (2944, 256, 128), (1472, 256, 256), (184, 256, 256), (92, 64, 128)]:
        M = B * V
        A = [torch.randn(M, Fin, device=dev) for _ in range(3)]
        W = torch.randn(3 * Fin, Fout, device=dev)
        bias = torch.randn(Fout, device=dev)
        ms = timeit(lambda: ops.gemm_planes(A, Fin, 0, W, bias, M, Fout, 1, True))
        fl = 2.0 * M * 3 * Fin * Fout
        G = torch.randn(M, Fout, device=dev)
        W2 = torch.randn(Fout, 3 * Fin, device=dev)
        ms2 = timeit(lambda: ops.gemm_planes([G], Fout, 0, W2, None, M, 3 * Fin, 3, False))
        print(f"gemm V={V:6d} {Fin}->{Fout}: fwd {ms:7.3f} ms {fl/ms/1e9:6.1f} TF | dZ {ms2:7.3f} ms {fl/ms2/1e9:6.1f} TF", flush=True)
        if "tn" in which:
            ms3 = timeit(lambda: ops.gemm_tn(A, Fin, 0, G, M, Fout))
            print(f"   tn: {ms3:7.3f} ms {fl/ms3/1e9:6.1f} TF", flush=True)
        del A, G
if "bn" in which:
    for V, F in [(11776, 128), (1472, 256)]:
        M = B * V
        y = torch.randn(M, F, device=dev); gx = torch.randn(M, F, device=dev)
        gamma = torch.rand(F, device=dev) + 0.5
        co = torch.randn(4, F, device=dev); co[1].abs_()
        ms = timeit(lambda: ops.bn_relu_bwd(gx, y, co, gamma, True, True, M, F))
        ms2 = timeit(lambda: ops.bn_act_fwd(y, co, True, gx, F, 0, M, F))
        print(f"bn V={V} F={F}: bwd(reduce+fin+apply) {ms:7.3f} ms ({5*4.0*M*F/ms/1e6:7.1f} GB/s) | act_fwd+res {ms2:7.3f} ms ({3*4.0*M*F/ms2/1e6:7.1f} GB/s)", flush=True)
if "rcm" in which:
    import numpy as np, scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    for V, F in [(11776, 128), (5888, 128), (1472, 256)]:
        L = [l for l in gL if l.shape[0] == V][0].tocsr()
        perm = reverse_cuthill_mckee(sp.csr_matrix(L != 0), symmetric_mode=True)
        Lp = L[perm][:, perm]
        coo = Lp.tocoo(); print("V", V, "bandwidth tree", np.abs(L.tocoo().row - L.tocoo().col).max(), "rcm", np.abs(coo.row - coo.col).max(), "mean", np.abs(L.tocoo().row - L.tocoo().col).mean(), np.abs(coo.row - coo.col).mean())
        for name, mat in (("tree", L), ("rcm", Lp)):
            g = ops.DeviceGraph(mat, dev)
            X = torch.randn(B * V, F, device=dev)
            ms = timeit(lambda: ops.cheb_basis_fwd(g, X, B, F, 0))
            by = 4.0 * B * V * F * 3
            print(f"  {name}: basis fwd V={V} F={F}: {ms:7.3f} ms {by/ms/1e6:7.1f} GB/s", flush=True)
            del X
