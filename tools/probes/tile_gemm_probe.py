"""Real-row convs of the split SMPL-like levels at B=256: basis kernel + plane contraction (round 2) against the
basis-inside-the-contraction kernel (p2m_cheb_tile_gemm), forward form (no planes out) and backward form (planes out),
plain / un-pooled / paired.   python tools/probes/tile_gemm_probe.py [all|finest] [B]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch  # noqa: E402

from pose2mesh_release_amd import ops, synth  # noqa: E402

ops.TILE_GEMM = True                      # the probe measures the opt-in kernel (P2M_TILE_GEMM=1)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
_, gL, _, J = synth.make_graphs("human36")
# (level, Ka, N, plan): the real-row launches of one train step of the SMPL network (forward and backward shapes)
CASES = [(0, 128, 128, 0), (0, 128, 128, 1), (0, 128, 64, 0), (0, 64, 128, 0), (0, 128, 128, 2),
         (1, 128, 128, 0), (1, 128, 128, 1), (1, 128, 128, 2),
         (2, 256, 128, 1), (2, 128, 128, 0), (2, 128, 256, 2),
         (3, 256, 256, 1), (3, 256, 256, 0), (3, 256, 256, 2), (4, 256, 256, 1), (4, 256, 256, 0)]
if which == "finest":
    CASES = CASES[:2]
if os.environ.get("PROBE_CASE"):          # e.g. PROBE_CASE=0,128,128,0 for a PMC pass
    CASES = [tuple(int(v) for v in os.environ["PROBE_CASE"].split(","))]
ONLY_TILE = os.environ.get("PROBE_ONLY_TILE", "0") == "1"


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
tot = [0.0, 0.0, 0.0, 0.0, 0.0]
for lvl, Ka, N, plan in CASES:
    g = graphs.setdefault(lvl, ops.DeviceGraph(gL[lvl], "cuda:0"))
    if not g.split or not ops.tile_gemm_ok(g, plan, Ka, N):
        print(f"V={g.V} Ka={Ka} N={N} plan={plan}: not supported", flush=True)
        continue
    shift = 1 if plan == 1 else 0
    nset = g.n_pair_real if plan == 2 else g.n_real
    X = torch.randn(B * (g.V >> shift), Ka, device="cuda")
    W = torch.randn(3 * Ka, N, device="cuda") / (3 * Ka) ** 0.5
    Bx = ops.weight_split(W)
    crow = g.V // 2 if plan == 2 else g.V
    C = torch.empty(B * crow, N, device="cuda")
    A0 = torch.randn(B * (g.V // 2), Ka, device="cuda") if plan == 2 else X
    if plan == 2:
        def two():
            P1, P2 = ops.cheb_basis_pair(g, X, B, Ka)
            ops.gemm_planes_rows(g, 3, B, [A0, P1, P2], Ka, 0, True, W, None, None, C, N, False, Bx=Bx,
                                 amax=ops.amax_of(X), amax_bits=g.plane_bits + 1)
    else:
        def two():
            T1, T2 = ops.cheb_basis_fwd_real(g, X, B, Ka, shift)
            ops.gemm_planes_rows(g, 1, B, [X, T1, T2], Ka, shift, True, W, None, None, C, N, True, Bx=Bx,
                                 amax=ops.amax_of(X), amax_bits=g.plane_bits)
    ms2 = 0.0 if ONLY_TILE else bench(two)
    if plan != 2 and not ONLY_TILE:       # the contraction alone (planes already in HBM) and the weight gradient of the same conv
        T1, T2 = ops.cheb_basis_fwd_real(g, X, B, Ka, shift)
        ms_g = bench(lambda: ops.gemm_planes_rows(g, 1, B, [X, T1, T2], Ka, shift, True, W, None, None, C, N, True, Bx=Bx,
                                                  amax=ops.amax_of(X), amax_bits=g.plane_bits))
        Gy = torch.randn(B * g.V, N, device="cuda")
        E1, E2 = ops.cheb_basis_fwd_real(g, Gy, B, N, 0)
        ms_t = bench(lambda: ops.gemm_tn_rows(g, 1, B, X, Ka, shift, [Gy, E1, E2], N, True, a_amax=ops.amax_of(X),
                                              g_amax=ops.amax_of(Gy), g_bits=g.plane_bits))
        tot[3] += ms_g
        tot[4] += ms_t
        print(f"   contraction alone {ms_g:7.3f} ms ({2.0 * B * nset * 3 * Ka * N / ms_g / 1e9:6.1f} TF) | weight gradient "
              f"{ms_t:7.3f} ms ({2.0 * B * nset * 3 * Ka * N / ms_t / 1e9:6.1f} TF)", flush=True)
        del T1, T2, Gy, E1, E2
    xa = ops.amax_of(X)          # (f16x2) the operand's amax word, computed once outside the timed calls
    ms_f = bench(lambda: ops.cheb_tile_gemm(g, plan, X, A0, Ka, Bx, None, None, C, N, B, stats=(plan != 2), amax=xa))
    ms_b = bench(lambda: ops.cheb_tile_gemm(g, plan, X, A0, Ka, Bx, None, None, C, N, B, want_planes=True, amax=xa))
    fl = 2.0 * B * nset * 3 * Ka * N
    tot[0] += ms2
    tot[1] += ms_f
    tot[2] += ms_b
    print(f"V={g.V:6d} rows={nset:5d} Ka={Ka:3d} N={N:3d} plan={plan}: basis+gemm {ms2:7.3f} ms | tile_gemm {ms_f:7.3f} ms "
          f"({fl / ms_f / 1e9:6.1f} TF) | with planes out {ms_b:7.3f} ms ({fl / ms_b / 1e9:6.1f} TF)", flush=True)
    del X, C, A0
print(f"TOTAL basis+gemm {tot[0]:.3f} ms | tile_gemm {tot[1]:.3f} ms | with planes {tot[2]:.3f} ms | contraction alone "
      f"{tot[3]:.3f} ms | weight gradient {tot[4]:.3f} ms   [P2M_GEMM_ARITH={ops.GEMM_ARITH}]")
