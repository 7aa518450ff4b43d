"""-m gpu: every C-ABI entry point against a plain fp32/fp64 torch restatement of the same op
(seeded inputs; tolerances are fp32 round-off class, written next to each check)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand_graph(V, seed, fake_frac=0.3):
    """Symmetric rescaled-Laplacian-like matrix with isolated (diagonal-only) fake rows."""
    rng = np.random.default_rng(seed)
    nreal = max(2, int(V * (1 - fake_frac)))
    rows, cols = [], []
    for i in range(nreal):
        for j in rng.choice(nreal, size=min(3, nreal - 1), replace=False):
            if i != j:
                rows += [i, j]
                cols += [j, i]
    A = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(V, V)).tocsr()
    A.data[:] = 1.0
    d = np.asarray(A.sum(axis=0)).ravel() + np.spacing(np.float64(0))
    Dm = sp.diags(1 / np.sqrt(d))
    L = sp.identity(V) - Dm @ A @ Dm
    return (L / 3.0 - sp.identity(V)).tocsr()


def _band_graph(V, seed, fake_frac=0.4):
    """Bounded-degree variant (ring + two chords over the real vertices, then a random renumbering of ALL vertices so the
    isolated fake vertices end up anywhere, as in a coarsening tree): every merged row fits a tile of the LDS basis kernel."""
    nreal = max(8, int(V * (1 - fake_frac)))
    i = np.arange(nreal)
    rows = np.concatenate([i, (i + 1) % nreal, i, (i + 5) % nreal, i, (i + 17) % nreal])
    cols = np.concatenate([(i + 1) % nreal, i, (i + 5) % nreal, i, (i + 17) % nreal, i])
    A = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(V, V)).tocsr()
    A.data[:] = 1.0
    d = np.asarray(A.sum(axis=0)).ravel() + np.spacing(np.float64(0))
    Dm = sp.diags(1 / np.sqrt(d))
    L = (sp.identity(V) - Dm @ A @ Dm) / 3.0 - sp.identity(V)
    perm = np.random.default_rng(seed).permutation(V)
    return L.tocsr()[perm][:, perm].tocsr()


@pytest.fixture(scope="module")
def ops(hip_libs):
    from pose2mesh_release_amd import ops as o
    return o


@pytest.fixture(params=["f32", "bf16x3", "f16x2"])
def arith(request, ops, monkeypatch):
    """The arithmetics of the dense contraction: native f32 MFMA, 3 bf16 slices, 2 scaled fp16 slices (include/p2m.h)."""
    monkeypatch.setattr(ops, "GEMM_ARITH", request.param)
    return request.param


@pytest.mark.parametrize("V,Fdim,shift", [(17, 5, 0), (96, 64, 0), (184, 256, 1), (1472, 128, 1), (736, 32, 0),
                                          (46, 3, 0)])
def test_cheb_basis_fwd_bwd(ops, V, Fdim, shift):
    B = 3
    L = _rand_graph(V, V)
    g = ops.DeviceGraph(L, "cuda:0")
    Ld = torch.from_numpy(L.toarray())
    gen = torch.Generator().manual_seed(V + Fdim)
    Xs = torch.randn(B, V >> shift, Fdim, generator=gen)
    X = Xs.repeat_interleave(1 << shift, dim=1).double()
    T1r = torch.einsum("vw,bwf->bvf", Ld, X)
    T2r = 2 * torch.einsum("vw,bwf->bvf", Ld, T1r) - X            # cheby_graph_conv.py:25,28
    T1, T2 = ops.cheb_basis_fwd(g, Xs.cuda().view(-1, Fdim).contiguous(), B, Fdim, shift)
    assert (T1.cpu().view(B, V, Fdim) - T1r).abs().max() < 2e-6 * max(1, T1r.abs().max())
    assert (T2.cpu().view(B, V, Fdim) - T2r).abs().max() < 4e-6 * max(1, T2r.abs().max())
    # backward: dX = d0 + L d1 + (2LL-I) d2, children summed when shifted
    d = [torch.randn(B, V, Fdim, generator=gen) for _ in range(3)]
    res = torch.randn(B, V, Fdim, generator=gen)
    L2 = 2 * Ld @ Ld - torch.eye(V, dtype=torch.float64)
    full = d[0].double() + res.double() + torch.einsum("vw,bwf->bvf", Ld, d[1].double()) + \
        torch.einsum("vw,bwf->bvf", L2, d[2].double())
    if shift:
        full = full.view(B, V // 2, 2, Fdim).sum(2)
    dX = ops.cheb_basis_bwd(g, *[t.cuda().view(-1, Fdim).contiguous() for t in d],
                            res.cuda().view(-1, Fdim).contiguous(), B, Fdim, shift)
    assert (dX.cpu().view(full.shape) - full).abs().max() < 5e-6 * max(1, full.abs().max())


@pytest.mark.parametrize("M,Ka,N,planes,shift", [(300, 32, 64, 3, 0), (1000, 128, 256, 3, 1), (257, 64, 128, 1, 0),
                                                 (513, 5, 32, 3, 0), (640, 64, 3, 3, 0), (256, 1088, 5888, 1, 0),
                                                 (129, 256, 96, 1, 0)])
def test_gemm_planes_and_stats(ops, arith, M, Ka, N, planes, shift):
    gen = torch.Generator().manual_seed(M + N)
    A = [torch.randn((M + 1) >> shift if (p == 0 and shift) else M, Ka, generator=gen) for p in range(planes)]
    Bm = torch.randn(planes * Ka, N, generator=gen) / np.sqrt(planes * Ka)
    bias = torch.randn(N, generator=gen)
    Afull = [a.repeat_interleave(2, 0)[:M] if (p == 0 and shift) else a for p, a in enumerate(A)]
    ref = torch.cat(Afull, 1).double() @ Bm.double() + bias.double()
    (C,), st = ops.gemm_planes([a.cuda() for a in A], Ka, shift, Bm.cuda(), bias.cuda(), M, N, 1, True)
    torch.cuda.synchronize()
    assert (C.cpu() - ref).abs().max() < 2e-5
    tr = ops.stats_tile_rows()
    nt = (M + tr - 1) // tr
    assert st.shape == (nt, 2, N)
    for t in range(nt):
        blk = ref[t * tr:(t + 1) * tr]
        assert (st[t, 0].cpu() - blk.sum(0)).abs().max() < 1e-3
        assert (st[t, 1].cpu() - ((blk - blk.mean(0)) ** 2).sum(0)).abs().max() < 2e-3


def test_gemm_planes_output_planes(ops, arith):
    M, K, Nc = 500, 128, 64
    gen = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=gen)
    Bm = torch.randn(K, 3 * Nc, generator=gen) / 11
    ref = A.double() @ Bm.double()
    C, _ = ops.gemm_planes([A.cuda()], K, 0, Bm.cuda(), None, M, 3 * Nc, 3, False)
    for q in range(3):
        assert (C[q].cpu() - ref[:, q * Nc:(q + 1) * Nc]).abs().max() < 2e-5


@pytest.mark.parametrize("M,Ka,N,planes,shift", [(3000, 128, 128, 3, 1), (700, 32, 64, 3, 0), (513, 5, 32, 3, 0),
                                                 (2000, 64, 3, 3, 0), (256, 1088, 320, 1, 0), (999, 256, 256, 3, 0)])
def test_gemm_tn_and_unpack(ops, arith, M, Ka, N, planes, shift):
    gen = torch.Generator().manual_seed(M + Ka)
    A = [torch.randn((M + 1) >> shift if (p == 0 and shift) else M, Ka, generator=gen) for p in range(planes)]
    G = torch.randn(M, N, generator=gen)
    Afull = [a.repeat_interleave(2, 0)[:M] if (p == 0 and shift) else a for p, a in enumerate(A)]
    Z = torch.cat(Afull, 1).double()                       # [M, planes*Ka], column = k*Ka + fin
    dWt = Z.t() @ G.double()                               # [planes*Ka, N]
    P, Pdb, nch = ops.gemm_tn([a.cuda() for a in A], Ka, shift, G.cuda(), M, N)
    dW, db = ops.weight_grad_unpack(P, Pdb, nch, N, Ka, planes)
    # nn.Linear layout: dW[fout][fin*K + k]
    ref = dWt.view(planes, Ka, N).permute(2, 1, 0).reshape(N, Ka * planes)
    scale = max(1.0, ref.abs().max().item())
    assert (dW.cpu() - ref).abs().max() < 3e-6 * scale * np.sqrt(M)
    assert (db.cpu() - G.double().sum(0)).abs().max() < 1e-4 * np.sqrt(M)


def test_weight_pack(ops):
    Fout, Fin, K = 64, 32, 3
    W = torch.randn(Fout, Fin * K)
    Wt, W2, W3 = ops.weight_pack(W.cuda(), Fin, K, need_w2=True, need_w3=True)
    ref = W.view(Fout, Fin, K).permute(2, 1, 0).reshape(K * Fin, Fout)
    assert torch.equal(Wt.cpu(), ref)
    assert torch.equal(W2.cpu(), ref.t().contiguous())
    assert torch.equal(W3.cpu(), W.view(Fout, Fin, K).permute(2, 0, 1).reshape(K * Fout, Fin))


@pytest.mark.parametrize("M,Fd,Fres,rshift,training", [(1000, 64, 64, 0, True), (2048, 256, 64, 0, True),
                                                       (640, 128, 256, 1, True), (777, 32, 32, 0, False),
                                                       (5000, 128, 128, 1, True)])
def test_bn_relu_residual_fwd_bwd(ops, M, Fd, Fres, rshift, training):
    """BatchNorm1d (train/eval) + ReLU + feature-axis-resized residual, forward and backward, against
    torch autograd in fp64 (cheby_graph_conv.py:39, meshnet.py:100,109-110)."""
    gen = torch.Generator().manual_seed(M)
    y = torch.randn(M, Fd, generator=gen) * 2 + 0.5
    gamma = torch.rand(Fd, generator=gen) + 0.5
    beta = torch.randn(Fd, generator=gen) * 0.2
    rm, rv = torch.randn(Fd, generator=gen) * 0.1, torch.rand(Fd, generator=gen) + 0.5
    resid = torch.randn((M + 1) >> rshift, Fres, generator=gen)
    gx = torch.randn(M, Fd, generator=gen)
    # --- reference in fp64
    yd = y.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rmd, rvd = rm.double().clone(), rv.double().clone()
    rd = resid.double().requires_grad_(True)
    o = F.relu(F.batch_norm(yd, rmd, rvd, gd, bd, training, 0.1, 1e-5))
    rfull = rd.repeat_interleave(1 << rshift, 0)[:M]
    o = o + F.interpolate(rfull.unsqueeze(0), size=Fd, mode="linear").squeeze(0) if Fres != Fd else o + rfull
    o.backward(gx.double())
    # --- HIP
    yc = y.cuda()
    if training:
        tr = ops.stats_tile_rows()
        nt = (M + tr - 1) // tr
        st = torch.empty(nt, 2, Fd)
        for t in range(nt):
            blk = y[t * tr:(t + 1) * tr].double()
            st[t, 0] = blk.sum(0).float()
            st[t, 1] = ((blk - blk.mean(0)) ** 2).sum(0).float()
        rmc, rvc = rm.cuda(), rv.cuda()
        co = ops.bn_finalize(st.cuda(), M, gamma.cuda(), beta.cuda(), rmc, rvc, 0.1, 1e-5)
        assert (rmc.cpu() - rmd).abs().max() < 1e-6 and (rvc.cpu() - rvd).abs().max() < 1e-5
    else:
        co = ops.bn_eval_coeffs(gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda(), 1e-5)
    x = ops.bn_act_fwd(yc, co, True, resid.cuda(), Fres, rshift, M, Fd)
    assert (x.cpu() - o.detach()).abs().max() < 1e-5
    gy, dgamma, dbeta = ops.bn_relu_bwd(gx.cuda(), yc, co, gamma.cuda(), True, training, M, Fd)
    assert (gy.cpu() - yd.grad).abs().max() < 2e-5 * max(1.0, yd.grad.abs().max().item())
    assert (dgamma.cpu() - gd.grad).abs().max() < 1e-5 * max(1.0, gd.grad.abs().max().item()) * np.sqrt(M)
    assert (dbeta.cpu() - bd.grad).abs().max() < 1e-5 * max(1.0, bd.grad.abs().max().item()) * np.sqrt(M)
    if M % 2 == 0 and Fd in (32, 64, 128, 256):
        # the pair-sum by-products of the apply pass: same gy bit for bit, pair_gx = S gx, pair_gy = S gy
        gy2, _, _, pgx, pgy = ops.bn_relu_bwd(gx.cuda(), yc, co, gamma.cuda(), True, training, M, Fd, pair_in=True,
                                              pair_out=True)
        assert torch.equal(gy2, gy)
        assert torch.equal(pgx, ops.pair_sum(gx.cuda(), M // 2, Fd)) and torch.equal(pgy, ops.pair_sum(gy, M // 2, Fd))
    # residual transpose
    G = gx.cuda()
    Mr = M >> rshift if rshift else M
    if rshift and M % 2 == 0:
        Gs = ops.pair_sum(G, M // 2, Fd)
    else:
        Gs = G
    if not rshift or M % 2 == 0:
        dst = torch.zeros(Mr, Fres, device="cuda")
        if Fres == Fd:
            dst += Gs
        else:
            ops.lerp_bwd_add(Gs, dst, Mr, Fd, Fres)
        assert (dst.cpu() - rd.grad[:Mr]).abs().max() < 1e-5


def test_gemm_tn_with_planes_and_layout1(ops, arith):
    """dW = X^T [g | E1 | E2] unpacked to nn.Linear layout (the fused backward's weight gradient)."""
    M, Fin, Fout = 1500, 64, 128
    gen = torch.Generator().manual_seed(5)
    X = torch.randn(M, Fin, generator=gen)
    G = [torch.randn(M, Fout, generator=gen) for _ in range(3)]
    P, Pdb, nch = ops.gemm_tn([X.cuda()], Fin, 0, [t.cuda() for t in G], M, 3 * Fout)
    dW, db = ops.weight_grad_unpack(P, Pdb, nch, Fout, Fin, 3, layout=1)
    ref = torch.stack([G[k].double().t() @ X.double() for k in range(3)], dim=2)      # [Fout][Fin][k]
    assert (dW.cpu() - ref.reshape(Fout, Fin * 3)).abs().max() < 2e-4
    assert (db.cpu() - G[0].double().sum(0)).abs().max() < 1e-3


def test_gemm_planes_addend_and_pair_out(ops, arith):
    """Epilogue extras used by the forward-form backward: + addend, and the un-pool pair-sum output."""
    M, K, N = 1000, 64, 128
    gen = torch.Generator().manual_seed(3)
    A = [torch.randn(M, K, generator=gen) for _ in range(3)]
    Bm = torch.randn(3 * K, N, generator=gen) / 14
    add = torch.randn(M, N, generator=gen)
    ref = torch.cat(A, 1).double() @ Bm.double() + add.double()
    (C,), _ = ops.gemm_planes([a.cuda() for a in A], K, 0, Bm.cuda(), None, M, N, 1, False, addend=add.cuda())
    assert (C.cpu() - ref).abs().max() < 2e-5
    (Cp,), _ = ops.gemm_planes([a.cuda() for a in A], K, 0, Bm.cuda(), None, M, N, 1, False, addend=add.cuda(),
                               pair_out=True)
    assert Cp.shape == (M // 2, N)
    assert (Cp.cpu() - ref.view(M // 2, 2, N).sum(1)).abs().max() < 4e-5


def test_fake_vertex_split_matches_unsplit(ops, arith):
    """Row-set launches (real: 3-plane K, fake: K=Fin with W0 + a W1 + b W2) reproduce the plain conv, its
    BatchNorm statistics, and the weight gradient (fake vertices are isolated)."""
    V, Fin, Fout, B = 736, 64, 128, 3
    L = _rand_graph(V, 11, fake_frac=0.4)
    g = ops.DeviceGraph(L, "cuda:0")
    assert g.n_real + g.n_fake == V and g.n_fake >= int(0.39 * V)
    gen = torch.Generator().manual_seed(2)
    M = B * V
    X = torch.randn(M, Fin, generator=gen).cuda()
    Wt = (torch.randn(3 * Fin, Fout, generator=gen) / 14).cuda()
    bias = torch.randn(Fout, generator=gen).cuda()
    # reference: full basis + plain GEMM
    T1, T2 = ops.cheb_basis_fwd(g, X, B, Fin, 0)
    (yref,), stref = ops.gemm_planes([X, T1, T2], Fin, 0, Wt, bias, M, Fout, 1, True)
    # split
    T1c, T2c = ops.cheb_basis_fwd_real(g, X, B, Fin, 0)
    y = torch.full((M, Fout), float("nan"), device="cuda")
    st1 = ops.gemm_planes_rows(g, 1, B, [X, T1c, T2c], Fin, 0, True, Wt, bias, None, y, Fout, True)
    We = ops.weight_eff(Wt, Fin, Fout, g.fake_a, g.fake_b)
    st2 = ops.gemm_planes_rows(g, 2, B, [X], Fin, 0, False, We, bias, None, y, Fout, True)
    assert torch.isfinite(y).all()
    assert (y - yref).abs().max() < 2e-5
    gamma, beta = torch.ones(Fout).cuda(), torch.zeros(Fout).cuda()
    co_ref = ops.bn_finalize(stref, M, gamma, beta, None, None, 0.1, 1e-5)
    co = ops.bn_finalize_rows(g, B, st1, st2, gamma, beta, None, None, 0.1, 1e-5)
    assert (co - co_ref).abs().max() < 1e-5
    # weight gradient: X^T [gy | L gy | L2 gy]
    gy = torch.randn(M, Fout, generator=gen).cuda()
    E1, E2 = ops.cheb_basis_fwd(g, gy, B, Fout, 0)
    P, Pdb, nch = ops.gemm_tn([X], Fin, 0, [gy, E1, E2], M, 3 * Fout)
    dWref, dbref = ops.weight_grad_unpack(P, Pdb, nch, Fout, Fin, 3, layout=1)
    E1c, E2c = ops.cheb_basis_fwd_real(g, gy, B, Fout, 0)
    P1, Pb1, n1 = ops.gemm_tn_rows(g, 1, B, X, Fin, 0, [gy, E1c, E2c], Fout, True)
    P2, Pb2, n2 = ops.gemm_tn_rows(g, 2, B, X, Fin, 0, [gy], Fout, False)
    dW, db = ops.weight_grad_unpack2(P1, Pb1, n1, P2, Pb2, n2, g.fake_a, g.fake_b, Fout, Fin)
    assert (dW - dWref).abs().max() < 1e-4 * max(1.0, dWref.abs().max().item())
    assert (db - dbref).abs().max() < 1e-3


@pytest.mark.parametrize("arith", ["bf16x3", "f16x2"])
def test_weight_gradient_chunks_of_whole_samples(ops, monkeypatch, arith):
    """p2m_gemm_tn_rows with splits = -S (S whole samples per chunk, the last chunk short): fewer partial buffers, the same
    gradient - against the one-chunk-per-sample form (sums in a different order: fp32 round-off) and against float64."""
    monkeypatch.setattr(ops, "GEMM_ARITH", arith)
    V, Fin, Fout, B = 736, 128, 64, 7
    L = _rand_graph(V, 11, fake_frac=0.4)
    g = ops.DeviceGraph(L, "cuda:0")
    gen = torch.Generator().manual_seed(9)
    X = torch.randn(B * V, Fin, generator=gen).cuda()
    gy = torch.randn(B * V, Fout, generator=gen).cuda()
    E1c, E2c = ops.cheb_basis_fwd_real(g, gy, B, Fout, 0)
    P1, Pb1, n1 = ops.gemm_tn_rows(g, 1, B, X, Fin, 0, [gy, E1c, E2c], Fout, True, samples_per_chunk=1)
    assert n1 % B == 0                      # (few samples: every sample's rows in n1 / B slices)
    P1 = P1.view(B, n1 // B, Fin, 3 * Fout).sum(1)
    Pb1 = Pb1.view(B, n1 // B, 3 * Fout).sum(1)
    for S in (2, 3, 7, 9):
        P, Pb, n = ops.gemm_tn_rows(g, 1, B, X, Fin, 0, [gy, E1c, E2c], Fout, True, samples_per_chunk=S)
        assert n == -(-B // S) and P.shape == (n, Fin, 3 * Fout)
        assert torch.isfinite(P).all()
        scale = max(1.0, P1.sum(0).abs().max().item())
        assert (P.sum(0) - P1.sum(0)).abs().max() < 2e-5 * scale, S
        assert (Pb.sum(0) - Pb1.sum(0)).abs().max() < 1e-4 * max(1.0, Pb1.sum(0).abs().max().item())
        # chunk c holds samples c*S ..: its partial is the sum of those samples' one-chunk partials
        for c in range(n):
            ref = P1[c * S:(c + 1) * S].sum(0)
            assert (P[c] - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item()), (S, c)
    # the fake rows' gradient (one plane), several samples per chunk
    Q1, Qb1, m1 = ops.gemm_tn_rows(g, 2, B, X, Fin, 0, [gy], Fout, False, samples_per_chunk=1)
    Q, Qb, m = ops.gemm_tn_rows(g, 2, B, X, Fin, 0, [gy], Fout, False, samples_per_chunk=4)
    assert m == 2 and (Q.sum(0) - Q1.sum(0)).abs().max() < 2e-5 * max(1.0, Q1.sum(0).abs().max().item())


def _bf16_bits_to_float(t):
    return (t.to(torch.int32) << 16).view(torch.float32)


def test_weight_split_is_exact(ops, monkeypatch):
    """p2m_weight_split: Bm == slice0 + slice1 + slice2 bit for bit, layout [K/16][3][Npad][16], zero padding rows."""
    monkeypatch.setattr(ops, "GEMM_ARITH", "bf16x3")
    K, N = 96, 160
    gen = torch.Generator().manual_seed(5)
    Bm = (torch.randn(K, N, generator=gen) * torch.exp(4 * torch.randn(K, N, generator=gen))).cuda()
    Bx = ops.weight_split(Bm)
    Npad = 256
    assert Bx.numel() == 3 * Npad * K
    sl = _bf16_bits_to_float(Bx.view(K // 16, 3, Npad, 16)).permute(1, 2, 0, 3).reshape(3, Npad, K)
    assert (sl[:, N:, :] == 0).all()
    rec = (sl[0, :N].double() + sl[1, :N].double() + sl[2, :N].double()).t()
    assert torch.equal(rec, Bm.double())
    # slices are ordered by magnitude: |slice1| <= 2^-7 |slice0|, |slice2| <= 2^-15 |slice0|
    assert (sl[1].abs() <= sl[0].abs() * 2.0 ** -7).all() and (sl[2].abs() <= sl[0].abs() * 2.0 ** -15).all()


@pytest.mark.parametrize("M,Ka,N", [(2048, 256, 256), (1000, 128, 64)])
def test_bf16x3_error_is_fp32_class(ops, monkeypatch, M, Ka, N):
    """The split-bf16 contraction is an fp32 contraction: against float64, its error stays within 4x of the native
    f32 MFMA's on inputs with a wide dynamic range (a plain bf16 product would be ~1e4 times worse)."""
    gen = torch.Generator().manual_seed(M)
    A = [(torch.randn(M, Ka, generator=gen) * torch.exp(2 * torch.randn(M, Ka, generator=gen))).cuda()
         for _ in range(3)]
    Bm = (torch.randn(3 * Ka, N, generator=gen) * torch.exp(2 * torch.randn(3 * Ka, N, generator=gen))).cuda()
    ref = torch.cat(A, 1).double() @ Bm.double()
    scale = torch.cat(A, 1).abs().double() @ Bm.abs().double()          # sum |a||b|: the natural error scale
    errs = {}
    for mode in ("f32", "bf16x3"):
        monkeypatch.setattr(ops, "GEMM_ARITH", mode)
        (C,), _ = ops.gemm_planes(A, Ka, 0, Bm, None, M, N, 1, False)
        errs[mode] = ((C.double() - ref).abs() / scale).max().item()
    print("max |err| / sum|a||b|:", errs)
    assert errs["f32"] < 5e-6 and errs["bf16x3"] < 5e-6, errs              # ~ sqrt(K) * 2^-24
    assert errs["bf16x3"] <= 4.0 * errs["f32"] + 2.4e-7, errs              # 2.4e-7 = 2 fp32 ulps of the scale


@pytest.mark.parametrize("M,Ka,N", [(2048, 256, 256), (1000, 128, 64)])
def test_f16x2_error_is_fp32_class(ops, monkeypatch, M, Ka, N):
    """Two scaled fp16 slices, three slice products: against float64 the error stays within 4x of the native f32 MFMA's
    (+ 2^-20 of sum |a||b|: 22-bit operands and the dropped low x low product) on inputs with a wide dynamic range, for
    the plane contraction and for the weight-gradient contraction; a plain fp16 product would be ~1e3 times worse."""
    gen = torch.Generator().manual_seed(M)
    A = [(torch.randn(M, Ka, generator=gen) * torch.exp(2 * torch.randn(M, Ka, generator=gen))).cuda()
         for _ in range(3)]
    Bm = (torch.randn(3 * Ka, N, generator=gen) * torch.exp(2 * torch.randn(3 * Ka, N, generator=gen))).cuda()
    G = (torch.randn(M, N, generator=gen) * torch.exp(2 * torch.randn(M, N, generator=gen))).cuda()
    Z = torch.cat(A, 1).double()
    ref, scale = Z @ Bm.double(), Z.abs() @ Bm.abs().double()
    ref_t, scale_t = Z.t() @ G.double(), Z.abs().t() @ G.abs().double()
    errs, errs_t = {}, {}
    for mode in ("f32", "f16x2"):
        monkeypatch.setattr(ops, "GEMM_ARITH", mode)
        (C,), _ = ops.gemm_planes(A, Ka, 0, Bm, None, M, N, 1, False)
        errs[mode] = ((C.double() - ref).abs() / scale).max().item()
        P, _, nch = ops.gemm_tn(A, Ka, 0, G, M, N)
        errs_t[mode] = ((P.double().sum(0) - ref_t).abs() / scale_t).max().item()
    print("max |err| / sum|a||b|:", errs, "weight gradient:", errs_t)
    for e in (errs, errs_t):
        assert e["f32"] < 5e-6 and e["f16x2"] < 5e-6, e
        assert e["f16x2"] <= 4.0 * e["f32"] + 2.0 ** -20, e


@pytest.mark.parametrize("magnitude", [1e-30, 1e-7, 1.0, 1e30])
def test_f16x2_is_scale_free_and_zero_safe(ops, monkeypatch, magnitude):
    """The power-of-two scaling makes the result independent of the operands' magnitudes (fp16 alone spans 2^-24..2^16):
    gradients of 1e-7, activations of 1e30 - the same relative error; an all-zero operand gives exact zeros; the amax
    word a contraction hands on (want_amax) is the maximum of what it stored."""
    monkeypatch.setattr(ops, "GEMM_ARITH", "f16x2")
    M, Ka, N = 640, 64, 128
    gen = torch.Generator().manual_seed(11)
    A = torch.randn(M, Ka, generator=gen).cuda()
    Bm = (torch.randn(Ka, N, generator=gen) / 8).cuda()
    ref = A.double() @ Bm.double()
    (C,), _ = ops.gemm_planes([A * magnitude], Ka, 0, Bm, None, M, N, 1, False, want_amax=True)
    assert ((C.double() / magnitude - ref).abs().max() < 2e-6 * ref.abs().max()).item()
    word = C._p2m_amax.view(torch.float32).item()
    assert word == C.abs().max().item()
    (Cw,), _ = ops.gemm_planes([A], Ka, 0, Bm * magnitude, None, M, N, 1, False)
    assert ((Cw.double() / magnitude - ref).abs().max() < 2e-6 * ref.abs().max()).item()
    (Z,), _ = ops.gemm_planes([torch.zeros_like(A)], Ka, 0, Bm, None, M, N, 1, False)
    assert (Z == 0).all()
    # one huge outlier: everything within 2^-18 of it keeps 22 bits, the rest an ABSOLUTE error of 2^-40 of the outlier
    A2 = A.clone()
    A2[3, 5] = 3.0e4
    (C2,), _ = ops.gemm_planes([A2], Ka, 0, Bm, None, M, N, 1, False)
    ref2 = A2.double() @ Bm.double()
    assert ((C2.double() - ref2).abs().max() < 2e-6 * ref2.abs().max()).item()
    rows = torch.arange(M, device="cuda") != 3
    assert ((C2.double() - ref2)[rows].abs().max() < 4e-6 * ref[rows.cpu()].abs().max()).item()
    # a NaN stays in its own row, as in fp32: the amax reduction skips non-finite values, the scale is that of the finite data
    (C0,), _ = ops.gemm_planes([A], Ka, 0, Bm, None, M, N, 1, False)
    A3 = A.clone()
    A3[7, 3] = float("nan")
    (C3,), _ = ops.gemm_planes([A3], Ka, 0, Bm, None, M, N, 1, False)
    assert torch.isnan(C3[7]).all()
    assert torch.equal(C3[rows7 := (torch.arange(M, device="cuda") != 7)], C0[rows7])
    A3[7, 3] = float("inf")                     # ... and so does an infinity (it does not enter the amax word either)
    (C4,), _ = ops.gemm_planes([A3], Ka, 0, Bm, None, M, N, 1, False)
    assert not torch.isfinite(C4[7]).any()
    assert torch.equal(C4[rows7], C0[rows7])


def test_f16x2_weight_image(ops, monkeypatch):
    """p2m_weight_split, P2M_ARITH_F16X2: [K/16][2][Npad][16] fp16 slices of Bm 2^sb followed by the weight's amax word;
    slice 0 + slice 1 reproduces Bm 2^sb to 2^-22 of each value that is within 2^-18 of the maximum."""
    monkeypatch.setattr(ops, "GEMM_ARITH", "f16x2")
    K, N = 96, 160
    gen = torch.Generator().manual_seed(5)
    Bm = (torch.randn(K, N, generator=gen) / 7).cuda()
    Bx = ops.weight_split(Bm)
    Npad = 256
    assert Bx.numel() == 2 * Npad * K + 8
    amax = Bx[2 * Npad * K:2 * Npad * K + 2].view(torch.float32).item()
    assert amax == Bm.abs().max().item()
    sb = 14 - int(np.floor(np.log2(amax)))                    # amax 2^sb in [2^14, 2^15)
    sl = Bx[:2 * Npad * K].view(torch.float16).view(K // 16, 2, Npad, 16).permute(1, 2, 0, 3).reshape(2, Npad, K)
    assert (sl[:, N:, :] == 0).all()
    rec = (sl[0, :N].double() + sl[1, :N].double()).t() * 2.0 ** -sb
    assert ((rec - Bm.double()).abs() <= 2.0 ** -22 * Bm.double().abs() + 2.0 ** -40 * amax).all()


@pytest.mark.parametrize("V,Fdim,shift,B", [(736, 128, 0, 5), (736, 128, 1, 3), (1472, 64, 1, 9), (2944, 32, 0, 4),
                                            (736, 256, 0, 2), (1472, 256, 1, 6)])
def test_tiled_basis_is_bitwise_the_full_basis(ops, V, Fdim, shift, B):
    """k_basis_tile (LDS-staged union of the neighbourhoods of a tile of real rows) keeps the merged-CSR order and
    the fmaf chain of the row kernel: the compact planes equal the real rows of the full planes bit for bit."""
    L = _band_graph(V, 100 + V + shift)
    g = ops.DeviceGraph(L, "cuda:0")
    assert g.n_fake > 0 and g.n_real >= 256 and g.plan_tiles[shift] > 0      # the tile kernel is what runs
    gen = torch.Generator().manual_seed(V + Fdim)
    X = torch.randn(B * (V >> shift), Fdim, generator=gen).cuda()
    T1, T2 = ops.cheb_basis_fwd(g, X, B, Fdim, shift)
    T1c, T2c = ops.cheb_basis_fwd_real(g, X, B, Fdim, shift)
    order = g.real_ids_host()                  # compact row i <-> vertex order[i] (a locality order since round 5)
    assert np.array_equal(np.sort(order), _real_ids(L))
    real = torch.as_tensor(order.astype(np.int64), device="cuda")
    assert real.numel() == g.n_real
    ref1 = T1.view(B, V, Fdim)[:, real].reshape(-1, Fdim)
    ref2 = T2.view(B, V, Fdim)[:, real].reshape(-1, Fdim)
    assert torch.equal(T1c, ref1) and torch.equal(T2c, ref2)


@pytest.mark.parametrize("V,Fin,Fout,B,fuse", [(1472, 128, 64, 3, True), (736, 256, 128, 5, False),
                                               (2944, 64, 256, 2, True)])
@pytest.mark.parametrize("tile_gemm", [False, True])
def test_paired_backward_equals_fine_backward_pair_summed(ops, arith, monkeypatch, tile_gemm, V, Fin, Fout, B, fuse):
    """Backward of an un-pooled conv at the coarse resolution (include/p2m.h "paired operator"): S L g / S L2 g from the
    paired tile plan, row sets 3 / 4, against the fine-resolution backward followed by the pair-sum.  tile_gemm: the
    real rows through p2m_cheb_tile_gemm (plan 2: the planes are formed inside the contraction; opt-in P2M_TILE_GEMM=1)."""
    if tile_gemm and arith == "f32":
        pytest.skip("the basis-inside-the-contraction kernels exist in the slice arithmetics only (bf16x3: VALU gather; "
                    "f16x2: gather on the matrix cores for N <= 128, VALU gather for N = 256)")
    monkeypatch.setattr(ops, "TILE_GEMM", tile_gemm)
    L = _band_graph(V, 31 + V)
    g = ops.DeviceGraph(L, "cuda:0")
    assert g.plan_tiles[2] > 0 and g.pair and g.n_pair_real + g.n_pair_fake == V // 2 and g.n_pair_fake > 0
    real = np.zeros(V, bool)
    real[_real_ids(L)] = True
    pr = np.where(real[0::2] | real[1::2])[0]
    assert pr.size == g.n_pair_real
    gen = torch.Generator().manual_seed(V + Fout)
    M, Mc = B * V, B * V // 2
    gy = torch.randn(M, Fout, generator=gen).cuda()
    Xc = torch.randn(Mc, Fin, generator=gen).cuda()
    W3 = (torch.randn(3 * Fout, Fin, generator=gen) / 16).cuda()
    Gres = torch.randn(M, Fin, generator=gen).cuda() if fuse else None
    # fine resolution
    E1, E2 = ops.cheb_basis_fwd(g, gy, B, Fout, 0)
    (dXf,), _ = ops.gemm_planes([gy, E1, E2], Fout, 0, W3, None, M, Fin, 1, False, addend=Gres)
    dX_ref = ops.pair_sum(dXf, Mc, Fin)
    Pw, Pb, nch = ops.gemm_tn([Xc], Fin, 1, [gy, E1, E2], M, 3 * Fout)
    dW_ref, db_ref = ops.weight_grad_unpack(Pw, Pb, nch, Fout, Fin, 3, layout=1)
    # the planes themselves
    P1c, P2c = ops.cheb_basis_pair(g, gy, B, Fout)
    pri = torch.as_tensor(pr, device="cuda")
    for Pc, E in ((P1c, E1), (P2c, E2)):
        ref = ops.pair_sum(E, Mc, Fout).view(B, V // 2, Fout)[:, pri].reshape(-1, Fout)
        assert (Pc - ref).abs().max() < 1e-5 * max(1.0, ref.abs().max().item())
    # coarse resolution
    opb = ops.split_operands(W3, Fout, Fin, g.fake_a, g.fake_b)
    dX = torch.full((Mc, Fin), float("nan"), device="cuda")
    add = ops.pair_sum(Gres, Mc, Fin) if fuse else None
    P0, P1c, P2c = ops.conv_pair(g, B, gy, Fout, W3, add, dX, Fin, opb)
    assert torch.isfinite(dX).all()
    assert (dX - dX_ref).abs().max() < 2e-5 * max(1.0, dX_ref.abs().max().item())
    Q1, Qb1, n1 = ops.gemm_tn_rows(g, 3, B, Xc, Fin, 0, [P0, P1c, P2c], Fout, True)
    Q2, Qb2, n2 = ops.gemm_tn_rows(g, 4, B, Xc, Fin, 0, [P0], Fout, False)
    dW, db = ops.weight_grad_unpack2(Q1, Qb1, n1, Q2, Qb2, n2, g.fake_a, g.fake_b, Fout, Fin)
    assert (dW - dW_ref).abs().max() < 1e-4 * max(1.0, dW_ref.abs().max().item())
    assert (db - db_ref).abs().max() < 1e-3


@pytest.mark.parametrize("V,B,graph", [(1472, 11, "band"), (736, 3, "band"), (184, 5, "rand")])
def test_narrow_conv_combine_and_expand(ops, V, B, graph):
    """The final conv's sparse halves (project-then-combine by linearity): Y = P0 + L P1 + L2 P2 + bias and its
    transpose-side E = [G | L G | L2 G | 0], tile-plan kernels + row kernel on split levels, row kernel alone below,
    against float64 sparse algebra; the real-only form with an output order (inference) against the same."""
    L = _band_graph(V, 5 + V) if graph == "band" else _rand_graph(V, V + 2)
    g = ops.DeviceGraph(L, "cuda:0")
    assert (g.plan_tiles[0] > 0) == (graph == "band")
    Ld = torch.as_tensor(L.toarray(), dtype=torch.float64)
    L2d = 2 * Ld @ Ld - torch.eye(V, dtype=torch.float64)
    gen = torch.Generator().manual_seed(V)
    P = torch.randn(B, V, 32, generator=gen)
    bias = torch.randn(3, generator=gen)
    Pd = P.double()
    ref = Pd[..., 0:3] + Ld @ Pd[..., 3:6] + L2d @ Pd[..., 6:9] + bias.double()
    Y = ops.cheb_combine_small(g, P.cuda().view(B * V, 32), 3, bias.cuda(), B)
    assert (Y.view(B, V, 3).cpu().double() - ref).abs().max() < 2e-5
    G = torch.randn(B, V, 3, generator=gen)
    E = ops.cheb_expand_small(g, G.cuda().view(B * V, 3), 3, 32, B).view(B, V, 32).cpu().double()
    Gd = G.double()
    assert (E[..., 0:3] - Gd).abs().max() == 0 and (E[..., 9:] != 0).sum() == 0
    assert (E[..., 3:6] - Ld @ Gd).abs().max() < 1e-5 and (E[..., 6:9] - L2d @ Gd).abs().max() < 1e-5
    if g.split:
        real = _real_ids(L)
        nv = real.size - 7
        order = np.random.default_rng(3).permutation(real)[:nv]            # vertex stored at output row k
        inv = np.full(V, -1, dtype=np.int32)
        inv[order] = np.arange(nv, dtype=np.int32)
        Yo = ops.cheb_combine_small_real(g, P.cuda().view(B * V, 32), 3, bias.cuda(), B,
                                         torch.from_numpy(inv).cuda(), nv, 1000.0)
        assert (Yo.cpu().double() - 1000.0 * ref[:, order]).abs().max() < 2e-2


def test_classes_of_identical_fake_rows(ops):
    """include/p2m.h "classes": with runs of identical fake rows declared, the kernels that see them give what the full
    computation gives -- weighted statistics, class-sum gradients, holes neither read (they are NaN here) nor needed."""
    V, F, B = 1472, 128, 3
    L = _band_graph(V, 77, fake_frac=0.55)
    g = ops.DeviceGraph(L, "cuda:0")
    fake = g.fake_ids_host()
    rep, fmask = ops.class_representatives(V, fake, 3)
    sizes = np.bincount(rep, minlength=V)
    assert (sizes > 1).sum() > 20 and sizes.max() >= 4            # the test graph has real classes
    n_fake_all = g.n_fake
    g.set_classes(rep)
    assert g.classes and g.n_fake == int(((rep == np.arange(V)) & fmask).sum()) and g.n_fake_all == n_fake_all
    hole = torch.as_tensor(rep != np.arange(V), device="cuda")
    rep_t = torch.as_tensor(rep, device="cuda", dtype=torch.long)
    gen = torch.Generator().manual_seed(5)
    M = B * V
    # a FULL tensor whose class members are identical, and its holed twin
    y_full = (torch.randn(B, V, F, generator=gen) * 1.5 + 0.3).cuda()[:, rep_t].contiguous()
    y_hole = y_full.clone()
    y_hole[:, hole] = float("nan")
    # --- forward statistics: real-row tiles (as the contraction epilogue emits them) + weighted representative tiles
    real = torch.as_tensor(np.where(~fmask)[0], device="cuda")
    tr = ops.stats_tile_rows()
    tps = (real.numel() + tr - 1) // tr
    st_real = torch.empty(B * tps, 2, F, device="cuda")
    for b in range(B):
        for t in range(tps):
            blk = y_full[b, real[t * tr:(t + 1) * tr]].double()
            st_real[b * tps + t, 0] = blk.sum(0).float()
            st_real[b * tps + t, 1] = ((blk - blk.mean(0)) ** 2).sum(0).float()
    ntf = (g.n_fake + tr - 1) // tr
    st_fake = torch.empty(B * ntf, 2, F, device="cuda")
    from pose2mesh_release_amd import _lib
    _lib.check(_lib.hip().p2m_stats_rows_w(g.handle, y_hole.data_ptr(), B, F, st_fake.data_ptr(), None), "stats_rows_w")
    gamma, beta = (torch.rand(F, generator=gen) + 0.5).cuda(), (torch.randn(F, generator=gen) * 0.1).cuda()
    # the representatives' contraction emits the same weighted partials from its epilogue (both tile widths, N = 64 / 128)
    for Fout in (64, 128):
        Wt = (torch.randn(F, Fout, generator=gen) / 11).cuda()
        bias = torch.randn(Fout, generator=gen).cuda()
        c = torch.full((M, Fout), float("nan"), device="cuda")
        st_epi = ops.gemm_planes_rows(g, 2, B, [y_hole.view(M, F)], F, 0, False, Wt, bias, None, c, Fout, True)
        st_sep = torch.empty(B * ntf, 2, Fout, device="cuda")
        _lib.check(_lib.hip().p2m_stats_rows_w(g.handle, c.data_ptr(), B, Fout, st_sep.data_ptr(), None), "stats_rows_w")
        assert st_epi.shape == st_sep.shape and torch.isfinite(st_epi).all()
        assert (st_epi - st_sep).abs().max().item() < 2e-5 * st_sep.abs().max().item()
    co = ops.bn_finalize_rows(g, B, st_real, st_fake, gamma, beta, None, None, 0.1, 1e-5)
    yd = y_full.view(M, F).double()
    assert (co[0].double() - yd.mean(0)).abs().max() < 1e-6
    assert (co[1].double() - 1.0 / torch.sqrt(yd.var(0, unbiased=False) + 1e-5)).abs().max() < 1e-5
    # --- forward activation skips the holes
    x_full = ops.bn_act_fwd(y_full.view(M, F), co, True, None, 0, 0, M, F)
    x_hole = ops.bn_act_fwd(y_hole.view(M, F), co, True, None, 0, 0, M, F, classes=g)
    live = ~hole.repeat(B)
    assert torch.equal(x_hole[live], x_full[live])
    # --- backward: per-member gradients on the full side, class sums on the holed side
    gx_full = torch.randn(M, F, generator=gen).cuda()
    gx_cls = ops.class_reduce(g, gx_full, B, F)
    assert torch.equal(gx_cls.view(B, V, F)[:, real], gx_full.view(B, V, F)[:, real])
    assert float(gx_cls.view(B, V, F)[:, hole].abs().max()) == 0.0
    gx_cls.view(B, V, F)[:, hole] = float("nan")
    gy_full, dg_full, db_full = ops.bn_relu_bwd(gx_full, y_full.view(M, F), co, gamma, True, True, M, F)
    gy, dg, db, pgx, pgy = ops.bn_relu_bwd(gx_cls, y_hole.view(M, F), co, gamma, True, True, M, F, pair_in=True,
                                           pair_out=True, classes=g, zero_holes=True)
    assert (dg - dg_full).abs().max() < 2e-5 * max(1.0, dg_full.abs().max().item()) * np.sqrt(M)
    assert (db - db_full).abs().max() < 2e-5 * max(1.0, db_full.abs().max().item()) * np.sqrt(M)
    ref_cls = ops.class_reduce(g, gy_full, B, F)                  # class sums of the full result
    assert torch.isfinite(gy).all() and float(gy.view(B, V, F)[:, hole].abs().max()) == 0.0
    # the statistics differ in summation order only -> the coefficients, hence gy, to round-off
    assert (gy - ref_cls).abs().max() < 5e-5 * max(1.0, ref_cls.abs().max().item())
    # pair-sums leave the holes out: S(class sums) = the class sums one level up (a class halves, its sum stays)
    both = fmask[0::2] & fmask[1::2]
    rep_c = np.where(both, rep[0::2] // 2, np.arange(V // 2))
    rc = torch.as_tensor(rep_c, device="cuda", dtype=torch.long)
    live_c = torch.as_tensor(rep_c == np.arange(V // 2), device="cuda")

    def coarse_class_sums(t):
        out = torch.zeros(B, V // 2, F, device="cuda", dtype=torch.float64)
        out.index_add_(1, rc, ops.pair_sum(t, M // 2, F).view(B, V // 2, F).double())
        return out[:, live_c]
    assert (pgy.view(B, V // 2, F)[:, live_c].double() - coarse_class_sums(gy_full)).abs().max() \
        < 1e-4 * max(1.0, ref_cls.abs().max().item())
    assert (pgx.view(B, V // 2, F)[:, live_c].double() - coarse_class_sums(gx_full)).abs().max() < 1e-5
    assert float(pgy.view(B, V // 2, F)[:, ~live_c].abs().max()) == 0.0          # hole parents: untouched zeros
    # the same pass without the by-products (k_bn_bwd_apply's every-row walk) and at a width outside {32, 64, 128, 256} (the
    # generic kernel): live rows as above, zeros stored at the holes - nothing pre-zeroes the outputs (round 5)
    gy1, _, _ = ops.bn_relu_bwd(gx_cls, y_hole.view(M, F), co, gamma, True, True, M, F, classes=g, zero_holes=True)
    assert torch.equal(gy1, gy)
    Fg = F - 8
    cog = co[:, :Fg].contiguous()
    gyg, _, _ = ops.bn_relu_bwd(gx_cls[:, :Fg].contiguous(), y_hole.view(M, F)[:, :Fg].contiguous(), cog, gamma[:Fg].contiguous(),
                                True, True, M, Fg, classes=g, zero_holes=True)
    assert torch.isfinite(gyg).all() and float(gyg.view(B, V, Fg)[:, hole].abs().max()) == 0.0
    assert (gyg - gy[:, :Fg]).abs().max() <= 2e-6 * max(1.0, float(gy.abs().max()))
    gy.view(B, V, F)[:, hole] = float("nan")
    assert torch.equal(ops.pair_sum(gy, M // 2, F, classes=g).view(B, V // 2, F)[:, live_c],
                       pgy.view(B, V // 2, F)[:, live_c])


@pytest.mark.parametrize("N,ntiles", [(32, 40), (64, 3000), (128, 23552), (256, 6000), (3, 700), (200, 5000)])
def test_statistics_finalize_is_exact_and_repeatable(hip_libs, N, ntiles):
    """The two-stage statistics finalize (csrc/bn.hip: stage 1 writes double partials to a per-stream scratch, stage 2 sums them
    in a fixed order), forward and backward form, for every split count (48 / 96 / 192) and odd widths: the values match
    float64, and 60 repetitions with 256 MB of unrelated traffic in between are bitwise the first one (written for the
    single-launch form of round 5 - last-ticket block runs stage 2 - which passed it and was 1.5 ms per step slower)."""
    from pose2mesh_release_amd import _lib, ops
    gen = torch.Generator().manual_seed(N * 7 + ntiles)
    tr = ops.stats_tile_rows()
    M = ntiles * tr - 5
    st = torch.randn(ntiles, 2, N, generator=gen)
    st[:, 1] = st[:, 1].abs() * tr
    st[:, 0] *= tr ** 0.5
    gamma, beta = torch.rand(N, generator=gen) + 0.5, torch.randn(N, generator=gen)
    cnt = torch.full((ntiles,), float(tr), dtype=torch.float64)
    cnt[-1] = tr - 5
    sd = st.double()
    mean = sd[:, 0].sum(0) / M
    var = ((sd[:, 1] + sd[:, 0] ** 2 / cnt[:, None]).sum(0) / M - mean ** 2).clamp_min(0)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    stc, gc, bc = st.cuda(), gamma.cuda(), beta.cuda()
    junk = torch.empty(64 << 20, device="cuda")
    # backward form: nblk partial rows [nblk, 2, N] -> dbeta, dgamma, coef
    nblk = max(1, ntiles // 2)
    part = torch.randn(nblk, 2, N, generator=gen).cuda()
    pd = part.double().cpu()
    first = first_b = None
    for rep in range(60):
        co = ops.bn_finalize(stc, M, gc, bc, None, None, 0.1, 1e-5)
        dg, db, coef = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(2, N, device="cuda")
        _lib.check(_lib.hip().p2m_bn_bwd_finalize(part.data_ptr(), nblk, M, dg.data_ptr(), db.data_ptr(), coef.data_ptr(), 0, N,
                                                  None), "p2m_bn_bwd_finalize")
        junk.fill_(float(rep))
        if first is None:
            first, first_b = co.clone(), (dg.clone(), db.clone(), coef.clone())
            assert (co[0].double().cpu() - mean).abs().max() <= 1e-6 * max(1.0, float(mean.abs().max()))
            assert ((co[1].double().cpu() - invstd) / invstd).abs().max() <= 1e-5
            assert (db.double().cpu() - pd[:, 0].sum(0)).abs().max() <= 1e-6 * max(1.0, float(pd[:, 0].sum(0).abs().max()))
            assert (dg.double().cpu() - pd[:, 1].sum(0)).abs().max() <= 1e-6 * max(1.0, float(pd[:, 1].sum(0).abs().max()))
            assert (coef[0].double().cpu() - pd[:, 0].sum(0) / M).abs().max() <= 1e-6
        else:
            assert torch.equal(co, first), rep
            assert all(torch.equal(a, b) for a, b in zip((dg, db, coef), first_b)), rep


def _real_ids(L):
    """Vertices whose row is not the lone diagonal (the complement of the isolated padding vertices)."""
    L = L.tocsr()
    deg = np.diff(L.indptr)
    return np.where(~((deg == 1) & (L.indices[L.indptr[:-1].clip(max=L.nnz - 1)] == np.arange(L.shape[0]))))[0]


def test_row_kernel_basis_variant_in_a_subprocess(hip_libs):
    """P2M_BASIS_TILED is read once per process by the library (no mutable global state), so the row-per-wave basis kernel
    for the real rows of split levels - part of the independent kernel set of the parity tests - runs the basis / split
    tests in a child process."""
    import os
    import subprocess
    import sys
    child_env = dict(os.environ, P2M_BASIS_TILED="0")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_ops.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k", "fake_vertex or tiled_basis or cheb_basis or paired_backward"],
                       env=child_env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("V,Fin,Fout,shift,B", [(736, 128, 128, 0, 5), (736, 128, 128, 1, 3), (1472, 256, 128, 0, 4),
                                                (1472, 128, 256, 1, 9), (2944, 128, 64, 0, 2), (736, 64, 128, 0, 7),
                                                (736, 256, 256, 1, 6)])
@pytest.mark.parametrize("slices", ["bf16x3", "f16x2"])
def test_basis_inside_the_contraction_matches_basis_plus_contraction(ops, monkeypatch, slices, V, Fin, Fout, shift, B):
    """p2m_cheb_tile_gemm (the Chebyshev planes formed per tile in LDS and contracted without touching HBM) against the
    two-kernel form it replaces, p2m_cheb_basis_fwd_real + p2m_gemm_planes_rows: same real rows of C (fp32 round-off:
    the k order of the accumulation differs), untouched fake rows, the optional planes BITWISE those of the basis kernel
    (same entry order, same fmaf chain), BatchNorm statistics through their own finalize, addend, fused activation.
    B is deliberately not a multiple of the 4 samples a block tile holds."""
    monkeypatch.setattr(ops, "GEMM_ARITH", slices)
    monkeypatch.setattr(ops, "TILE_GEMM", True)          # opt-in path (P2M_TILE_GEMM=1)
    L = _band_graph(V, 7 + V + shift)
    g = ops.DeviceGraph(L, "cuda:0")
    assert g.plan_tiles[shift] > 0 and ops.tile_gemm_ok(g, shift, Fin, Fout)
    gen = torch.Generator().manual_seed(V + Fin + Fout)
    M = B * V
    X = torch.randn(B * (V >> shift), Fin, generator=gen).cuda()
    Wt = (torch.randn(3 * Fin, Fout, generator=gen) / (3 * Fin) ** 0.5).cuda()
    bias = torch.randn(Fout, generator=gen).cuda()
    add = torch.randn(M, Fout, generator=gen).cuda()
    Bx = ops.weight_split(Wt)
    T1c, T2c = ops.cheb_basis_fwd_real(g, X, B, Fin, shift)
    real = torch.as_tensor(_real_ids(L), device="cuda")
    fake = torch.ones(V, dtype=torch.bool, device="cuda")
    fake[real] = False
    for addend, act, stats in ((None, None, True), (add, None, False),
                               (None, (torch.rand(Fout).cuda() + 0.5, torch.randn(Fout).cuda(), True), False)):
        yref = torch.full((M, Fout), 7.0, device="cuda")
        st_ref = ops.gemm_planes_rows(g, 1, B, [X, T1c, T2c], Fin, shift, True, Wt, bias, addend, yref, Fout, stats,
                                      Bx=Bx, act=act)
        y = torch.full((M, Fout), 7.0, device="cuda")
        st, planes = ops.cheb_tile_gemm(g, shift, X, X, Fin, Bx, bias, addend, y, Fout, B, stats=stats,
                                        want_planes=stats, act=act)
        torch.cuda.synchronize()
        assert torch.isfinite(y).all()
        assert torch.equal(y.view(B, V, Fout)[:, fake], yref.view(B, V, Fout)[:, fake])        # fake rows untouched
        err = (y - yref).abs().max().item()
        assert err < 2e-5 * max(1.0, yref.abs().max().item()), err
        if stats:
            from pose2mesh_release_amd import _lib
            mg = bool(_lib.hip().p2m_cheb_tile_gemm_mg(ops.arith_code(), Fout))
            if not mg:                  # VALU gather (N = 256; P2M_MG_EXACT=0): the fmaf chain of the basis kernel, bit for bit
                assert torch.equal(planes[0], T1c) and torch.equal(planes[1], T2c)
            else:                       # N <= 128: the planes come off the matrix cores, fp32 sums in another order - of
                                        # 22-bit operands (f16x2) or of exact bf16 slices (bf16x3, round 5: fp32 round-off only)
                tol = 2e-6 if slices == "f16x2" else 5e-7
                for got, ref in ((planes[0], T1c), (planes[1], T2c)):
                    assert (got - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
            gamma, beta = (torch.rand(Fout) + 0.5).cuda(), torch.randn(Fout).cuda()
            We = ops.weight_eff(Wt, Fin, Fout, g.fake_a, g.fake_b)
            st2 = ops.gemm_planes_rows(g, 2, B, [X], Fin, shift, False, We, bias, None, yref, Fout, True)
            co_ref = ops.bn_finalize_rows(g, B, st_ref, st2, gamma, beta, None, None, 0.1, 1e-5)
            co = ops.bn_finalize_tiles(g, shift, B, st, st2, gamma, beta, None, None, 0.1, 1e-5)
            assert (co - co_ref).abs().max() < 2e-5 * max(1.0, co_ref.abs().max().item())
    # float64 anchor for the real rows (the two HIP forms above could share an error)
    Xd = X.double().view(B, V >> shift, Fin)
    if shift:
        Xd = Xd.repeat_interleave(2, dim=1)
    Ld = torch.from_numpy(sp.csr_matrix(L).toarray()).double().cuda()
    T1d = torch.einsum("uv,bvf->buf", Ld, Xd)
    T2d = 2 * torch.einsum("uv,bvf->buf", Ld, T1d) - Xd
    yd = torch.cat((Xd, T1d, T2d), dim=2) @ Wt.double() + bias.double()
    y = torch.zeros((M, Fout), device="cuda")
    ops.cheb_tile_gemm(g, shift, X, X, Fin, Bx, bias, None, y, Fout, B)
    err = (y.view(B, V, Fout)[:, real].double() - yd[:, real]).abs().max().item()
    assert err < 2e-5 * max(1.0, yd.abs().max().item()), err


@pytest.mark.parametrize("V,Fin,Fout,B,addend", [(736, 128, 128, 5, False), (1472, 128, 64, 9, True), (2944, 64, 128, 4, True)])
def test_tile_kernel_sums_the_next_batchnorm_backward_reduction(ops, monkeypatch, V, Fin, Fout, B, addend):
    """p2m_cheb_tile_gemm bnr_* (round 6): while the kernel copies its staged output tile g out it sums g m and g m yhat
    (m = [y scale + shift > 0], yhat = (y - mean) invstd) over the rows it stores; with p2m_bn_bwd_reduce_fake for the
    fake-vertex rows and p2m_bn_bwd_finalize over both partial sets this IS ops.bn_relu_bwd (d gamma, d beta, g_y) - against
    the stand-alone pass on the same g and against a float64 sum.  B is not a multiple of the 4-sample groups."""
    monkeypatch.setattr(ops, "GEMM_ARITH", "bf16x3")
    monkeypatch.setattr(ops, "TILE_GEMM", True)
    monkeypatch.setattr(ops, "BN_FUSE", True)
    L = _band_graph(V, 11 + V)
    g = ops.DeviceGraph(L, "cuda:0")
    assert g.plan_tiles[0] > 0 and ops.tile_bnr_slots(g, 0, Fout, B) > 0
    gen = torch.Generator().manual_seed(V + Fin)
    M = B * V
    X = torch.randn(M, Fin, generator=gen).cuda()
    Wt = (torch.randn(3 * Fin, Fout, generator=gen) / (3 * Fin) ** 0.5).cuda()
    add = torch.randn(M, Fout, generator=gen).cuda() if addend else None
    y = torch.randn(M, Fout, generator=gen).cuda()                       # raw output of the conv in front
    gamma, beta = (torch.rand(Fout, generator=gen) + 0.5).cuda(), torch.randn(Fout, generator=gen).cuda()
    mean, var = y.mean(0), y.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    co = torch.stack([mean, invstd, gamma * invstd, beta - mean * gamma * invstd]).contiguous()
    Bx = ops.weight_split(Wt)
    We = ops.weight_eff(Wt, Fin, Fout, g.fake_a, g.fake_b)
    part, nfake = ops.bnr_parts(g, 0, Fout, B, X.device)
    part.fill_(float("nan"))                                             # every slot must be written
    C = torch.zeros(M, Fout, device="cuda")
    ops.cheb_tile_gemm(g, 0, X, X, Fin, Bx, None, add, C, Fout, B, want_planes=True, bnr=(y, co, part[nfake:]))
    ops.gemm_planes_rows(g, 2, B, [X], Fin, 0, False, We, None, add, C, Fout, False)       # the fake-vertex rows of C
    res = ops.bn_relu_bwd(C, y, co, gamma, True, True, M, Fout, real_part=(g, part, nfake))
    ref = ops.bn_relu_bwd(C, y, co, gamma, True, True, M, Fout)
    torch.cuda.synchronize()
    assert torch.isfinite(part).all()
    # float64 sums over all rows
    m = (y.double() * co[2].double() + co[3].double()) > 0
    gm = torch.where(m, C.double(), torch.zeros_like(C, dtype=torch.float64))
    dbeta64 = gm.sum(0)
    dgamma64 = (gm * ((y.double() - mean.double()) * invstd.double())).sum(0)
    for got, r64, r32 in ((res[2], dbeta64, ref[2]), (res[1], dgamma64, ref[1])):
        scale = float(r64.abs().max())
        assert float((got.double() - r64).abs().max()) <= 2e-5 * scale
        assert float((r32.double() - r64).abs().max()) <= 2e-5 * scale
    assert float((res[0] - ref[0]).abs().max()) <= 1e-5 * float(ref[0].abs().max())
    with pytest.raises(Exception):                                       # N = 256 has no fused form
        ops.cheb_tile_gemm(g, 0, X, X, Fin, ops.weight_split(torch.randn(3 * Fin, 256).cuda()), None, None,
                           torch.zeros(M, 256, device="cuda"), 256, B, bnr=(y, co, part[nfake:]))


def test_exact_matrix_core_gather_in_a_subprocess(hip_libs):
    """Round 5: k_cheb_mg_gemm in the three-bf16-slice arithmetic (operator block and operands as exact bf16 triples, 2 samples
    per unit) is opt-in - P2M_MG_EXACT=1, read once per process by the library - because it only ties the VALU-gather kernel
    (profiles/r05_d_probe_mg*.txt).  Its parity runs here in a child process: the tile-kernel op tests (C against basis kernel +
    plane contraction and against float64, planes to fp32 round-off, BatchNorm partials, addend, fused activation), the
    activation on load, and the paired backward, all in bf16x3."""
    import os
    import subprocess
    import sys
    child_env = dict(os.environ, P2M_MG_EXACT="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_ops.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k",
                        "(basis_inside_the_contraction or activation_on_load or paired_backward) and bf16x3"],
                       env=child_env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_tile_kernel_v2_in_a_subprocess(hip_libs):
    """Round 6: k_cheb_tile_gemm_v2 (N <= 128, bf16x3) - union rows by LDS-DMA into a union image double-buffered in the sample
    dimension, plane 0 from that image, activation applied where x is read - is opt-in (P2M_TILE_V2=1, read once per process
    by the library): built to take the global-load issue off the producers' critical path, measured 4-9 % SLOWER than the
    round-5 kernel (profiles/r06_v2_*.txt).  Its parity runs here in a child process: C against basis
    kernel + plane contraction and float64, planes BITWISE the basis kernel's, BatchNorm partials, addend, fused activation,
    activation on load, the paired backward."""
    import os
    import subprocess
    import sys
    child_env = dict(os.environ, P2M_TILE_V2="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_ops.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k",
                        "(basis_inside_the_contraction or activation_on_load or paired_backward) and bf16x3"],
                       env=child_env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("V,Fin,Fout,B", [(736, 128, 128, 5), (1472, 128, 64, 3), (2944, 64, 128, 2)])
@pytest.mark.parametrize("slices", ["f16x2", "bf16x3"])
def test_activation_on_load_is_bitwise_the_separate_pass(ops, monkeypatch, slices, V, Fin, Fout, B):
    """include/p2m.h "activation on load" (lib/models/backbones/cheby_graph_conv.py:39 + lib/models/meshnet.py:100 folded into the
    next conv's loads): the tile kernel and the weight-gradient contraction reading the RAW conv output y with
    relu(y * scale + shift) applied between the global load and the LDS image give, bit for bit, what they give on the
    tensor x that p2m_bn_act_fwd materialises (same two roundings, same amax word) - tile kernel, LDS-staged basis kernel,
    plane contraction over the real and the fake rows, both weight-gradient launches; p2m_act_bound really bounds the
    activated tensor.  Round 5: in BOTH slice arithmetics (bf16x3 = the exact path of the bench's headline: no amax words, the
    VALU-gather tile kernel applies the activation in its producer waves)."""
    monkeypatch.setattr(ops, "GEMM_ARITH", slices)
    monkeypatch.setattr(ops, "TILE_GEMM", True)
    L = _band_graph(V, 11 + V)
    g = ops.DeviceGraph(L, "cuda:0")
    assert ops.fold_act_ok(g, Fin, Fout, B)
    gen = torch.Generator().manual_seed(V + Fin)
    M = B * V
    y = (torch.randn(M, Fin, generator=gen) * 3.0 + 0.7).cuda()
    scale = (torch.randn(Fin, generator=gen) * 0.8).cuda()          # both signs, as trained BatchNorm weights may have
    shift = torch.randn(Fin, generator=gen).cuda()
    co = torch.stack((torch.zeros_like(scale), torch.ones_like(scale), scale, shift)).contiguous()
    x = ops.bn_act_fwd(y, co, True, None, 0, 0, M, Fin)            # the separate pass: every row
    real = torch.as_tensor(_real_ids(L), device="cuda")
    fake = torch.ones(V, dtype=torch.bool, device="cuda")
    fake[real] = False
    # the bound: >= the true maximum, and not absurdly loose on this input
    word = None
    if slices == "f16x2":
        yw = ops.amax_of(y)
        word = ops.act_bound(co[2], co[3], yw, ops.new_amax("cuda:0"))
        bound = word.view(torch.float32).item()
        true_max = x.abs().max().item()
        assert true_max <= bound <= 8.0 * true_max, (true_max, bound)
    # forward: tile kernel on x vs on y with the activation on load, SAME amax word -> same slices -> same bits
    Wt = (torch.randn(3 * Fin, Fout, generator=gen) / (3 * Fin) ** 0.5).cuda()
    bias = torch.randn(Fout, generator=gen).cuda()
    Bx = ops.weight_split(Wt)
    c_ref = torch.zeros((M, Fout), device="cuda")
    st_ref, _ = ops.cheb_tile_gemm(g, 0, x, x, Fin, Bx, bias, None, c_ref, Fout, B, stats=True, amax=word)
    c = torch.zeros((M, Fout), device="cuda")
    st, _ = ops.cheb_tile_gemm(g, 0, y, y, Fin, Bx, bias, None, c, Fout, B, stats=True, amax=word,
                               in_act=(co[2], co[3]))
    torch.cuda.synchronize()
    assert torch.equal(c, c_ref) and torch.equal(st, st_ref)
    # weight gradient: X^T [g | E1 | E2] over the real rows
    gy = torch.randn(M, Fout, generator=gen).cuda()
    E1, E2 = ops.cheb_basis_fwd_real(g, gy, B, Fout, 0)
    ga = ops.amax_of(gy, g, B)
    P_ref, Pb_ref, n_ref = ops.gemm_tn_rows(g, 1, B, x, Fin, 0, [gy, E1, E2], Fout, True, a_amax=word, g_amax=ga,
                                            g_bits=g.plane_bits)
    P, Pb, n = ops.gemm_tn_rows(g, 1, B, y, Fin, 0, [gy, E1, E2], Fout, True, a_amax=word, g_amax=ga,
                                g_bits=g.plane_bits, a_act=(co[2], co[3]))
    torch.cuda.synchronize()
    assert n == n_ref and torch.equal(P, P_ref) and torch.equal(Pb, Pb_ref)
    if B >= 2:                       # ... and with two whole samples per chunk (round 6: the policy of the large batches)
        Ps_ref, Pbs_ref, ns_ref = ops.gemm_tn_rows(g, 1, B, x, Fin, 0, [gy, E1, E2], Fout, True, a_amax=word, g_amax=ga,
                                                   g_bits=g.plane_bits, samples_per_chunk=2)
        Ps, Pbs, ns = ops.gemm_tn_rows(g, 1, B, y, Fin, 0, [gy, E1, E2], Fout, True, a_amax=word, g_amax=ga,
                                       g_bits=g.plane_bits, a_act=(co[2], co[3]), samples_per_chunk=2)
        torch.cuda.synchronize()
        assert ns == ns_ref == -(-B // 2) and torch.equal(Ps, Ps_ref) and torch.equal(Pbs, Pbs_ref)
        assert (Ps.sum(0) - P.sum(0)).abs().max() < 2e-5 * max(1.0, P.sum(0).abs().max().item())
    # the two-kernel form (LDS-staged basis kernel + plane contraction, what the N = 256 levels run) and the fake-row /
    # narrow-projection contractions: planes and C bitwise those on the materialised x
    T1r, T2r = ops.cheb_basis_fwd_real(g, x, B, Fin, 0)
    T1a, T2a = ops.cheb_basis_fwd_real(g, y, B, Fin, 0, in_act=(co[2], co[3]))
    assert torch.equal(T1a, T1r) and torch.equal(T2a, T2r)
    We = ops.weight_eff(Wt, Fin, Fout, g.fake_a, g.fake_b)
    Wex = ops.weight_split(We)
    for rs, planes_x, planes_y, Bm, Bxx, compact in ((1, [x, T1r, T2r], [y, T1a, T2a], Wt, Bx, True),
                                                     (2, [x], [y], We, Wex, False)):
        c_ref = torch.zeros((M, Fout), device="cuda")
        st_ref = ops.gemm_planes_rows(g, rs, B, planes_x, Fin, 0, compact, Bm, bias, None, c_ref, Fout, True, Bx=Bxx,
                                      amax=word, amax_bits=g.plane_bits if rs == 1 else 0)
        c = torch.zeros((M, Fout), device="cuda")
        st = ops.gemm_planes_rows(g, rs, B, planes_y, Fin, 0, compact, Bm, bias, None, c, Fout, True, Bx=Bxx,
                                  amax=word, amax_bits=g.plane_bits if rs == 1 else 0, in_act=(co[2], co[3]))
        torch.cuda.synchronize()
        assert torch.equal(c, c_ref) and torch.equal(st, st_ref), rs
    P2r, Pb2r, n2r = ops.gemm_tn_rows(g, 2, B, x, Fin, 0, [gy], Fout, False, a_amax=word, g_amax=ga)
    P2, Pb2, n2 = ops.gemm_tn_rows(g, 2, B, y, Fin, 0, [gy], Fout, False, a_amax=word, g_amax=ga, a_act=(co[2], co[3]))
    assert n2 == n2r and torch.equal(P2, P2r) and torch.equal(Pb2, Pb2r)
    # the activation on load is refused where no kernel implements it
    from pose2mesh_release_amd._lib import P2MError
    with pytest.raises(P2MError):
        ops.cheb_tile_gemm(g, 0, y, y, Fin, Bx, bias, None, c, Fout, B, want_planes=True, amax=word, in_act=(co[2], co[3]))
