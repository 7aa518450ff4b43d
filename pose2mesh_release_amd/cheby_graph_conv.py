"""graph_conv_cheby -- drop-in for the reference's functional op
(lib/models/backbones/cheby_graph_conv.py:5-42), same signature and semantics:

    y = graph_conv_cheby(x, cl, bn, L, Fout, K)
      x  : (B, V, Fin) fp32 on the GPU
      cl : nn.Linear(Fin*K, Fout)   (weight column index = fin*K + k, as the reference stacks it)
      bn : nn.BatchNorm1d(Fout) or None   (statistics over all B*V rows; running stats updated in train())
      L  : rescaled Laplacian -- scipy sparse, torch sparse (what the reference passes) or an ops.DeviceGraph
    returns (B, V, Fout); no activation inside (the caller applies ReLU, meshnet.py:100).

Differentiable w.r.t. x, cl.weight, cl.bias, bn.weight, bn.bias.  K = 3 is the native kernel path
(every Pose2Mesh layer); K = 1, 2 reuse it with zero-padded weight planes; K > 3 runs the recurrence on the HIP basis
kernel (both of its planes used per launch) and the stock nn.Linear / BatchNorm1d.
"""
import threading

import torch

from . import ops
from ._lib import P2MError

_graph_cache = {}
_graph_lock = threading.Lock()      # nn.DataParallel calls the op from one thread per GPU


def _device_graph(L, device):
    if isinstance(L, ops.DeviceGraph):
        return L
    dev = torch.device(device)
    key = (id(L), dev.index if dev.index is not None else torch.cuda.current_device())
    with _graph_lock:
        hit = _graph_cache.get(key)
        if hit is not None and hit[0] is L:
            return hit[1]
        g = ops.DeviceGraph(L, torch.device("cuda", key[1]))
        if len(_graph_cache) > 256:
            _graph_cache.clear()
        _graph_cache[key] = (L, g)      # holding L keeps id(L) unique
        return g


def bn_momentum(bn):
    """nn.BatchNorm1d's exponential_average_factor: `momentum`, or the cumulative average 1/num_batches_tracked when
    momentum is None (torch/nn/modules/batchnorm.py; num_batches_tracked counts THIS batch already)."""
    if bn.momentum is not None:
        return float(bn.momentum)
    return 1.0 / float(int(bn.num_batches_tracked) + 1)


class _ChebConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, g, bn, training):
        B, V, Fin = x.shape
        Fout = weight.shape[0]
        if V != g.V:
            raise P2MError(f"x has {V} vertices but the graph has {g.V}")
        M = B * V
        with torch.cuda.device(x.device):
            # (amax words, f16x2 mode: drawn from the device's current chunk - ops.new_amax renews it when it runs out or when
            #  the capture state flips; a network built from this op does not pay a chunk per layer call)
            xc = x.contiguous().float().view(M, Fin)
            T1, T2 = ops.cheb_basis_fwd(g, xc, B, Fin, 0)
            Wt, W2, _ = ops.weight_pack(weight.contiguous(), Fin, 3, need_w2=True)
            stats = bn is not None and training
            (y,), st = ops.gemm_planes([xc, T1, T2], Fin, 0, Wt, bias.contiguous(), M, Fout, 1, stats,
                                       amax=ops.amax_of(xc), amax_bits=g.plane_bits)
            co = None
            out = y
            if bn is not None:
                if training:
                    track = bn.track_running_stats and bn.running_mean is not None
                    co = ops.bn_finalize(st, M, gamma.contiguous(), beta.contiguous(),
                                         bn.running_mean if track else None, bn.running_var if track else None,
                                         bn_momentum(bn) if track else 0.1, bn.eps)
                    if track:
                        bn.num_batches_tracked.add_(1)
                else:
                    co = ops.bn_eval_coeffs(gamma.contiguous(), beta.contiguous(), bn.running_mean, bn.running_var,
                                            bn.eps)
                out = ops.bn_act_fwd(y, co, False, None, 0, 0, M, Fout)
        ctx.stuff = (g, xc, T1, T2, y, co, W2, gamma, B, Fin, Fout, training, x.shape)
        return out.view(B, V, Fout)

    @staticmethod
    def backward(ctx, gout):
        g, xc, T1, T2, y, co, W2, gamma, B, Fin, Fout, training, xshape = ctx.stuff
        M = B * g.V
        with torch.cuda.device(gout.device):
            gx = gout.contiguous().float().view(M, Fout)
            dgamma = dbeta = None
            if co is not None:
                gy, dgamma, dbeta = ops.bn_relu_bwd(gx, y, co, gamma.contiguous(), False, training, M, Fout)
            else:
                gy = gx
            P, Pdb, nch = ops.gemm_tn([xc, T1, T2], Fin, 0, gy, M, Fout, a_amax=ops.amax_of(xc), a_bits=g.plane_bits)
            dW, db = ops.weight_grad_unpack(P, Pdb, nch, Fout, Fin, 3)
            d, _ = ops.gemm_planes([gy], Fout, 0, W2, None, M, 3 * Fin, 3, False)
            dX = ops.cheb_basis_bwd(g, d[0], d[1], d[2], None, B, Fin, 0)
        return dX.view(xshape), dW, db, dgamma, dbeta, None, None, None


class _LL2Fn(torch.autograd.Function):
    """(L x, L2 x) on the vertex axis, L2 = 2 L L - I: BOTH planes of one p2m_cheb_basis_fwd launch.  L and L2 are symmetric
    (lib/coarsening.py:23), so the backward is gx = L g1 + L2 g2 - one launch per incoming gradient that is not None."""

    @staticmethod
    def forward(ctx, x, g):
        B, V, F = x.shape
        ctx.g, ctx.B, ctx.F = g, B, F
        ctx.set_materialize_grads(False)        # an unused plane arrives as None in backward, not as a zero tensor
        with torch.cuda.device(x.device):
            T1, T2 = ops.cheb_basis_fwd(g, x.contiguous().float().view(B * V, F), B, F, 0)
        return T1.view(B, V, F), T2.view(B, V, F)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None and g2 is None:
            return None, None
        gx = None
        with torch.cuda.device((g1 if g1 is not None else g2).device):
            if g1 is not None:
                gx, _ = ops.cheb_basis_fwd(ctx.g, g1.contiguous().float().view(-1, ctx.F), ctx.B, ctx.F, 0)
            if g2 is not None:
                _, t2 = ops.cheb_basis_fwd(ctx.g, g2.contiguous().float().view(-1, ctx.F), ctx.B, ctx.F, 0)
                gx = t2 if gx is None else gx + t2
        return gx.view(ctx.B, -1, ctx.F), None


def _graph_conv_cheby_any_order(x, cl, bn, g, Fout, K):
    """Chebyshev orders K > 3 (lib/models/backbones/cheby_graph_conv.py:27-30 is generic in K; Pose2Mesh itself uses K = 3
    everywhere, which is what the fused kernels are built for).  Every launch of the HIP basis kernel yields L x and
    (2 L L - I) x, and both are used: x_1, x_2 come from x_0 as in the K = 3 path, then one launch on x_(k-1) gives
        x_k     = 2 L x_(k-1) - x_(k-2)                       (the reference's recurrence, :28)
        x_(k+1) = 2 (2 L L - I) x_(k-1) - x_(k-3)             (T_(n+2) = 2 T_2 T_n - T_(n-2): the same polynomial)
    - half the launches of a plane-by-plane recurrence and no discarded plane.  Then the reference's own (B*V, Fin*K)
    layout, column fin*K + k (:32-34), through nn.Linear / BatchNorm1d."""
    B, V, Fin = x.shape
    x0 = x.float()
    x1, x2 = _LL2Fn.apply(x0, g)
    planes = [x0, x1, x2]
    while len(planes) < K:
        k = len(planes)
        a, b = _LL2Fn.apply(planes[k - 1], g)
        planes.append(2 * a - planes[k - 2])
        if k + 1 < K:
            planes.append(2 * b - planes[k - 3])
    y = cl(torch.stack(planes[:K], dim=3).reshape(B * V, Fin * K))
    if bn is not None:
        y = bn(y)
    return y.view(B, V, Fout)


def graph_conv_cheby(x, cl, bn, L, Fout, K):
    if not x.is_cuda:
        raise P2MError("graph_conv_cheby (HIP) needs GPU tensors; this package has no CPU path")
    B, V, Fin = x.shape
    if cl.weight.shape != (Fout, Fin * K):
        raise P2MError(f"cl.weight is {tuple(cl.weight.shape)}, expected {(Fout, Fin * K)}")
    g = _device_graph(L, x.device)
    if V != g.V:
        raise P2MError(f"x has {V} vertices but the graph has {g.V}")
    weight = cl.weight
    if K > 3:
        return _graph_conv_cheby_any_order(x, cl, bn, g, Fout, K)
    if K != 3:
        if K not in (1, 2):
            raise P2MError(f"Chebyshev order K must be a positive integer, not {K}")
        pad = weight.new_zeros(Fout, Fin, 3)
        weight = torch.cat((weight.view(Fout, Fin, K), pad[:, :, K:]), dim=2).reshape(Fout, Fin * 3)
    bias = cl.bias if cl.bias is not None else x.new_zeros(Fout)
    gamma = beta = None
    if bn is not None:
        if bn.num_features != Fout:
            raise P2MError(f"bn has {bn.num_features} features, expected {Fout}")
        # affine=False: identity scale / zero shift (constants, no gradient)
        gamma = bn.weight if bn.weight is not None else x.new_ones(Fout)
        beta = bn.bias if bn.bias is not None else x.new_zeros(Fout)
    training = bn.training if bn is not None else False
    if bn is not None and not bn.training and not bn.track_running_stats:
        training = True            # nn.BatchNorm semantics: no running stats -> always batch statistics
    return _ChebConvFn.apply(x, weight, bias, gamma, beta, g, bn, training)
