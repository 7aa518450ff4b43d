"""MeshNet -- the coarse-to-fine Chebyshev GCN of Pose2Mesh, on libp2m_hip.so.

Drop-in for the reference's lib/models/meshnet.py: same constructor signature, same parameter
names/shapes/initialisation (state dicts are interchangeable), same forward I/O
  forward(x: FloatTensor[B, J*5 or B,J,5]) -> FloatTensor[B, V0, 3]   (tree order incl. fake vertices)
but every device op is a hand-written gfx950 kernel reached through the C ABI (include/p2m.h):

  reference op (file:line)                              here
  ---------------------------------------------------  ------------------------------------------
  permute/view, 2x torch.sparse.mm, 2x cat, permute     p2m_cheb_basis_fwd  (one gather pass, CSR of L|2LL-I), or, on the
    (backbones/cheby_graph_conv.py:16-34)                 big levels, inside the contraction: p2m_cheb_tile_gemm
  cl(x) nn.Linear (cheby_graph_conv.py:37)              p2m_gemm_planes     (fp32 product as 2 scaled fp16 slices on the
                                                          matrix pipe - the amax words travel with the tensors -, BN
                                                          partials in the epilogue)
  bn(x) BatchNorm1d over B*V rows (:39) + F.relu        p2m_bn_finalize + p2m_bn_act_fwd
    (meshnet.py:100) + F.interpolate residual (:109)
  nn.Upsample x2 (meshnet.py:71-78)                     not materialised: consumers index r>>1
  self.fc (meshnet.py:105-106)                          p2m_gemm_planes
  autograd backward of all of the above                 p2m_bn_bwd_*, p2m_gemm_tn, p2m_gemm_planes,
                                                        p2m_cheb_basis_bwd, p2m_pair_sum, p2m_lerp_bwd_add
The whole network is ONE autograd.Function so that intermediate buffers are owned explicitly
(X|T1|T2 planes and raw conv outputs are kept for backward: ~145 MB per mesh, sized for 288 GB HBM).
"""
import sys

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import P2MError
from .cheby_graph_conv import bn_momentum

K_CHEB = 3


def channel_plan(mano, cin, cout):
    """Block/layer feature widths (lib/models/meshnet.py:21-33)."""
    if mano:
        return [(cin, 32, 64, 64), (64, 128, 256), (256, 256, 256), (256, 256, 256), (256, 256, 256),
                (256, 128, 128), (128, 64, cout)]
    return [(cin, 32, 64, 64), (64, 128, 256), (256, 256, 256), (256, 256, 256), (256, 256, 256),
            (256, 256, 256), (256, 128, 128), (128, 128, 128), (128, 128, 128), (128, 64, cout)]


def _cfg_says_mano():
    """Inside the reference's scripts the topology switch is cfg.DATASET.target_joint_set
    (lib/models/meshnet.py:21).  Honour it when that module is loaded; otherwise None."""
    mod = sys.modules.get("core.config")
    try:
        return None if mod is None else (mod.cfg.DATASET.target_joint_set == "mano")
    except AttributeError:
        return None


class _Layer:
    """Static description of one ChebConv (+BN+ReLU) layer."""
    __slots__ = ("ci", "block", "graph", "Fin", "Fout", "has_bn", "first_in_block", "last_in_block")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def _bwd_forward_form(L):
    """Layers whose backward runs as dX = [g|Lg|L2g] W3 on the plane GEMM (MFMA path needs 32-multiples)."""
    return L.Fout % 32 == 0 and L.Fin % 32 == 0


def _narrow(L):
    """Layers whose 3*Fout output columns fit one 32-wide MFMA tile take the project-then-combine path."""
    return (not L.has_bn) and K_CHEB * L.Fout <= 32 and L.Fin % 32 == 0 and not L.first_in_block and L.Fout <= 4


def _narrow_operands(W, L, wa):
    """[Fin, 32] operand of the project-then-combine path: columns k*Fout + fo = W[fo][fin*3 + k], zero padded.
    wa: the parameter's amax word (f16x2; every derived operand here is a permutation of the parameter)."""
    Wp = torch.nn.functional.pad(W.detach().view(L.Fout, L.Fin, K_CHEB).permute(1, 2, 0).reshape(L.Fin, -1),
                                 (0, 32 - K_CHEB * L.Fout)).contiguous()
    return Wp, ops.weight_split(Wp, wa)


def _transposed_operands(Wp, wa):
    Wpt = Wp.t().contiguous()
    return Wpt, ops.weight_split(Wpt, wa)


def _fc_operands(fw, wa):
    fwt, _, _ = ops.weight_pack(fw, fw.shape[1], 1, need_w2=False)
    return fwt, (ops.weight_split(fwt, wa) if fw.shape[0] % 32 == 0 and fw.shape[1] % 32 == 0 else None)


def _forward_inference(net, graphs, x, params):
    """eval() + no_grad fast path (Pose2Mesh.set_inference): SURVEY 8(f4) / A3 -- in eval() the real-vertex outputs are
    bitwise independent of the padding vertices, so at every split level only the real rows are computed (no fake-row
    contraction, no effective weight); eval-mode BatchNorm + ReLU ride in the contraction's epilogue (same two roundings
    as the separate pass: the real-vertex outputs are bitwise those of the general path); the last conv stores straight
    into mesh-model vertex order when an output order is set (lib/core/base.py:201-202 folded in).  Rows of padding
    vertices hold unspecified values in the intermediate tensors and zeros in a tree-order output."""
    wc = net._weight_cache
    P = net._param_index
    J, cin = net.num_joint, net.num_joint_input_chan
    x = x.reshape(-1, J * cin).contiguous().float()
    B = x.shape[0]
    cur, cur_shift = x.view(B * J, cin), 0
    nblk = len(net.CL_F)
    block_in, block_in_shift, block_in_F = None, 0, 0
    dev = x.device
    f16 = ops.f16x2()       # every contraction output is the next one's operand here: each comes back with its amax word
    for L in net._layers:
        g = graphs[L.graph]
        M = B * g.V
        if L.first_in_block:
            block_in, block_in_shift, block_in_F = cur, cur_shift, L.Fin
        W, bvec = params[P[f"cl.{L.ci}.weight"]], params[P[f"cl.{L.ci}.bias"]]
        if _narrow(L):
            Wp, Wpx = wc.get((L.ci, "narrow"), W, lambda: _narrow_operands(W, L, wc.get((L.ci, "amax"), W, lambda: ops.param_amax(W))))
            oi = net._out_index_on(dev)
            if g.split:
                Pm = torch.empty((M, 32), device=dev, dtype=torch.float32)
                ops.gemm_planes_rows(g, 1, B, [cur], L.Fin, cur_shift, False, Wp, None, None, Pm, 32, Bx=Wpx)
            else:
                (Pm,), _ = ops.gemm_planes([cur], L.Fin, cur_shift, Wp, None, M, 32, 1, False, Bx=Wpx)
            if oi is not None:
                return ops.cheb_combine_small_real(g, Pm, L.Fout, bvec, B, oi, net._out_nv, net._out_scale)
            cur = ops.cheb_combine_small_real(g, Pm, L.Fout, bvec, B) if g.split \
                else ops.cheb_combine_small(g, Pm, L.Fout, bvec, B)
            cur_shift = 0
            continue
        act = None
        if L.has_bn:
            bn = net.bn[L.ci]
            gamma, beta = params[P[f"bn.{L.ci}.weight"]], params[P[f"bn.{L.ci}.bias"]]
            co = wc.get((L.ci, "bn_eval"), (gamma, beta, bn.running_mean, bn.running_var),
                        lambda: ops.bn_eval_coeffs(gamma, beta, bn.running_mean, bn.running_var, bn.eps))
            act = (co[2], co[3], True)
        want_w3 = _bwd_forward_form(L)
        Wt, _, _ = wc.get((L.ci, "pack"), W, lambda: ops.weight_pack(W, L.Fin, K_CHEB, need_w2=not want_w3,
                                                                    need_w3=want_w3))
        mfma = L.Fin % 32 == 0 and L.Fout % 32 == 0
        if g.split and mfma:
            y = torch.empty((M, L.Fout), device=dev, dtype=torch.float32)
            Wtx = wc.get((L.ci, "wtx"), W, lambda: ops.weight_split(Wt, wc.get((L.ci, "amax"), W, lambda: ops.param_amax(W))))
            T1 = T2 = None
            ya = ops.tag_amax(y, ops.new_amax(dev))._p2m_amax if f16 else None
            if ops.tile_gemm_ok(g, cur_shift, L.Fin, L.Fout, B=B):
                ops.cheb_tile_gemm(g, cur_shift, cur, cur, L.Fin, Wtx, bvec, None, y, L.Fout, B, act=act, amax_out=ya)
            else:
                T1, T2 = ops.cheb_basis_fwd_real(g, cur, B, L.Fin, cur_shift)
                ops.gemm_planes_rows(g, 1, B, [cur, T1, T2], L.Fin, cur_shift, True, Wt, bvec, None, y, L.Fout, Bx=Wtx,
                                     act=act, amax=ops.amax_of(cur), amax_bits=g.plane_bits, amax_out=ya)
        else:
            T1, T2 = ops.cheb_basis_fwd(g, cur, B, L.Fin, cur_shift)
            Wtx = wc.get((L.ci, "wtx"), W, lambda: ops.weight_split(Wt, wc.get((L.ci, "amax"), W, lambda: ops.param_amax(W)))) if mfma else None
            (y,), _ = ops.gemm_planes([cur, T1, T2], L.Fin, cur_shift, Wt, bvec, M, L.Fout, 1, False, Bx=Wtx, act=act,
                                      amax=ops.amax_of(cur), amax_bits=g.plane_bits, want_amax=True)
        del T1, T2
        if L.has_bn and L.last_in_block and 1 <= L.block <= nblk - 2:        # residual (meshnet.py:108-115)
            y = ops.bn_act_fwd(y, None, False, block_in, block_in_F, block_in_shift, M, L.Fout,
                               real_rows=g if g.split else None)
        cur, cur_shift = y, 0
        if L.last_in_block:
            if L.block == 0:                                                  # fc lift (:104-106)
                h = ops.view_tagged(cur, B, J * L.Fout)
                fw, fb = params[P["fc.weight"]], params[P["fc.bias"]]
                fwt, fwx = wc.get("fc", fw, lambda: _fc_operands(fw, wc.get("fc_amax", fw, lambda: ops.param_amax(fw))))
                (u,), _ = ops.gemm_planes([h], fw.shape[1], 0, fwt, fb, B, fw.shape[0], 1, False, Bx=fwx, want_amax=True)
                cur = ops.view_tagged(u, B * net._Vc, net.CL_F[1][0])
            elif L.block < nblk - 2:
                cur_shift = 1
    return cur.view(B, graphs[0].V, net.num_mesh_output_chan)


class _MeshNetFn(torch.autograd.Function):
    """forward/backward of the whole coarse-to-fine stack (lib/models/meshnet.py:80-117)."""

    @staticmethod
    def forward(ctx, net, keep, x, *params):
        dev = x.device
        if not x.is_cuda:
            raise P2MError("Pose2Mesh (HIP) needs CUDA/ROCm tensors: there is no CPU path in this package")
        ctx.x_shape = x.shape
        ctx.saved_params = params
        with torch.cuda.device(dev):
            return _MeshNetFn._forward(ctx, net, keep, x, params)

    @staticmethod
    def _forward(ctx, net, keep, x, params):
        # `keep` (save activations for backward) is decided by the caller, where the grad mode is still visible:
        # ctx.needs_input_grad only reflects requires_grad of the inputs and stays True under torch.no_grad(), which
        # would make inference hold every layer's X / y until the forward returns (training-sized peak memory).
        graphs = net._graph_cache.on(x.device)
        training = net.training
        wc = net._weight_cache
        ops.amax_begin_step(x.device)
        # every slice image of the split convs' weights in two launches (ops.ConvWeightSet), refreshed when they changed
        cws = net._conv_weight_set(graphs, params, x.device)
        if cws is not None:
            cws.refresh()
        J, cin = net.num_joint, net.num_joint_input_chan
        if training:
            ops.bump_weight_epoch()       # running statistics change behind torch's back: cached eval coefficients are stale
        elif not keep and net._infer_real_only:
            ctx.saved = None
            return _forward_inference(net, graphs, x, params)
        x = x.reshape(-1, J * cin).contiguous().float()
        B = x.shape[0]
        P = net._param_index
        saved = []
        cur = x.view(B * J, cin)
        cur_shift = 0
        nblk = len(net.CL_F)
        block_in, block_in_shift, block_in_F = None, 0, 0
        fc_saved = None
        nbt = []           # num_batches_tracked of every BatchNorm that ran on batch statistics: incremented in ONE launch
        fold = None        # activation on load: (scale, shift) when `cur` IS the raw output y of the previous conv (tagged with the
                           # bound of its activated form, ops.fold_act_ok) - consumed by the very next conv, which applies
                           # relu(y * scale + shift) wherever it loads its input
        for L in net._layers:
            g = graphs[L.graph]
            M = B * g.V
            if L.first_in_block:
                block_in, block_in_shift, block_in_F = cur, cur_shift, L.Fin
            W, bvec = params[P[f"cl.{L.ci}.weight"]], params[P[f"cl.{L.ci}.bias"]]
            if _narrow(L):
                # final 64 -> 3 conv by linearity: project to 9 columns on the MFMA first, then combine sparsely
                Wp, Wpx = wc.get((L.ci, "narrow"), W, lambda: _narrow_operands(W, L, wc.get((L.ci, "amax"), W, lambda: ops.param_amax(W))))
                if g.classes:       # holes of `cur` hold no data: project the live rows only; the combine fills the holes
                    Pm = torch.empty((M, 32), device=cur.device, dtype=torch.float32)
                    for rs in (1, 2):
                        ops.gemm_planes_rows(g, rs, B, [cur], L.Fin, cur_shift, False, Wp, None, None, Pm, 32, Bx=Wpx,
                                             amax=ops.amax_of(cur) if fold is not None else None, in_act=fold)
                else:
                    (Pm,), _ = ops.gemm_planes([cur], L.Fin, cur_shift, Wp, None, M, 32, 1, False, Bx=Wpx)
                out = ops.cheb_combine_small(g, Pm, L.Fout, bvec, B)
                del Pm
                if keep:
                    saved.append((cur, cur_shift, None, None, None, None, Wp, fold))
                cur, cur_shift, fold = out, 0, None
                continue
            need_stats = L.has_bn and training
            bwd_fwdform = _bwd_forward_form(L)
            split = g.split and bwd_fwdform
            images = cws is not None and L.ci in cws.images      # this conv's slice images come from the batched set
            batched = split and images
            # packed / transposed weights: constant between optimizer steps -> cached per layer (ops.WeightCache); the
            # convs of the slice arithmetics read nothing but the images of `cws`: W stands in for the fp32 operands
            want_w3 = bwd_fwdform
            if images:
                Wt, W2, W3 = W, None, W
            else:
                Wt, W2, W3 = wc.get((L.ci, "pack"), W, lambda: ops.weight_pack(W, L.Fin, K_CHEB, need_w2=not want_w3,
                                                                               need_w3=want_w3))
            if split:
                # real / fake vertex launches: fake vertices are isolated, T1 = a x and T2 = b x, so they take a
                # K = Fin contraction with W0 + a W1 + b W2 and no basis planes at all
                y = torch.empty((M, L.Fout), device=cur.device, dtype=torch.float32)
                opf = cws.fwd(L.ci, W) if batched else wc.get(
                    (L.ci, "split_fwd"), W, lambda: ops.split_operands(Wt, L.Fin, L.Fout, g.fake_a, g.fake_b,
                                                                      wc.get((L.ci, "amax"), W, lambda: ops.param_amax(W))))
                # does the NEXT conv take this one's activation on load?  Then y's amax word is needed for the bound
                Ln = net._layers[L.ci + 1] if L.ci + 1 < len(net._layers) else None
                # (train mode only: the folded operand is scaled from a BOUND of its maximum, the separate pass from the exact
                #  maximum - in eval() the general path stays bitwise the real-rows-only inference path, tests/test_gpu_infer.py)
                fold_out = bool(training and L.has_bn and not L.last_in_block and Ln is not None and Ln.graph == L.graph
                                and (ops.fold_act_ok(g, Ln.Fin, 32, B, narrow=True) if _narrow(Ln) else
                                     (_bwd_forward_form(Ln) and ops.fold_act_ok(g, Ln.Fin, Ln.Fout, B))))
                yword = ops.new_amax(cur.device) if (fold_out and ops.f16x2()) else None
                T1, T2, st, st2, tiled = ops.conv_split(g, B, cur, L.Fin, cur_shift, Wt, bvec, None, y, L.Fout, g.fake_a,
                                                        g.fake_b, need_stats, operands=opf, want_planes=False,
                                                        amax_out=yword, in_act=fold)
                tile_rows = ("tiles", cur_shift) if tiled else "rows"
            else:
                T1, T2 = ops.cheb_basis_fwd(g, cur, B, L.Fin, cur_shift)
                if images:
                    Wtx = cws.images[L.ci][0]
                else:
                    Wtx = wc.get((L.ci, "wtx"), W, lambda: ops.weight_split(Wt, wc.get((L.ci, "amax"), W, lambda: ops.param_amax(W)))) \
                        if (L.Fin % 32 == 0 and L.Fout % 32 == 0) else None
                (y,), st = ops.gemm_planes([cur, T1, T2], L.Fin, cur_shift, Wt, bvec, M, L.Fout, 1, need_stats, Bx=Wtx,
                                           amax=ops.amax_of(cur), amax_bits=g.plane_bits)
                tile_rows = None
                fold_out, yword = False, None
            fold_in, fold = fold, None
            co = None
            if L.has_bn:
                bn = net.bn[L.ci]
                gamma, beta = params[P[f"bn.{L.ci}.weight"]], params[P[f"bn.{L.ci}.bias"]]
                if training and isinstance(tile_rows, tuple):
                    co = ops.bn_finalize_tiles(g, tile_rows[1], B, st, st2, gamma, beta, bn.running_mean, bn.running_var,
                                               bn_momentum(bn), bn.eps)
                    nbt.append(bn.num_batches_tracked)
                elif training and tile_rows == "rows":
                    co = ops.bn_finalize_rows(g, B, st, st2, gamma, beta, bn.running_mean, bn.running_var,
                                              bn_momentum(bn), bn.eps)
                    nbt.append(bn.num_batches_tracked)
                elif training:
                    co = ops.bn_finalize(st, M, gamma, beta, bn.running_mean, bn.running_var, bn_momentum(bn), bn.eps,
                                         tile_rows)
                    nbt.append(bn.num_batches_tracked)
                else:
                    co = ops.bn_eval_coeffs(gamma, beta, bn.running_mean, bn.running_var, bn.eps)
                resid = None
                if L.last_in_block and 1 <= L.block <= nblk - 2:       # meshnet.py:108-115
                    resid = block_in
                if fold_out:
                    # x = relu(bn(y)) is not materialised at all: the next conv's kernels (tile kernel / basis kernel +
                    # plane 0 of the contraction / fake-row contraction / narrow projection) and its weight gradients read y
                    # and apply the activation on load.  `out` is y under a second tensor object that carries the bound of
                    # the ACTIVATED tensor as its amax word.
                    out = y.view(M, L.Fout)
                    if yword is not None:     # (f16x2; the exact bf16 slices need no bound)
                        ops.tag_amax(out, ops.act_bound(co[2], co[3], yword, ops.new_amax(y.device)))
                    fold = (co[2], co[3])
                else:
                    out = ops.bn_act_fwd(y, co, True, resid, block_in_F, block_in_shift, M, L.Fout,
                                         classes=g if (g.classes and split) else None)
                if net._tap is not None:          # test hook: raw conv output + BN scale/shift of every ReLU layer
                    yt = y
                    if g.classes and g.split:     # holes are never computed: show the class value, as the full net has it
                        rep = torch.as_tensor(net._class_rep[L.graph], device=y.device, dtype=torch.long)
                        yt = y.view(B, g.V, L.Fout)[:, rep].reshape(M, L.Fout)
                    net._tap.append((L.ci, yt, co[2], co[3]))
            else:
                out = y                                               # final conv: no BN, no ReLU (:52-55,99)
            if keep:
                if bwd_fwdform:
                    T1 = T2 = None          # the backward never needs the basis of X (dW = X^T [g|Lg|L2g])
                saved.append((cur, cur_shift, T1, T2, y, co, W3 if bwd_fwdform else W2, fold_in))
            cur, cur_shift = out, 0
            if L.last_in_block:
                if L.block == 0:                                      # fc lift (:104-106)
                    h = ops.view_tagged(cur, B, J * L.Fout)
                    fw, fb = params[P["fc.weight"]], params[P["fc.bias"]]
                    Nf, Kf = fw.shape
                    if ops.GEMM_ARITH != "f32" and B % 4 == 0 and B >= 32 and Nf % 32 == 0 and Kf % 32 == 0:
                        # a batch-sized Linear as a reduction over its K input columns with both operands row-major over K
                        # (the weight-gradient-shaped contraction, split over K): h^T [K, B] and the transposed weight - a
                        # plane contraction would put 2 x N/128 blocks on 256 CUs (see the backward)
                        fwt = wc.get("fc_t", fw, lambda: ops.weight_pack(fw, Kf, 1, need_w2=False)[0])
                        ha = ops.amax_of(h)
                        Pu, _, nchu = ops.gemm_tn([ops.tag_amax(h.t().contiguous(), ha)], B, 0, fwt, Kf, Nf, a_amax=ha,
                                                  g_amax=wc.get("fc_amax", fw, lambda: ops.param_amax(fw)))
                        u = (Pu[0] if nchu == 1 else Pu.sum(0)).add_(fb)
                    else:
                        fwt, fwx = wc.get("fc", fw, lambda: _fc_operands(fw, wc.get("fc_amax", fw, lambda: ops.param_amax(fw))))
                        (u,), _ = ops.gemm_planes([h], Kf, 0, fwt, fb, B, Nf, 1, False, Bx=fwx, want_amax=True)
                    if keep:
                        fc_saved = h
                    cur = ops.view_tagged(u, B * net._Vc, net.CL_F[1][0])
                elif L.block < nblk - 2:                              # virtual x2 un-pool (:111)
                    cur_shift = 1
        if nbt:
            torch._foreach_add_(nbt, 1)      # nn.BatchNorm1d.forward's `num_batches_tracked += 1`, 20 launches as one
        V0 = graphs[0].V
        ctx.net, ctx.saved, ctx.fc_saved, ctx.B, ctx.training, ctx.cws = net, saved, fc_saved, B, training, cws
        ctx.n_params = len(params)
        ctx.param_shapes = [p.shape for p in params]
        return cur.view(B, V0, net.num_mesh_output_chan)

    @staticmethod
    def backward(ctx, grad_out):
        with torch.cuda.device(grad_out.device):
            ops.set_phase("_bwd")
            try:
                return _MeshNetFn._backward(ctx, grad_out)
            finally:
                ops.set_phase("")

    @staticmethod
    def _backward(ctx, grad_out):
        net, saved, B, training = ctx.net, ctx.saved, ctx.B, ctx.training
        if saved is None or len(saved) == 0:
            raise P2MError("backward called but the forward ran without gradient tracking (or twice)")
        graphs = net._graph_cache.on(grad_out.device)
        wc = net._weight_cache
        cws = ctx.cws
        P = net._param_index
        params = ctx.saved_params
        grads = [None] * ctx.n_params
        nblk = len(net.CL_F)
        J = net.num_joint

        def tgt(*names):
            """In-place gradient accumulation (Pose2Mesh.accumulate_grads_in_place): the parameters' .grad tensors when
            every one of them is a usable fp32 buffer (e.g. views into optim.FlatAdam.flat_grad), else None."""
            if not net._direct_grad:
                return None
            out = []
            for n in names:
                gbuf = params[P[n]].grad
                if gbuf is None or gbuf.dtype != torch.float32 or not gbuf.is_contiguous() \
                        or gbuf.device != grad_out.device:
                    return None
                out.append(gbuf)
            return out
        def ready(*names):
            """the kernels writing these parameters' gradients in place are enqueued on the CURRENT stream"""
            if net._grad_sink is not None and net._direct_grad:
                net._grad_sink([params[P[n]] for n in names])
        G = grad_out.contiguous().float().view(-1, net.num_mesh_output_chan)   # grad wrt current block output
        g_out = graphs[net._layers[-1].graph]
        if g_out.classes:      # class-sum form: a representative carries the gradient of its whole class, holes nothing
            G = ops.class_reduce(g_out, G, B, net.num_mesh_output_chan)
        g_cur = G
        keep = []          # tensors read by the side stream: kept alive until the join at the end of backward
        main_stream = torch.cuda.current_stream()
        side = ops.side_stream(grad_out.device) if ops.DW_SIDE_STREAM else None

        class side_ctx:
            """Runs the enclosed launches on the side stream, ordered after everything issued so far on main."""

            def __init__(self, keep_list, *tensors):
                keep_list.extend(tensors)

            def __enter__(self):
                if side is not None:
                    side.wait_stream(main_stream)
                    self.cm = torch.cuda.stream(side)
                    self.cm.__enter__()

            def __exit__(self, *a):
                if side is not None:
                    self.cm.__exit__(*a)
                return False
        Gs_block = None    # S G of the current block (pair-sum of the gradient w.r.t. the block output), when produced
        bnr_pending = None  # (graph, partial sums, fake blocks): the real-row half of the NEXT layer's BatchNorm-backward
                            # reduction, summed by the kernel that produced g_cur (ops.cheb_tile_gemm bnr=; round 6)
        for L in reversed(net._layers):
            gph = graphs[L.graph]
            M = B * gph.V
            X, x_shift, T1, T2, y, co, W2, fold_in = saved[L.ci]
            if L.last_in_block:
                if L.block == 0:
                    # fc backward (meshnet.py:105-106): G is d(fc out) as [B*Vc, 64]
                    fw = params[P["fc.weight"]]
                    dU = ops.view_tagged(G, B, fw.shape[0])
                    h = ctx.fc_saved
                    Nf, Kf = fw.shape
                    tg = tgt("fc.weight", "fc.bias")
                    if ops.GEMM_ARITH != "f32" and B % 4 == 0 and B >= 32 and Nf % 32 == 0 and Kf % 32 == 0:
                        # (the forward's gate: small / odd batches keep the plane contraction - their dh reduction would
                        #  have Ka = B and fall to the scalar kernel)
                        # A batch-sized Linear (M = B rows) is two more weight-gradient-shaped contractions (reductions over a
                        # long index with both operands row-major over it), not plane contractions - those would put
                        # 2 x N/128 blocks on 256 CUs:
                        #   dW[n, k] = sum_m dU[m, n] h[m, k]: the operands of p2m_gemm_tn swapped, so that the partials
                        #       come out in nn.Linear layout and the unpack is coalesced on both sides (layout 2);
                        #   dh[m, k] = sum_n dU^T[n, m] W[n, k]: W exactly as nn.Linear stores it - no packed / split copy.
                        dUa = ops.amax_of(dU)
                        Pw, _, nch = ops.gemm_tn([dU], Nf, 0, h, B, Kf, a_amax=dUa, g_amax=ops.amax_of(h))
                        dbv = dU.sum(0)
                        if tg is not None:
                            if nch == 1:      # one chunk (the reduction index is the batch): already THE gradient
                                tg[0].add_(Pw[0])
                            else:
                                ops.weight_grad_unpack(Pw, None, nch, Nf, Kf, 1, dW=tg[0], db=None, layout=2)
                            tg[1].add_(dbv)
                            ready("fc.weight", "fc.bias")
                        else:
                            grads[P["fc.weight"]] = Pw[0] if nch == 1 else \
                                ops.weight_grad_unpack(Pw, None, nch, Nf, Kf, 1, layout=2)[0]
                            grads[P["fc.bias"]] = dbv
                        dUt = ops.tag_amax(dU.t().contiguous(), dUa)
                        Ph, _, nch2 = ops.gemm_tn([dUt], B, 0, fw, Nf, Kf, a_amax=dUa,
                                                  g_amax=wc.get("fc_amax", fw, lambda: ops.param_amax(fw)))
                        dh = Ph[0] if nch2 == 1 else Ph.sum(0)
                    else:
                        Pw, Pb, nch = ops.gemm_tn([h], Kf, 0, dU, B, Nf)
                        if tg is not None:
                            ops.weight_grad_unpack(Pw, Pb, nch, Nf, Kf, 1, dW=tg[0], db=tg[1])
                            ready("fc.weight", "fc.bias")
                        else:
                            grads[P["fc.weight"]], grads[P["fc.bias"]] = ops.weight_grad_unpack(Pw, Pb, nch, Nf, Kf, 1)
                        (dh,), _ = ops.gemm_planes([dU], Nf, 0, fw, None, B, Kf, 1, False)
                    G = dh.view(B * J, L.Fout)
                g_cur = G
            if _narrow(L):
                Wp = W2
                E = ops.cheb_expand_small(gph, g_cur, L.Fout, 32, B)
                # E = [G | L G | L2 G | 0] is bounded by 2^plane_bits max |G|: one small pass over the 3-wide G instead of a pass
                # over the 32-wide E (holes of a class-reduced G are zero)
                ew = ops.amax_of(g_cur.view(-1)) if (gph.classes and g_cur.numel() % 4 == 0) else None
                ebits = gph.plane_bits if ew is not None else 0
                if gph.classes:     # live rows only (row sets 1 / 2): the holes of X and E hold no data
                    dW32 = db32 = None
                    for rs in (1, 2):
                        Pw, Pb, nch = ops.gemm_tn_rows(gph, rs, B, X, L.Fin, x_shift, [E], 32, False,
                                                       a_amax=ops.amax_of(X) if fold_in is not None else None, a_act=fold_in,
                                                       g_amax=ew, g_bits=ebits)
                        dW32, db32 = ops.weight_grad_unpack(Pw, Pb, nch, 32, L.Fin, 1, dW=dW32, db=db32)
                else:
                    Pw, Pb, nch = ops.gemm_tn([X], L.Fin, x_shift, E, M, 32)
                    dW32, db32 = ops.weight_grad_unpack(Pw, Pb, nch, 32, L.Fin, 1)
                nco = K_CHEB * L.Fout
                dWn = dW32[:nco].view(K_CHEB, L.Fout, L.Fin).permute(1, 2, 0).reshape(L.Fout, L.Fin * K_CHEB)
                tg = tgt(f"cl.{L.ci}.weight", f"cl.{L.ci}.bias")
                if tg is not None:
                    tg[0].add_(dWn)
                    tg[1].add_(db32[:L.Fout])
                    ready(f"cl.{L.ci}.weight", f"cl.{L.ci}.bias")
                else:
                    grads[P[f"cl.{L.ci}.weight"]] = dWn.contiguous()
                    grads[P[f"cl.{L.ci}.bias"]] = db32[:L.Fout].contiguous()
                Wl = params[P[f"cl.{L.ci}.weight"]]
                Wpt, Wptx = wc.get((L.ci, "narrow_bwd"), Wl, lambda: _transposed_operands(
                    Wp, wc.get((L.ci, "amax"), Wl, lambda: ops.param_amax(Wl))))
                if gph.classes:
                    dX = torch.empty((M, L.Fin), device=E.device, dtype=torch.float32)
                    for rs in (1, 2):
                        ops.gemm_planes_rows(gph, rs, B, [E], 32, 0, False, Wpt, None, None, dX, L.Fin, Bx=Wptx, amax=ew,
                                             amax_bits=ebits)
                else:
                    (dX,), _ = ops.gemm_planes([E], 32, 0, Wpt, None, M, L.Fin, 1, False, Bx=Wptx)
                saved[L.ci] = None
                g_cur = dX
                continue
            has_res = L.first_in_block and 1 <= L.block <= nblk - 2
            Fblk = net.CL_F[L.block][-1]
            fuse_res = has_res and (L.Fin == Fblk)
            pair_path = _bwd_forward_form(L) and gph.split and x_shift and gph.pair \
                and (L.Fout in (32, 64) or L.Fout % 128 == 0)
            # pair-sums this block's first conv will need come out of the BatchNorm-backward pass as by-products:
            # S G (the residual gradient at the coarser resolution) when G is read as g_cur by the block's last conv,
            # S gy (plane 0 of the paired operator) when the first conv's own gy is written
            want_Gs = want_P0 = False
            if L.has_bn and M % 2 == 0 and L.Fout in (32, 64, 128, 256):
                if L.last_in_block and not L.first_in_block:
                    F0 = net._layers[net._block_first[L.block]]
                    want_Gs = bool(saved[F0.ci][1]) and 1 <= F0.block <= nblk - 2 and L.Fout == Fblk
                want_P0 = pair_path
            # ---- BN + ReLU backward -> gy
            P0 = None
            if L.has_bn:
                gamma = params[P[f"bn.{L.ci}.weight"]]
                tg = tgt(f"bn.{L.ci}.weight", f"bn.{L.ci}.bias")
                kw = dict(dgamma=tg[0], dbeta=tg[1]) if tg is not None else {}
                res = ops.bn_relu_bwd(g_cur, y, co, gamma, True, training, M, L.Fout, pair_in=want_Gs,
                                      pair_out=want_P0, classes=gph, zero_holes=not gph.split, real_part=bnr_pending, **kw)
                bnr_pending = None
                gy = res[0]
                if tg is None:
                    grads[P[f"bn.{L.ci}.weight"]], grads[P[f"bn.{L.ci}.bias"]] = res[1], res[2]
                else:
                    ready(f"bn.{L.ci}.weight", f"bn.{L.ci}.bias")
                if want_Gs:
                    Gs_block = res[3]
                if want_P0:
                    P0 = res[4]
            else:
                gy = g_cur
            if pair_path:
                # the input was un-pooled (X_fine[r] = X[r >> 1]): with S = the pair-sum, dX = [S g | S L g | S L2 g] W3
                # and dW = X^T [S g | S L g | S L2 g] -- both contractions run over V/2 rows; S L and S L2 are one
                # baked operator (the level's paired tile plan), its rows split into "has a real child" / "both
                # children fake" like the real / fake split of the forward
                Mc = M >> 1
                # classes: the rows of hole parents are not written; a coarser level that runs full-row kernels needs
                # them to be zero (a hole carries no gradient), a split one never reads them
                coarse_full = gph.classes and not graphs[net._layers[L.ci - 1].graph].split
                dX = (torch.zeros if coarse_full else torch.empty)((Mc, L.Fin), device=gy.device, dtype=torch.float32)
                add = (Gs_block if Gs_block is not None else ops.pair_sum(G, Mc, Fblk, classes=gph)) if fuse_res else None
                Wl = params[P[f"cl.{L.ci}.weight"]]
                opb = cws.bwd(L.ci, Wl) if (cws is not None and L.ci in cws.images) else wc.get(
                    (L.ci, "split_bwd"), Wl, lambda: ops.split_operands(W2, L.Fout, L.Fin, gph.fake_a, gph.fake_b,
                                                                      wc.get((L.ci, "amax"), Wl, lambda: ops.param_amax(Wl))))
                P0, E1, E2 = ops.conv_pair(gph, B, gy, L.Fout, W2, add, dX, L.Fin, opb, P0=P0)
                ga = ops.amax_of(gy, gph, B)       # bounds S g (x 2) and the paired planes (x 2^(plane_bits + 1))
                with side_ctx(keep, X, P0, E1, E2):
                    Pw, Pb, nch = ops.gemm_tn_rows(gph, 3, B, X, L.Fin, 0, [P0, E1, E2], L.Fout, True, g_amax=ga,
                                                   g_bits=gph.plane_bits + 1)
                    Pw2, Pb2, nch2 = ops.gemm_tn_rows(gph, 4, B, X, L.Fin, 0, [P0], L.Fout, False, g_amax=ga, g_bits=1)
                    tg = tgt(f"cl.{L.ci}.weight", f"cl.{L.ci}.bias")
                    dW, db = ops.weight_grad_unpack2(Pw, Pb, nch, Pw2, Pb2, nch2, gph.fake_a, gph.fake_b, L.Fout,
                                                     L.Fin, *(tg or ()))
                    keep.extend((Pw, Pb, Pw2, Pb2))
                    if tg is not None:
                        ready(f"cl.{L.ci}.weight", f"cl.{L.ci}.bias")
                if tg is None:
                    grads[P[f"cl.{L.ci}.weight"]], grads[P[f"cl.{L.ci}.bias"]] = dW, db
                del Pw, Pb, Pw2, Pb2, P0, E1, E2, add
            elif _bwd_forward_form(L) and gph.split:
                # forward-form backward, split into real / fake vertex launches (see the forward)
                dXf = torch.empty((M, L.Fin), device=gy.device, dtype=torch.float32)
                add = G if fuse_res else None
                Wl = params[P[f"cl.{L.ci}.weight"]]
                opb = cws.bwd(L.ci, Wl) if (cws is not None and L.ci in cws.images) else wc.get(
                    (L.ci, "split_bwd"), Wl, lambda: ops.split_operands(W2, L.Fout, L.Fin, gph.fake_a, gph.fake_b,
                                                                      wc.get((L.ci, "amax"), Wl, lambda: ops.param_amax(Wl))))
                # dXf of a block's SECOND conv is exactly the gradient the first conv's BatchNorm backward reduces next (no
                # residual, no un-pool in between): its real rows are summed while the tile kernel stores them
                bnr = None
                if not L.first_in_block and not x_shift and L.ci > 0:
                    Lp, sp = net._layers[L.ci - 1], saved[L.ci - 1]
                    if Lp.has_bn and Lp.graph == L.graph and sp is not None and sp[5] is not None \
                            and ops.tile_gemm_ok(gph, 0, L.Fout, L.Fin, True, B=B):
                        pr = ops.bnr_parts(gph, 0, L.Fin, B, gy.device)
                        if pr is not None:
                            bnr = (sp[4], sp[5], pr[0][pr[1]:])
                            bnr_pending = (gph, pr[0], pr[1])
                E1, E2, _, _, _ = ops.conv_split(gph, B, gy, L.Fout, 0, W2, None, add, dXf, L.Fin, gph.fake_a,
                                                 gph.fake_b, operands=opb, bnr=bnr)
                dX = ops.pair_sum(dXf, M >> 1, L.Fin) if x_shift else dXf
                # the weight gradient is off the critical path (nothing downstream in backward reads it): it runs on
                # a side stream, so its MFMA work overlaps the HBM-bound BatchNorm / basis passes of the next layers
                # fold_in (forward: ops.fold_act_ok): X is the RAW output of the previous conv, tagged with the bound of its
                # activated form; both weight-gradient launches apply the activation on load
                xa_w = ops.amax_of(X) if fold_in is not None else None
                with side_ctx(keep, X, gy, E1, E2):
                    Pw, Pb, nch = ops.gemm_tn_rows(gph, 1, B, X, L.Fin, x_shift, [gy, E1, E2], L.Fout, True, a_amax=xa_w,
                                                   g_amax=ops.amax_of(gy, gph, B), g_bits=gph.plane_bits, a_act=fold_in)
                    Pw2, Pb2, nch2 = ops.gemm_tn_rows(gph, 2, B, X, L.Fin, x_shift, [gy], L.Fout, False, a_amax=xa_w,
                                                      a_act=fold_in)
                    tg = tgt(f"cl.{L.ci}.weight", f"cl.{L.ci}.bias")
                    dW, db = ops.weight_grad_unpack2(Pw, Pb, nch, Pw2, Pb2, nch2, gph.fake_a, gph.fake_b, L.Fout,
                                                     L.Fin, *(tg or ()))
                    keep.extend((Pw, Pb, Pw2, Pb2))
                    if tg is not None:
                        ready(f"cl.{L.ci}.weight", f"cl.{L.ci}.bias")
                if tg is None:
                    grads[P[f"cl.{L.ci}.weight"]], grads[P[f"cl.{L.ci}.bias"]] = dW, db
                del Pw, Pb, Pw2, Pb2, E1, E2, dXf
            elif _bwd_forward_form(L):
                # backward in FORWARD form (L symmetric): E = basis(gy), dX = [gy|E1|E2] W3 with the residual
                # gradient and the un-pool pair-sum in the GEMM epilogue, dW = X^T [gy|E1|E2].  One single-source
                # gather (12.5 rows/row) replaces the two-source gather of p2m_cheb_basis_bwd (25 rows/row).
                E1, E2 = ops.cheb_basis_fwd(gph, gy, B, L.Fout, 0)
                Wl = params[P[f"cl.{L.ci}.weight"]]
                if cws is not None and L.ci in cws.images:
                    W3x = cws.images[L.ci][2]
                else:
                    W3x = wc.get((L.ci, "w3x"), Wl, lambda: ops.weight_split(
                        W2, wc.get((L.ci, "amax"), Wl, lambda: ops.param_amax(Wl))))
                (dX,), _ = ops.gemm_planes([gy, E1, E2], L.Fout, 0, W2, None, M, L.Fin, 1, False,
                                           addend=G if fuse_res else None, pair_out=bool(x_shift), Bx=W3x,
                                           amax=ops.amax_of(gy), amax_bits=gph.plane_bits)
                with side_ctx(keep, X, gy, E1, E2):
                    Pw, Pb, nch = ops.gemm_tn([X], L.Fin, x_shift, [gy, E1, E2], M, K_CHEB * L.Fout,
                                              g_amax=ops.amax_of(gy), g_bits=gph.plane_bits)
                    tg = tgt(f"cl.{L.ci}.weight", f"cl.{L.ci}.bias")
                    if tg is not None:
                        ops.weight_grad_unpack(Pw, Pb, nch, L.Fout, L.Fin, K_CHEB, dW=tg[0], db=tg[1], layout=1)
                        ready(f"cl.{L.ci}.weight", f"cl.{L.ci}.bias")
                    else:
                        dW, db = ops.weight_grad_unpack(Pw, Pb, nch, L.Fout, L.Fin, K_CHEB, layout=1)
                    keep.extend((Pw, Pb))
                if tg is None:
                    grads[P[f"cl.{L.ci}.weight"]], grads[P[f"cl.{L.ci}.bias"]] = dW, db
                del Pw, Pb, E1, E2
            else:
                Pw, Pb, nch = ops.gemm_tn([X, T1, T2], L.Fin, x_shift, gy, M, L.Fout, a_amax=ops.amax_of(X),
                                          a_bits=gph.plane_bits)
                tg = tgt(f"cl.{L.ci}.weight", f"cl.{L.ci}.bias")
                if tg is not None:
                    ops.weight_grad_unpack(Pw, Pb, nch, L.Fout, L.Fin, K_CHEB, dW=tg[0], db=tg[1])
                    ready(f"cl.{L.ci}.weight", f"cl.{L.ci}.bias")
                else:
                    dW, db = ops.weight_grad_unpack(Pw, Pb, nch, L.Fout, L.Fin, K_CHEB)
                    grads[P[f"cl.{L.ci}.weight"]], grads[P[f"cl.{L.ci}.bias"]] = dW, db
                del Pw, Pb
                d, _ = ops.gemm_planes([gy], L.Fout, 0, W2, None, M, K_CHEB * L.Fin, K_CHEB, False)
                dX = ops.cheb_basis_bwd(gph, d[0], d[1], d[2], G if fuse_res else None, B, L.Fin, x_shift)
            if has_res and not fuse_res:                              # transpose of the feature resize
                Gs = (Gs_block if Gs_block is not None else ops.pair_sum(G, M >> 1, Fblk, classes=gph)) if x_shift else G
                ops.lerp_bwd_add(Gs, dX, M >> x_shift, Fblk, L.Fin)
            saved[L.ci] = None
            g_cur = dX
            if L.first_in_block:
                G = dX
                Gs_block = None
        if side is not None:
            main_stream.wait_stream(side)      # join: all weight gradients are complete before autograd consumes them
        keep.clear()
        ctx.saved = None
        gx = None
        if ctx.needs_input_grad[2]:
            gx = g_cur.view(ctx.x_shape)
        return (None, None, gx) + tuple(grads)


class Pose2Mesh(nn.Module):
    """Same constructor as the reference (lib/models/meshnet.py:12): graph_L is the list returned by
    build_coarse_graphs (levels+1 scipy matrices, finest first, joint graph last).  Differences, all
    deliberate: the caller's list is NOT mutated (the reference deletes graph_L[-2] and overwrites
    the entries, meshnet.py:35,62); the MANO/SMPL switch may be given explicitly (`mano=`), else it is
    taken from the reference's cfg when that is loaded, else from len(graph_L) (7 -> MANO, 10 -> SMPL).
    """

    def __init__(self, num_joint_input_chan, num_mesh_output_chan, graph_L, mano=None):
        super().__init__()
        self.num_joint_input_chan = num_joint_input_chan
        self.num_mesh_output_chan = num_mesh_output_chan
        graph_L = list(graph_L)
        if mano is None:
            mano = _cfg_says_mano()
        if mano is None:
            if len(graph_L) not in (7, 10):
                raise ValueError(f"cannot infer SMPL/MANO topology from {len(graph_L)} graph levels; pass mano=")
            mano = len(graph_L) == 7
        self.mano = bool(mano)
        self.CL_F = channel_plan(self.mano, num_joint_input_chan, num_mesh_output_chan)
        self.CL_K = [K_CHEB] * len(self.CL_F)
        need = len(self.CL_F)          # graphs used: blocks 0..n-2 one each, last block shares
        if len(graph_L) != need:
            raise ValueError(f"{'MANO' if self.mano else 'SMPL'} topology needs {need} graph levels "
                             f"(levels={need - 1} in build_coarse_graphs), got {len(graph_L)}")
        del graph_L[-2]                                                # meshnet.py:35 (on our copy)
        self.graph_L = graph_L
        self.num_joint = int(graph_L[-1].shape[0])
        self._Vc = int(graph_L[-2].shape[0])
        for a, b in zip(graph_L[:-2], graph_L[1:-1]):
            if a.shape[0] != 2 * b.shape[0]:
                raise ValueError("mesh levels must halve exactly (binary coarsening tree)")
        self.fc = nn.Linear(self.num_joint * self.CL_F[0][-1], self._Vc * self.CL_F[1][0])   # meshnet.py:36-37

        cl, bn, layers = [], [], []
        nblk = len(self.CL_F)
        for i, blk in enumerate(self.CL_F):
            for l in range(len(blk) - 1):
                Fin, Fout = blk[l], blk[l + 1]
                lin = nn.Linear(K_CHEB * Fin, Fout)
                scale = np.sqrt(2.0 / (K_CHEB * Fin + Fout))                 # meshnet.py:48-50
                lin.weight.data.uniform_(-scale, scale)
                lin.bias.data.fill_(0.0)
                cl.append(lin)
                last = (i == nblk - 1 and l == len(blk) - 2)
                bn.append(None if last else nn.BatchNorm1d(Fout))              # meshnet.py:52-55
                ldx = -(i + 1) + (1 if i == nblk - 1 else 0)                   # meshnet.py:91-93
                layers.append(_Layer(ci=len(cl) - 1, block=i, graph=len(graph_L) + ldx, Fin=Fin, Fout=Fout,
                                     has_bn=not last, first_in_block=(l == 0), last_in_block=(l == len(blk) - 2)))
        self.cl = nn.ModuleList(cl)
        self.bn = nn.ModuleList(bn)
        self._layers = layers
        self._block_first = {L.block: i for i, L in enumerate(layers) if L.first_in_block}
        self._class_rep = {}
        self._graph_cache = ops.GraphCache(graph_L, class_plan=self._class_plan)
        self._weight_cache = ops.WeightCache()
        self._conv_weight_sets = {}
        self._direct_grad = False
        self._grad_sink = None
        self._infer_real_only = False
        self._out_perm, self._out_nv, self._out_scale, self._out_index = None, 0, 1.0, {}
        self._tap = None        # tests set this to a list to receive (conv index, y_raw, bn scale, bn shift) per layer
        names, _ = self._param_list()
        self._param_index = {n: i for i, n in enumerate(names)}

    def _class_plan(self, graphs):
        """{graph index: un-pool steps above that level} when runs of identical fake rows may be declared on the levels
        of the un-pool chain (include/p2m.h "classes"; ops.CLASSES), else None.  Conditions: the split levels are the
        finest ones, every conv on them runs through the row-set paths (so no full-row kernel ever reads a hole), the
        backward of every un-pooled conv on them has the paired operator, and the fake vertices of consecutive levels
        agree with the tree (a vertex is fake iff both of its children are)."""
        nblk = len(self.CL_F)
        chain = [self._layers[self._block_first[b]].graph for b in range(1, nblk - 1)]      # coarse -> fine
        if not any(graphs[gi].split for gi in chain):
            return None
        seen = False
        for gi in chain:
            if graphs[gi].split:
                seen = True
            elif seen:
                return None
        for L in self._layers:
            g = graphs[L.graph]
            if L.graph not in chain or not g.split or _narrow(L):
                continue
            if not _bwd_forward_form(L):
                return None
            if L.first_in_block and 2 <= L.block <= nblk - 2 \
                    and not (g.pair and (L.Fout in (32, 64) or L.Fout % 128 == 0)):
                return None
        masks = []
        for gi in chain:
            m = np.zeros(graphs[gi].V, dtype=bool)
            m[graphs[gi].fake_ids_host()] = True
            masks.append(m)
        for k in range(len(chain) - 1):
            if graphs[chain[k + 1]].V != 2 * graphs[chain[k]].V \
                    or not np.array_equal(masks[k], masks[k + 1][0::2] & masks[k + 1][1::2]):
                return None
        plan = {gi: k for k, gi in enumerate(chain)}
        for gi, depth in plan.items():
            self._class_rep[gi] = ops.class_representatives(graphs[gi].V, np.where(masks[chain.index(gi)])[0], depth)[0]
        return plan

    def _conv_weight_set(self, graphs, params, device):
        """ops.ConvWeightSet over the split convs (slice arithmetics only), one per (device, arithmetic)."""
        if ops.GEMM_ARITH == "f32":
            return None
        key = (torch.device(device).index, ops.GEMM_ARITH)
        cws = self._conv_weight_sets.get(key)
        if cws is None:
            P = self._param_index
            # every conv whose contractions read slice images: the split convs (four images each) and the unsplit
            # MFMA-shaped ones of the coarse levels (they use the forward / backward image; their effective-weight images
            # are computed along and never read - cheaper than ~4 extra launches per layer and step)
            ent = [(L.ci, params[P[f"cl.{L.ci}.weight"]], graphs[L.graph].fake_a, graphs[L.graph].fake_b)
                   for L in self._layers if _bwd_forward_form(L) and not _narrow(L)]
            cws = ops.ConvWeightSet(ent, device) if ent else False
            self._conv_weight_sets[key] = cws
        if cws is False:
            return None
        # the same parameter objects as when the set was built? (replicas, re-created parameters)
        P = self._param_index
        if any(w is not params[P[f"cl.{k}.weight"]] for k, w in zip(cws.keys, cws.weights)):
            cws.weights = [params[P[f"cl.{k}.weight"]] for k in cws.keys]
            cws.tag = None
        return cws

    def _param_list(self):
        names, params = ["fc.weight", "fc.bias"], [self.fc.weight, self.fc.bias]
        for i, lin in enumerate(self.cl):
            names += [f"cl.{i}.weight", f"cl.{i}.bias"]
            params += [lin.weight, lin.bias]
        for i, b in enumerate(self.bn):
            if b is not None:
                names += [f"bn.{i}.weight", f"bn.{i}.bias"]
                params += [b.weight, b.bias]
        return names, params

    def init_weights(self, W, Fin, Fout):                                      # meshnet.py:64-68
        scale = np.sqrt(2.0 / (Fin + Fout))
        W.uniform_(-scale, scale)
        return W

    def accumulate_grads_in_place(self, enable=True):
        """Opt-in for training loops that own the gradient buffers (optim.FlatAdam / FlatRMSprop): the backward adds
        the weight / bias / BatchNorm gradients straight into the parameters' existing .grad tensors (views of the flat
        gradient buffer) and reports None to autograd, instead of returning ~60 temporaries that autograd then adds
        with one small kernel each.  Same values; `loss.backward()` only -- torch.autograd.grad() on these parameters
        would see no gradient, which is why this is not the default."""
        self._direct_grad = bool(enable)
        return self

    def set_grad_sink(self, sink):
        """sink(list of parameters): called from the backward, on the stream that just received the kernels writing those
        parameters' gradients, when the gradients are accumulated in place (accumulate_grads_in_place) - autograd gets
        None for them, and its post-accumulate hooks only fire once the whole backward has returned.  dist.BucketedAllReduce.notify is the intended sink: the gradient
        all-reduce of a finished bucket then starts under the backward of the coarser levels."""
        self._grad_sink = sink
        return self

    def set_inference(self, real_only=True, perm_reverse=None, nv=None, scale=1.0):
        """Opt-in inference fast path, used in eval() under torch.no_grad() only (the Tester, lib/core/base.py:196;
        the demo, demo/run.py:166).  real_only: compute only the real vertices at the split levels -- the real-vertex
        outputs stay bitwise those of the general path, rows of padding vertices become 0 (no caller reads them: every
        consumer gathers graph_perm_reverse[:nv] first).  perm_reverse / nv [/ scale]: additionally return the mesh in
        mesh-model vertex order, [B, nv, 3] * scale -- `pred_mesh[:, graph_perm_reverse[:nv], :] * 1000`
        (base.py:201-202) folded into the last conv's store."""
        self._infer_real_only = bool(real_only)
        self._out_index = {}
        if perm_reverse is None:
            self._out_perm, self._out_nv, self._out_scale = None, 0, 1.0
        else:
            if not real_only:
                raise ValueError("an output order needs real_only=True")
            self._out_perm = np.asarray(perm_reverse)[:int(nv)].astype(np.int64)
            self._out_nv, self._out_scale = int(nv), float(scale)
        return self

    def _out_index_on(self, device):
        if self._out_perm is None:
            return None
        t = self._out_index.get(device)
        if t is None:
            V0 = int(self.graph_L[0].shape[0])
            inv = np.full((V0,), -1, dtype=np.int32)
            inv[self._out_perm] = np.arange(self._out_nv, dtype=np.int32)
            t = torch.from_numpy(inv).to(device)
            self._out_index[device] = t
        return t

    def forward(self, x):
        _, params = self._param_list()
        keep = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return _MeshNetFn.apply(self, keep, x, *params)


def get_model(num_joint_input_chan, num_mesh_output_chan, graph_L, mano=None):
    """lib/models/meshnet.py:120-123."""
    return Pose2Mesh(num_joint_input_chan, num_mesh_output_chan, graph_L, mano=mano)
