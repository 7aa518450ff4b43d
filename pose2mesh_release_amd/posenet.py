"""PoseNet -- the 2D->3D pose lifter in front of MeshNet (reference: lib/models/posenet.py).

A 4096-wide residual MLP.  Module/parameter names match the reference so `pose_lifter.*` checkpoints load unchanged, and
`pretrained=True` loads `cfg.MODEL.posenet_path` exactly as lib/models/posenet.py:74-75,89-92 does.

On the GPU the forward / backward of the whole lifter is ONE autograd.Function on libp2m_hip.so (round 5; include/p2m.h
"PoseNet"):

  reference op (lib/models/posenet.py)                       here
  --------------------------------------------------------  ---------------------------------------------------------
  nn.Linear w1 / w2 of every block, pre / post (:19,22,59,68)  p2m_gemm_tn: at a batch of a few hundred rows a 4096 x 4096
    and their autograd (dX, dW)                                Linear streams 67 MB of weights against 4 MB of activations -
                                                               a reduction-split contraction with both operands row-major over
                                                               the reduction index (the fc lift's shape, meshnet.py:105); dW
                                                               straight into the parameter's .grad (p2m_gemm_tn_acc)
  BatchNorm1d -> ReLU -> Dropout (:28-30,33-35), bias add,     p2m_pn_stage_fwd / _bwd: one launch per stage - a block owns 32
    residual add (:38), their autograd                         columns and all B rows, so the batch statistics are block-local

P2M_POSENET=stock keeps the stock torch modules (hipBLASLt GEMMs) on the GPU as the A/B form; CPU tensors always take them
(PoseNet on the CPU is plain torch, as in the reference).  Every batch size takes the HIP path (round 6): B < 32 or B % 4 != 0 -
B = 1 of demo/run.py:160 included - is zero-padded, the stage kernels keep the padding out of the statistics (B_real).
"""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .cheby_graph_conv import bn_momentum

HIP_POSENET = os.environ.get("P2M_POSENET", "hip")
if HIP_POSENET not in ("hip", "stock"):
    raise ValueError(f"P2M_POSENET must be hip or stock, not {HIP_POSENET!r}")
_PN_BLOCKS = 512   # blocks of a reduction-split contraction: 2 per CU = one full wave of blocks (8 chunks of 512 rows for
                   # 4096 -> 4096 at a batch of 256: 33 MB of partials instead of 50)
_PAD = 64          # the 2J-wide input and the 3J-wide output are zero-padded to one 64-wide MFMA tile


def _reference_cfg():
    """The reference's global config (lib/core/config.py) when its scripts have loaded it, else None."""
    mod = sys.modules.get("core.config")
    return getattr(mod, "cfg", None) if mod is not None else None


def cfg_posenet_pretrained():
    """cfg.MODEL.posenet_pretrained (lib/models/pose2mesh_net.py:13 reads it at construction); False outside the
    reference's scripts (config.py:56 default)."""
    cfg = _reference_cfg()
    try:
        return bool(cfg.MODEL.posenet_pretrained) if cfg is not None else False
    except AttributeError:
        return False


def load_checkpoint(load_dir, epoch=0, pick_best=False):
    """lib/funcs_utils.py:122-128: the reference's own loader when that module is imported (so a patched loader is
    honoured), else the same behaviour restated: print, torch.load, ValueError("No checkpoint exists!") on any failure.
    map_location is 'cuda' as in the reference when a GPU is there (CPU-only hosts: 'cpu', where the reference fails)."""
    fu = sys.modules.get("funcs_utils")
    if fu is not None and hasattr(fu, "load_checkpoint"):
        return fu.load_checkpoint(load_dir=load_dir, epoch=epoch, pick_best=pick_best)
    try:
        print(f"Fetch model weight from {load_dir}")
        return torch.load(load_dir, map_location="cuda" if torch.cuda.is_available() else "cpu")
    except Exception as e:
        raise ValueError("No checkpoint exists!\n", e)


def _wt(W):
    """[K, N] transposed copy of an nn.Linear weight [N, K] (p2m_weight_pack with one plane)."""
    return ops.weight_pack(W, W.shape[1], 1, need_w2=False)[0]


def _pad_rows(W, rows):
    """W [n, K] -> [rows, K], zero rows appended."""
    out = W.new_zeros((rows, W.shape[1]))
    out[:W.shape[0]] = W.detach()
    return out


class _PoseNetFn(torch.autograd.Function):
    """forward / backward of the whole lifter (lib/models/posenet.py:77-87 with the blocks of :25-38)."""

    @staticmethod
    def forward(ctx, net, keep, x, *params):
        with torch.cuda.device(x.device):
            return _PoseNetFn._forward(ctx, net, keep, x, params)

    @staticmethod
    def _forward(ctx, net, keep, x, params):
        wc = net._weight_cache
        names, _ = net._param_list()
        P = {n: t for n, t in zip(names, params)}
        Br, Fh = x.shape[0], net.linear_size
        # the contractions take B >= 32, B % 4 == 0 (B is a tile dimension of p2m_gemm_tn): any other batch - B = 1 of
        # demo/run.py:160 included - is zero-padded; the stage kernels leave the padding rows out of the statistics and
        # keep them exactly zero (include/p2m.h, B_real)
        B = max(32, (Br + 3) // 4 * 4)
        nin, nout = net.input_size, net.output_size
        training = net.training
        # the dropout probability is read where the reference applies it - each block's own nn.Dropout module
        # (posenet.py:20,29,34) - so that `m.p = 0` on those modules switches it off here as it does there
        p_drop = [float(st.dropout.p) for st in net.linear_stages]
        nst = len(net.linear_stages)
        x = x.contiguous().float()
        if training and Br == 1 and nst > 0:
            # F.batch_norm's own check (posenet.py:28 in train mode with one sample)
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {[Br, Fh]}")

        aux = {}          # operands the backward needs again (MeshNet's train-mode forward bumps ops.WEIGHT_EPOCH in between,
                          # which would make the cache rebuild them: one 67 MB amax pass per weight in f16x2 mode)

        def amax_w(key, W):
            w = wc.get((key, "amax"), W, lambda: ops.param_amax(W))
            aux[key] = w
            return w

        def lin_fwd(a_T, key, W):
            """partials of a W^T: reduction over the input features, A = a^T [K, B], G = W^T [K, N]"""
            Wt = wc.get((key, "t"), W, lambda: _wt(W))
            return ops.gemm_tn([a_T], B, 0, Wt, W.shape[1], W.shape[0], a_amax=ops.amax_of(a_T), g_amax=amax_w(key, W),
                               tname="pn_gemm", target_blocks=_PN_BLOCKS)

        def bn_of(i, sub):
            m = getattr(net.linear_stages[i], sub)
            track = m.track_running_stats and m.running_mean is not None
            use_batch = training or not track
            return (P[f"linear_stages.{i}.{sub}.weight"], P[f"linear_stages.{i}.{sub}.bias"],
                    m.running_mean if track else None, m.running_var if track else None,
                    bn_momentum(m) if (track and use_batch) else 0.1, m.eps, use_batch), m

        rnd = None
        if training and any(p > 0.0 for p in p_drop):
            rnd = torch.rand((2 * nst, B, Fh), device=x.device, dtype=torch.float32)      # one launch for every mask
        nbt = []
        saved = {"x": x, "rnd": rnd, "stages": []}
        # pre-processing Linear (posenet.py:79): the 2J-wide input as one zero-padded 64-wide tile
        xp = F.pad(x, (0, _PAD - nin, 0, B - Br))
        W1 = P["w1.weight"]
        W1t = wc.get(("w1", "tpad"), W1, lambda: _pad_rows(W1.t(), _PAD))                          # W^T, [64, Fh]
        Pp, _, nch = ops.gemm_tn([xp.t().contiguous()], B, 0, W1t, _PAD, Fh, g_amax=amax_w("w1", W1), tname="pn_gemm", target_blocks=_PN_BLOCKS)
        bias = P["w1.bias"]
        resid = None
        y = None
        for i in range(nst):
            bn1, m1 = bn_of(i, "batch_norm1")
            y, a1, a1T, mu1, is1 = ops.pn_stage_fwd(Pp, nch, B, Fh, bias=bias, resid=resid, bn=bn1, B_real=Br,
                                                    rnd=None if rnd is None else rnd[2 * i], p_drop=p_drop[i])
            Wa = P[f"linear_stages.{i}.w1.weight"]
            Pp, _, nch = lin_fwd(a1T, f"s{i}.w1", Wa)
            bn2, m2 = bn_of(i, "batch_norm2")
            z1, a2, a2T, mu2, is2 = ops.pn_stage_fwd(Pp, nch, B, Fh, bias=P[f"linear_stages.{i}.w1.bias"], bn=bn2, B_real=Br,
                                                     rnd=None if rnd is None else rnd[2 * i + 1], p_drop=p_drop[i])
            Wb = P[f"linear_stages.{i}.w2.weight"]
            Pp, _, nch = lin_fwd(a2T, f"s{i}.w2", Wb)
            bias, resid = P[f"linear_stages.{i}.w2.bias"], y
            tap = getattr(net, "_tap", None)
            if tap is not None:
                # test hook (tests/test_gpu_posenet.py): what decides the ReLU masks - the stage input, the statistics used
                # and the affine pair; the kernels' mask is exactly fmaf((z - mean) * invstd, gamma, beta) > 0
                tap.append((2 * i, y, mu1, is1, bn1[0], bn1[1]))
                tap.append((2 * i + 1, z1, mu2, is2, bn2[0], bn2[1]))
            if training:
                for m in (m1, m2):
                    if m.track_running_stats and m.num_batches_tracked is not None:
                        nbt.append(m.num_batches_tracked)
            if keep:
                saved["stages"].append((y, a1, mu1, is1, z1, a2, mu2, is2, bn1[6], bn2[6]))
            del a1T, a2T
        # the last block's output (its own z + residual), transposed for the post-processing Linear (posenet.py:85)
        y_last, _, yT, _, _ = ops.pn_stage_fwd(Pp, nch, B, Fh, bias=bias, resid=resid, want_a=False, B_real=Br)
        W2 = P["w2.weight"]
        W2p = wc.get(("w2", "pad"), W2, lambda: _pad_rows(W2, _PAD))                               # [64, Fh]
        W2t = wc.get(("w2", "tpad"), W2, lambda: _wt(W2p))                                         # [Fh, 64]
        aux["W2p"] = W2p
        Pp, _, nch = ops.gemm_tn([yT], B, 0, W2t, Fh, _PAD, a_amax=ops.amax_of(yT), g_amax=amax_w("w2", W2),
                                 tname="pn_gemm", target_blocks=_PN_BLOCKS)
        b2 = wc.get(("w2", "bpad"), P["w2.bias"], lambda: F.pad(P["w2.bias"].detach(), (0, _PAD - nout)))
        out64, _, _, _, _ = ops.pn_stage_fwd(Pp, nch, B, _PAD, bias=b2, want_a=False, want_aT=False, B_real=Br)
        if nbt:
            torch._foreach_add_(nbt, 1)           # nn.BatchNorm1d.forward's `num_batches_tracked += 1`, one launch
            ops.bump_weight_epoch()               # running statistics moved behind torch's back (cached eval operands)
        saved["y_last"], saved["aux"] = y_last, aux
        ctx.net, ctx.saved, ctx.params, ctx.names = net, (saved if keep else None), params, names
        ctx.training, ctx.p_drop, ctx.B, ctx.Br, ctx.ran = training, p_drop, B, Br, False
        return out64[:Br, :nout].contiguous()

    @staticmethod
    def backward(ctx, g_out):
        with torch.cuda.device(g_out.device):
            return _PoseNetFn._backward(ctx, g_out)

    @staticmethod
    def _backward(ctx, g_out):
        net, saved, params, names = ctx.net, ctx.saved, ctx.params, ctx.names
        if saved is None:
            if ctx.ran:
                raise ops.P2MError("PoseNet: backward called a second time - the saved activations are freed by the first "
                                   "backward (retain_graph is not supported on this path)")
            raise ops.P2MError("backward called but the forward ran without gradient tracking")
        ctx.ran = True
        wc = net._weight_cache
        P = {n: t for n, t in zip(names, params)}
        idx = {n: i for i, n in enumerate(names)}
        B, Br, Fh = ctx.B, ctx.Br, net.linear_size
        nin, nout = net.input_size, net.output_size
        nst = len(net.linear_stages)
        rnd, p_drop = saved["rnd"], ctx.p_drop
        # parameters that take a gradient (frozen ones: no contraction, no reduction, None to autograd)
        needs = {n: bool(ctx.needs_input_grad[3 + i]) for i, n in enumerate(names)}
        dev = g_out.device
        grads = [None] * len(params)
        direct = net._direct_grad and all(
            p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.device == dev
            for p in params if p.requires_grad)

        aux = saved["aux"]

        def amax_w(key, W):
            return aux[key] if key in aux else wc.get((key, "amax"), W, lambda: ops.param_amax(W))

        def target(name, shape=None):
            """where the gradient of `name` is written: its .grad (accumulated into) or a fresh zero tensor; None for a
            parameter that takes no gradient"""
            if not needs[name]:
                return None
            if direct:
                return P[name].grad
            t = torch.zeros_like(P[name])
            grads[idx[name]] = t
            return t

        def ready(*ns):
            if direct and net._grad_sink is not None:
                net._grad_sink([P[n] for n in ns if needs[n]])

        g = g_out.contiguous().float()
        g64 = F.pad(g, (0, _PAD - nout, 0, B - Br))                        # [B, 64]
        g64T = g64.t().contiguous()                                        # [64, B]
        y_last = saved["y_last"]
        # post-processing Linear (posenet.py:85): dW = g^T y, db = sum g, g_y = g W
        W2 = P["w2.weight"]
        if needs["w2.weight"]:
            Pw, _, _ = ops.gemm_tn([g64], _PAD, 0, y_last, B, Fh, tname="pn_gemm", target_blocks=_PN_BLOCKS)           # [1, 64, Fh]: one chunk
            target("w2.weight").add_(Pw.sum(0)[:nout] if Pw.shape[0] > 1 else Pw[0, :nout])
        if needs["w2.bias"]:
            target("w2.bias").add_(g.sum(0))
        ready("w2.weight", "w2.bias")
        W2p = aux["W2p"]                                                                       # [64, Fh]
        Pg, _, nchg = ops.gemm_tn([g64T], B, 0, W2p, _PAD, Fh, g_amax=amax_w("w2", W2), tname="pn_gemm", target_blocks=_PN_BLOCKS)
        gz2 = gz2T = None
        for i in reversed(range(nst)):
            y, a1, mu1, is1, z1, a2, mu2, is2, tr1, tr2 = saved["stages"][i]
            pre = f"linear_stages.{i}."
            if gz2 is None:
                # gradient w.r.t. the last block's output: the partials summed, row-major and transposed; its row sums
                # are the bias gradient of this block's w2
                gz2, gz2T = ops.pn_stage_bwd(Pg, nchg, B, Fh, dbias=target(pre + "w2.bias"), accumulate=True, B_real=Br)
            Wb = P[pre + "w2.weight"]
            if needs[pre + "w2.weight"]:
                ops.gemm_tn_acc(gz2, a2, target(pre + "w2.weight"))                           # dW2 += g_z2^T a2
            ready(pre + "w2.weight", pre + "w2.bias")
            Pa, _, nch = ops.gemm_tn([gz2T], B, 0, Wb, Fh, Fh, a_amax=ops.amax_of(gz2T), g_amax=amax_w(f"s{i}.w2", Wb),
                                     tname="pn_gemm", target_blocks=_PN_BLOCKS)
            gz1, gz1T = ops.pn_stage_bwd(Pa, nch, B, Fh, bn=(z1, mu2, is2, P[pre + "batch_norm2.weight"],
                                                            P[pre + "batch_norm2.bias"], tr2),
                                         rnd=None if rnd is None else rnd[2 * i + 1], p_drop=p_drop[i],
                                         dgamma=target(pre + "batch_norm2.weight"), dbeta=target(pre + "batch_norm2.bias"),
                                         dbias=target(pre + "w1.bias"), accumulate=True, B_real=Br)
            Wa = P[pre + "w1.weight"]
            if needs[pre + "w1.weight"]:
                ops.gemm_tn_acc(gz1, a1, target(pre + "w1.weight"))                           # dW1 += g_z1^T a1
            ready(pre + "batch_norm2.weight", pre + "batch_norm2.bias", pre + "w1.weight", pre + "w1.bias")
            Pa, _, nch = ops.gemm_tn([gz1T], B, 0, Wa, Fh, Fh, a_amax=ops.amax_of(gz1T), g_amax=amax_w(f"s{i}.w1", Wa),
                                     tname="pn_gemm", target_blocks=_PN_BLOCKS)
            prev_bias = f"linear_stages.{i - 1}.w2.bias" if i > 0 else "w1.bias"
            gy, gyT = ops.pn_stage_bwd(Pa, nch, B, Fh, addend=gz2, bn=(y, mu1, is1, P[pre + "batch_norm1.weight"],
                                                                       P[pre + "batch_norm1.bias"], tr1),
                                       rnd=None if rnd is None else rnd[2 * i], p_drop=p_drop[i],
                                       dgamma=target(pre + "batch_norm1.weight"), dbeta=target(pre + "batch_norm1.bias"),
                                       dbias=target(prev_bias), accumulate=True, B_real=Br,
                                       want_T=(i > 0 or ctx.needs_input_grad[2]))
            ready(pre + "batch_norm1.weight", pre + "batch_norm1.bias")
            gz2, gz2T = gy, gyT
            saved["stages"][i] = None
        # pre-processing Linear (posenet.py:79): dW = g_y0^T x (the input zero-padded to 64 columns)
        if needs["w1.weight"]:
            xp = F.pad(saved["x"], (0, _PAD - nin, 0, B - Br))
            Pw, _, nchw = ops.gemm_tn([gz2], Fh, 0, xp, B, _PAD, a_amax=ops.amax_of(gz2), tname="pn_gemm", target_blocks=_PN_BLOCKS)   # [1, Fh, 64]
            target("w1.weight").add_((Pw[0] if nchw == 1 else Pw.sum(0))[:, :nin])
        ready("w1.weight", "w1.bias")
        gx = None
        if ctx.needs_input_grad[2]:
            W1 = P["w1.weight"]
            W1p = wc.get(("w1", "pad"), W1, lambda: F.pad(W1.detach(), (0, _PAD - nin)))       # [Fh, 64]
            Px, _, nchx = ops.gemm_tn([gz2T], B, 0, W1p, Fh, _PAD, a_amax=ops.amax_of(gz2T), g_amax=amax_w("w1", W1),
                                      tname="pn_gemm", target_blocks=_PN_BLOCKS)
            gx = (Px[0] if nchx == 1 else Px.sum(0))[:Br, :nin].contiguous()
        ctx.saved = None
        return (None, None, gx) + tuple(grads)


class Linear(nn.Module):
    """Pre-activation residual block: BN-ReLU-Dropout-Linear twice (posenet.py:11-38)."""

    def __init__(self, linear_size, p_dropout=0.5):
        super().__init__()
        self.l_size = linear_size
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(p_dropout)
        self.w1 = nn.Linear(linear_size, linear_size)
        self.batch_norm1 = nn.BatchNorm1d(linear_size)
        self.w2 = nn.Linear(linear_size, linear_size)
        self.batch_norm2 = nn.BatchNorm1d(linear_size)

    def forward(self, x):
        y = self.w1(self.dropout(self.relu(self.batch_norm1(x))))
        y = self.w2(self.dropout(self.relu(self.batch_norm2(y))))
        return x + y


class LinearModel(nn.Module):
    """posenet.py:41-92.  As in the reference, the outer batch_norm1/relu/dropout are created (their
    parameters are part of the state dict) but never applied in forward (posenet.py:77-87)."""

    def __init__(self, num_joint, linear_size=4096, num_stage=2, p_dropout=0.5, pretrained=False, posenet_path=None):
        super().__init__()
        self.linear_size, self.p_dropout, self.num_stage = linear_size, p_dropout, num_stage
        self.input_size, self.output_size = num_joint * 2, num_joint * 3
        self.w1 = nn.Linear(self.input_size, linear_size)
        self.batch_norm1 = nn.BatchNorm1d(linear_size)
        self.linear_stages = nn.ModuleList([Linear(linear_size, p_dropout) for _ in range(num_stage)])
        self.w2 = nn.Linear(linear_size, self.output_size)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(p_dropout)
        self._weight_cache = ops.WeightCache()
        self._direct_grad = False
        self._grad_sink = None
        if pretrained:
            self._load_pretrained_model(posenet_path)

    def forward(self, x):
        B = x.shape[0]
        if x.is_cuda and HIP_POSENET == "hip" and x.dtype == torch.float32 and B >= 1 \
                and self.linear_size % 128 == 0 and self.input_size <= _PAD and self.output_size <= _PAD:
            names, params = self._param_list()
            keep = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
            return _PoseNetFn.apply(self, keep, x, *params)
        y = self.w1(x)
        for stage in self.linear_stages:
            y = stage(y)
        return self.w2(y)

    # ---- HIP path ------------------------------------------------------------------------------------------------------
    def _param_list(self):
        """Parameters the forward uses, in a fixed order (the outer batch_norm1 is never applied, posenet.py:77-87)."""
        names, params = ["w1.weight", "w1.bias"], [self.w1.weight, self.w1.bias]
        for i, st in enumerate(self.linear_stages):
            for sub in ("batch_norm1", "w1", "batch_norm2", "w2"):
                m = getattr(st, sub)
                names += [f"linear_stages.{i}.{sub}.weight", f"linear_stages.{i}.{sub}.bias"]
                params += [m.weight, m.bias]
        names += ["w2.weight", "w2.bias"]
        params += [self.w2.weight, self.w2.bias]
        return names, params

    def accumulate_grads_in_place(self, enable=True):
        """Opt-in (as Pose2Mesh.accumulate_grads_in_place): the HIP backward adds every gradient straight into the
        parameters' existing .grad tensors (views of optim.FlatAdam's flat buffer) and reports None to autograd - the
        weight gradients come out of the contraction's epilogue, no 67 MB temporaries, no accumulate pass.
        `loss.backward()` only."""
        self._direct_grad = bool(enable)
        return self

    def set_grad_sink(self, sink):
        """sink(list of parameters): called from the backward right after the kernels writing those parameters' gradients
        are enqueued (in-place accumulation; dist.BucketedAllReduce.notify)."""
        self._grad_sink = sink
        return self

    def _load_pretrained_model(self, posenet_path=None):
        """lib/models/posenet.py:89-92: checkpoint['model_state_dict'] of cfg.MODEL.posenet_path (strict load)."""
        print("Loading pretrained posenet...")
        if posenet_path is None:
            cfg = _reference_cfg()
            if cfg is None:
                raise ValueError("pretrained=True needs cfg.MODEL.posenet_path (the reference's core.config is not "
                                 "loaded) or an explicit posenet_path=")
            posenet_path = cfg.MODEL.posenet_path
        checkpoint = load_checkpoint(load_dir=posenet_path, pick_best=True)
        self.load_state_dict(checkpoint["model_state_dict"])


def get_model(num_joint, hid_dim, num_layer, p_dropout, pretrained=False, posenet_path=None):
    """lib/models/posenet.py:95-98 (+ an explicit path for callers without the reference's cfg)."""
    return LinearModel(num_joint, hid_dim, num_layer, p_dropout, pretrained, posenet_path)
