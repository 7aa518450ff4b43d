"""-m gpu: the hand-written PoseNet path (lib/models/posenet.py:11-92) - every Linear on p2m_gemm_tn /
p2m_gemm_tn_acc, BatchNorm1d + ReLU + dropout + bias + residual in p2m_pn_stage_fwd / _bwd - against the oracle
(oracle/meshnet_oracle.posenet_forward: the reference's operator sequence on the CPU, fp32 and float64), in eval() and in
train() (batch statistics; dropout off, or on with the SAME mask fed to a float64 restatement), forward, every parameter
gradient, running statistics; against the stock torch modules on the GPU (the P2M_POSENET=stock A/B form); and the
in-place gradient accumulation used with the flat optimizers."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import helpers
import meshnet_oracle as mo

pytestmark = pytest.mark.gpu


def _model(J, seed=3, p_dropout=0.5, hid=4096):
    from pose2mesh_release_amd import posenet
    torch.manual_seed(seed)
    net = posenet.get_model(J, hid, 2, p_dropout)
    sd = helpers.numpy_state(net.state_dict(), seed)          # random weights AND random running statistics
    net.load_state_dict(sd)
    return net, {k: v.clone() for k, v in sd.items()}


class _MaskedReLU(torch.autograd.Function):
    """relu with a GIVEN mask (tests/kinks.py: the mask the implementation under test used), forward and backward."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x * mask

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


def _hip_relu_masks(tap, Br):
    """The masks the stage kernels used, reproduced exactly: xhat = (z - mean) * invstd in fp32 (two roundings, as in
    k_pn_stage_fwd / _bwd), then the sign of fmaf(xhat, gamma, beta) - one rounding never changes a sign, so the product and
    sum are formed exactly in float64."""
    out = []
    for _, z, mu, istd, ga, be in sorted(tap, key=lambda t: t[0]):
        xh = (z[:Br] - mu) * istd
        out.append(((xh.double() * ga.double() + be.double()) > 0).cpu())
    return out


def _ref(sd, x, training, masks=None, p=0.0, dtype=torch.float64, relu_masks=None, kinks=None):
    """posenet.py:77-87 + :25-38 restated with optional dropout masks (uniform numbers: keep where u >= p, scale 1/(1-p)).
    relu_masks: one bool tensor per BatchNorm-ReLU (stage order) to use INSTEAD of the restatement's own y > 0; `kinks`
    (a dict) then receives how many elements differ and the largest |pre-activation| among them."""
    calls = [0]
    sd = {k: (v.to(dtype).clone() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k and not k.startswith("batch_norm1")]
    for k in names:
        sd[k].requires_grad_(True)
    x = x.to(dtype)

    def lin(n, t):
        return F.linear(t, sd[n + ".weight"], sd[n + ".bias"])

    def stage(n, t, u):
        t = F.batch_norm(t, sd[n + ".running_mean"], sd[n + ".running_var"], sd[n + ".weight"], sd[n + ".bias"],
                         training, 0.1, 1e-5)
        if relu_masks is None:
            t = F.relu(t)
        else:
            m = relu_masks[calls[0]]
            diff = m != (t.detach() > 0)
            if kinks is not None:
                kinks["flips"] = kinks.get("flips", 0) + int(diff.sum())
                kinks["elements"] = kinks.get("elements", 0) + t.numel()
                if diff.any():
                    kinks["max_abs"] = max(kinks.get("max_abs", 0.0), float(t.detach().abs()[diff].max()))
            t = _MaskedReLU.apply(t, m.to(dtype))
        calls[0] += 1
        if u is not None:
            t = torch.where(u.to(dtype) >= p, t / (1.0 - p), torch.zeros_like(t))
        return t
    y = lin("w1", x)
    s = 0
    while f"linear_stages.{s}.w1.weight" in sd:
        q = f"linear_stages.{s}."
        z = lin(q + "w1", stage(q + "batch_norm1", y, None if masks is None else masks[2 * s]))
        z = lin(q + "w2", stage(q + "batch_norm2", z, None if masks is None else masks[2 * s + 1]))
        y = y + z
        s += 1
    return lin("w2", y), sd, names


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


@pytest.fixture(params=["f16x2", "bf16x3", "f32"])
def arith(request, hip_libs, monkeypatch):
    from pose2mesh_release_amd import ops
    monkeypatch.setattr(ops, "GEMM_ARITH", request.param)
    ops.bump_weight_epoch()
    yield request.param
    ops.bump_weight_epoch()


@pytest.mark.parametrize("J,B", [(17, 32), (19, 64), (21, 36)])
def test_posenet_eval_forward_and_gradients_vs_oracle(arith, J, B):
    from pose2mesh_release_amd import synth
    net, sd = _model(J)
    net = net.cuda().eval()
    x = synth.pose2d_batch(B, J, seed=5).reshape(B, -1)
    w = torch.randn(B, 3 * J, generator=torch.Generator().manual_seed(9))
    out = net(x.cuda())
    assert out.shape == (B, 3 * J)
    (out * w.cuda()).sum().backward()
    ref32 = mo.posenet_forward({k: v.clone() for k, v in sd.items()}, x, False)          # the reference's fp32 CPU path
    ref, sd64, names = _ref(sd, x, False)
    (ref * w.double()).sum().backward()
    ref = ref.detach()
    scale = float(ref.abs().max())
    assert float((out.detach().cpu().double() - ref).abs().max()) <= 2e-5 * max(1.0, scale)
    assert float((out.detach().cpu() - ref32).abs().max()) <= 1e-4 * max(1.0, scale)
    got = dict(net.named_parameters())
    for k in names:
        assert _rel(got[k].grad, sd64[k].grad) <= 5e-5, k
    assert got["batch_norm1.weight"].grad is None                    # never applied in forward (posenet.py:77-87)


ALIGNED_GRAD_TOL = 5e-5      # every gradient tensor vs float64 with the ReLU masks aligned (was 2e-3 unaligned)


def _check_train(net, sd, x, w, Br, masks=None, p=0.0, seed=None, loose=1.0):
    """One train-mode forward + backward of the HIP path with the ReLU masks tapped; the float64 restatement runs with THOSE
    masks.  Returns the kink statistics.  loose: factor on the tolerances for batches of a handful of samples, where
    BatchNorm1d itself is ill-conditioned (a channel on which the 2-5 samples nearly agree is divided by sqrt(var + 1e-5):
    fp32 round-off of z is amplified up to 316 x - in the reference's own fp32 arithmetic just the same)."""
    net._tap = []
    if seed is not None:
        torch.manual_seed(seed)
    out = net(x.cuda())
    (out * w.cuda()).sum().backward()
    relu_masks = _hip_relu_masks(net._tap, Br)
    net._tap = None
    kinks = {}
    ref, sd64, names = _ref(sd, x, True, masks=masks, p=p, relu_masks=relu_masks, kinks=kinks)
    (ref * w.double()).sum().backward()
    scale = float(ref.detach().abs().max())
    assert out.shape == ref.shape
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= loose * 5e-5 * max(1.0, scale)
    # every mask difference is a genuine kink element (pre-activations are O(1)), and there are only a handful
    assert kinks.get("max_abs", 0.0) <= loose * 1e-4, kinks
    assert kinks.get("flips", 0) <= max(4, 1e-5 * kinks["elements"]), kinks
    got = dict(net.named_parameters())
    errs = {}
    for k in names:
        if k.endswith("w1.bias") and k.startswith("linear_stages"):
            # a bias in front of a train-mode BatchNorm: the true gradient is exactly 0 (round-off only)
            assert float(got[k].grad.norm()) <= 1e-3 * float(got[k.replace("bias", "weight")].grad.norm()), k
            continue
        errs[k] = _rel(got[k].grad, sd64[k].grad)
    bad = {k: v for k, v in errs.items() if v > loose * ALIGNED_GRAD_TOL}
    assert not bad, (bad, errs, kinks)
    return kinks, errs


@pytest.mark.parametrize("J,B", [(19, 64), (21, 256), (17, 2), (19, 5), (21, 31), (17, 38)])
def test_posenet_train_batch_statistics_vs_oracle(arith, J, B):
    """train(): BatchNorm1d on batch statistics (dropout off, so that the oracle is deterministic): forward, running
    statistics, num_batches_tracked, every gradient against the float64 restatement run with the ReLU masks the kernels used
    (tests/kinks.py's method: an element within fp32 rounding of 0 may take either mask; with the masks aligned every gradient
    agrees to fp32 round-off).  B = 2, 5, 31, 38: the zero-padded batches (B < 32 or B % 4 != 0; round 6) - statistics over
    the real rows only."""
    from pose2mesh_release_amd import synth
    net, sd = _model(J, p_dropout=0.0)
    net = net.cuda().train()
    x = synth.pose2d_batch(B, J, seed=6).reshape(B, -1)
    w = torch.randn(B, 3 * J, generator=torch.Generator().manual_seed(10))
    # B = 2: BatchNorm over two samples maps every channel to +-1 whatever the input - the true gradient through it is pure
    # cancellation (non-zero only through eps); B = 5: channels with nearly equal samples are divided by sqrt(var + 1e-5)
    _check_train(net, sd, x, w, B, loose=1.0 if B >= 30 else (20.0 if B >= 5 else 200.0))
    sd32 = {k: v.clone() for k, v in sd.items()}
    mo.posenet_forward(sd32, x, True)                                 # fp32 oracle: updates its running statistics in place
    now = net.state_dict()
    for k, v in sd32.items():
        if "running" in k:
            assert float((now[k].cpu() - v).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max())), k
        if k.endswith("num_batches_tracked") and k.startswith("linear_stages"):
            assert int(now[k]) == int(sd[k]) + 1, k


@pytest.mark.parametrize("B", [1, 2, 5, 31])
def test_posenet_small_and_ragged_batches_take_the_hip_path(arith, B):
    """B = 1 (demo/run.py:160), B < 32 and B % 4 != 0 used to fall back to the stock modules.  eval():
    forward and every gradient vs float64; the launches are pn_gemm contractions (no library GEMM); train() with one sample
    raises like F.batch_norm does in the reference."""
    from pose2mesh_release_amd import ops, synth
    J = 17
    net, sd = _model(J)
    net = net.cuda().eval()
    x = synth.pose2d_batch(B, J, seed=15).reshape(B, -1)
    w = torch.randn(B, 3 * J, generator=torch.Generator().manual_seed(16))
    xg = x.cuda().requires_grad_(True)
    ops.TIMER = ops.KernelTimer()
    try:
        out = net(xg)
        (out * w.cuda()).sum().backward()
        summ = ops.TIMER.summary()
    finally:
        ops.TIMER = None
    assert summ["pn_gemm"]["launches"] == 6 + 12, summ               # (+1: the input gradient)
    assert out.shape == (B, 3 * J) and xg.grad.shape == x.shape
    x64 = x.double().requires_grad_(True)
    ref, sd64, names = _ref(sd, x64, False)
    (ref * w.double()).sum().backward()
    ref = ref.detach()
    scale = float(ref.abs().max())
    assert float((out.detach().cpu().double() - ref).abs().max()) <= 2e-5 * max(1.0, scale)
    got = dict(net.named_parameters())
    for k in names:
        assert _rel(got[k].grad, sd64[k].grad) <= 5e-5, k
    assert _rel(xg.grad, x64.grad) <= 5e-5
    if B == 1:
        with pytest.raises(ValueError, match="more than 1 value per channel"):
            net.train()(x.cuda())


def test_posenet_frozen_parameters_and_repeated_backward(hip_libs):
    """frozen parameters with accumulate_grads_in_place() used to crash the backward (None.add_); now they
    are skipped - no contraction, no reduction, None to autograd - and the others are unchanged.  A second backward on the
    same graph says what happened."""
    from pose2mesh_release_amd import ops, synth
    J, B = 17, 32
    x = synth.pose2d_batch(B, J, seed=17).reshape(B, -1).cuda()
    w = torch.randn(B, 3 * J, generator=torch.Generator().manual_seed(18)).cuda()
    res = {}
    frozen = ("w1.weight", "linear_stages.0.w2.weight", "linear_stages.1.batch_norm1.weight", "linear_stages.1.w1.bias", "w2.bias")
    for mode in ("all", "frozen", "frozen_in_place"):
        net, _ = _model(J, p_dropout=0.0)
        net = net.cuda().train()
        P = dict(net.named_parameters())
        if mode != "all":
            for k in frozen:
                P[k].requires_grad_(False)
        if mode == "frozen_in_place":
            net.accumulate_grads_in_place(True)
            for k, p in P.items():
                if p.requires_grad:
                    p.grad = torch.zeros_like(p)
        ops.TIMER = ops.KernelTimer()
        try:
            out = net(x)
            loss = (out * w).sum()
            loss.backward()
            launches = ops.TIMER.summary()["pn_gemm"]["launches"]
        finally:
            ops.TIMER = None
        res[mode] = ({k: (None if p.grad is None else p.grad.clone()) for k, p in P.items()}, launches)
        if mode == "frozen":
            with pytest.raises(Exception, match="second time|retain_graph|freed"):
                loss.backward()
    assert res["frozen"][1] == res["all"][1] - 2          # the dW contractions of the two frozen weight matrices are not issued
    for mode in ("frozen", "frozen_in_place"):
        for k, g in res["all"][0].items():
            if k in frozen:
                assert res[mode][0][k] is None, (mode, k)
            elif g is not None:
                assert torch.equal(res[mode][0][k], g), (mode, k)


@pytest.mark.parametrize("B", [64, 30])
def test_posenet_train_with_dropout_vs_reference_given_the_same_mask(arith, B):
    """train() with p = 0.5: the HIP path draws its uniform numbers with ONE torch.rand of shape [2 * stages, B padded, 4096]
    on the device; re-seeding the device generator reproduces them, and the float64 restatement is run with those dropout
    masks and the ReLU masks the kernels used."""
    from pose2mesh_release_amd import synth
    J = 19
    Bp = max(32, (B + 3) // 4 * 4)
    net, sd = _model(J, p_dropout=0.5)
    net = net.cuda().train()
    x = synth.pose2d_batch(B, J, seed=7).reshape(B, -1)
    w = torch.randn(B, 3 * J, generator=torch.Generator().manual_seed(11))
    torch.manual_seed(1234)
    rnd = torch.rand((4, Bp, 4096), device="cuda").cpu()[:, :B]
    keep = float((rnd >= 0.5).float().mean())
    assert 0.49 < keep < 0.51
    _check_train(net, sd, x, w, B, masks=[rnd[i] for i in range(4)], p=0.5, seed=1234)
    # a second forward draws a different mask
    torch.manual_seed(1234)
    out = net(x.cuda())
    out2 = net(x.cuda())
    assert not torch.equal(out2, out)


def test_posenet_hip_vs_stock_modules_and_in_place_gradients(hip_libs, monkeypatch):
    """The A/B form (P2M_POSENET=stock: nn.Linear / BatchNorm1d -> hipBLASLt) on the same GPU, train mode without dropout;
    and accumulate_grads_in_place(): the gradients land in the parameters' existing .grad tensors (+=), autograd sees None."""
    from pose2mesh_release_amd import posenet, synth
    J, B = 17, 64
    x = synth.pose2d_batch(B, J, seed=8).reshape(B, -1).cuda()
    w = torch.randn(B, 3 * J, generator=torch.Generator().manual_seed(12)).cuda()
    res = {}
    for mode in ("hip", "stock", "hip_in_place"):
        monkeypatch.setattr(posenet, "HIP_POSENET", "stock" if mode == "stock" else "hip")
        net, _ = _model(J, p_dropout=0.0)
        net = net.cuda().train()
        pre = {}
        if mode == "hip_in_place":
            net.accumulate_grads_in_place(True)
            for k, p in net.named_parameters():
                p.grad = torch.full_like(p, 0.25)                      # something to accumulate INTO
                pre[k] = 0.25
        out = net(x)
        (out * w).sum().backward()
        res[mode] = (out.detach(), {k: (p.grad - pre.get(k, 0.0)) for k, p in net.named_parameters() if p.grad is not None},
                     {k: v.clone() for k, v in net.state_dict().items() if "running" in k})
    for other in ("stock", "hip_in_place"):
        assert float((res["hip"][0] - res[other][0]).abs().max()) <= 2e-5 * max(1.0, float(res["hip"][0].abs().max()))
        for k, g in res["hip"][1].items():
            if k.startswith("batch_norm1"):
                continue
            if k.endswith("w1.bias") and k.startswith("linear_stages"):
                continue                                               # exactly-zero gradients (round-off on both sides)
            assert _rel(res[other][1][k], g) <= (2e-3 if other == "stock" else 1e-6), (other, k)   # (stock: ReLU kinks)
        for k, v in res["hip"][2].items():
            assert float((res[other][2][k] - v).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max())), (other, k)


def test_posenet_train_step_has_no_library_gemm(hip_libs):
    """The HIP path issues no hipBLASLt / rocBLAS kernel: the step's launches are p2m kernels and small torch copies only
    (checked through the package's own launch timer: every PoseNet contraction is a `pn_gemm` launch)."""
    from pose2mesh_release_amd import ops, synth
    J, B = 19, 64
    net, _ = _model(J, p_dropout=0.0)
    net = net.cuda().train()
    x = synth.pose2d_batch(B, J, seed=9).reshape(B, -1).cuda()
    ops.TIMER = ops.KernelTimer()
    try:
        net(x).sum().backward()
        summ = ops.TIMER.summary()
    finally:
        ops.TIMER = None
    # 6 forward contractions + (2 + 2 x 4 + 1) backward ones
    assert summ["pn_gemm"]["launches"] == 6 + 11, summ
