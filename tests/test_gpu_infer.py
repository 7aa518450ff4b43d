"""-m gpu: the inference fast path (VERDICT r1 "next round" 8): real vertices only in eval(),
BatchNorm + ReLU fused into the contraction epilogue, mesh-order store in the last conv, hipGraph replay -- all pinned to
the general drop-in path BITWISE on the real vertices, and to the oracle's Tester epilogue."""
import numpy as np
import pytest
import torch

import helpers
import loss_oracle as lo
import meshnet_oracle as mo

pytestmark = pytest.mark.gpu


def _flat(joint_set, seed=2):
    from pose2mesh_release_amd import pose2mesh_net
    gL, _, rev = helpers.golden_graphs(joint_set)
    J = int(gL[-1].shape[0])
    net = pose2mesh_net.get_model(J, gL, mano=(joint_set == "mano"))
    sd = helpers.numpy_state(net.state_dict(), seed)
    net.load_state_dict(sd)
    return net.cuda().eval(), sd, gL, np.asarray(rev), J


def test_fused_activation_epilogue_is_bitwise_the_separate_pass(hip_libs):
    """p2m_gemm_planes(_rows) with act_scale/act_shift/act_relu == contraction, then p2m_bn_act_fwd (MFMA and scalar paths)."""
    from pose2mesh_release_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    for M, Ka, N, planes in ((1000, 64, 128, 3), (517, 128, 64, 1), (300, 5, 32, 3)):
        A = [torch.randn(M, Ka, device="cuda", generator=g) for _ in range(planes)]
        W = torch.randn(planes * Ka, N, device="cuda", generator=g) * 0.1
        bias = torch.randn(N, device="cuda", generator=g)
        sc = torch.rand(N, device="cuda", generator=g) + 0.5
        sh = torch.randn(N, device="cuda", generator=g)
        (y,), _ = ops.gemm_planes(A, Ka, 0, W, bias, M, N)
        co = torch.stack([sc, sc, sc, sh])
        ref = ops.bn_act_fwd(y, co, True, None, 0, 0, M, N)
        (fused,), _ = ops.gemm_planes(A, Ka, 0, W, bias, M, N, act=(sc, sh, True))
        assert torch.equal(fused, ref)
        (noact,), _ = ops.gemm_planes(A, Ka, 0, W, bias, M, N, act=(sc, sh, False))
        assert torch.equal(noact, ops.bn_act_fwd(y, co, False, None, 0, 0, M, N))


@pytest.mark.parametrize("joint_set,B", [("mano", 4), ("human36", 3)])
def test_real_only_eval_is_bitwise_the_general_path_on_real_vertices(hip_libs, joint_set, B):
    from pose2mesh_release_amd import synth
    net, sd, gL, rev, J = _flat(joint_set)
    nv = 778 if joint_set == "mano" else 6890
    x = synth.pose2d_batch(B, J, seed=5).cuda()
    with torch.no_grad():
        ref_mesh, ref_pose = net(x)
        net.set_inference(real_only=True)
        fast_mesh, fast_pose = net(x)
        net.set_inference(real_only=True, perm_reverse=rev, nv=nv, scale=1000.0)
        ordered, _ = net(x)
        net.set_inference(real_only=False)
        again, _ = net(x)
    real = torch.as_tensor(rev[:nv], device="cuda")
    assert torch.equal(again, ref_mesh) and torch.equal(fast_pose, ref_pose)
    assert torch.equal(fast_mesh[:, real], ref_mesh[:, real])                # bitwise on the real vertices
    fake = np.setdiff1d(np.arange(gL[0].shape[0]), rev[:nv])
    assert float(fast_mesh[:, torch.as_tensor(fake, device="cuda")].abs().max()) == 0.0
    assert ordered.shape == (B, nv, 3) and torch.equal(ordered, ref_mesh[:, real] * 1000.0)
    # gradients still work through the general path while the flag is set (training / fine-tuning is unaffected)
    net.set_inference(real_only=True)
    xg = x.clone().requires_grad_(True)
    m, _ = net(xg)
    assert torch.equal(m, ref_mesh)
    m.sum().backward()
    assert xg.grad is None or torch.isfinite(xg.grad).all()


def test_graphed_inference_vs_general_path_and_oracle(hip_libs):
    """pose2mesh_release_amd.infer.GraphedInference (hipGraph replay) == general module + MeshEpilogue, bit for bit,
    across replays with different inputs; and == the oracle's Tester epilogue (lib/core/base.py:200-204) to 1e-4."""
    from pose2mesh_release_amd import infer, loss as L, synth
    net, sd, gL, rev, J = _flat("human36")
    nv, B = 6890, 4
    jreg = synth.synthetic_regressor(J, nv)
    epi = L.MeshEpilogue(rev, nv, jreg, scale=1000.0)
    xs = [synth.pose2d_batch(B, J, seed=s).cuda() for s in (7, 8, 7)]
    with torch.no_grad():
        refs = []
        for x in xs:
            cam, pose3d = net(x)
            refs.append((*epi(cam), pose3d))
    step = infer.GraphedInference(net, rev, nv, jreg, B, scale=1000.0)
    assert step.graph is not None
    for x, (rm, rj, rp) in zip(xs, refs):
        mesh, joints, pose3d = step(x)
        torch.cuda.synchronize()
        assert torch.equal(mesh, rm) and torch.equal(joints, rj) and torch.equal(pose3d, rp)
    torch.set_num_threads(16)
    with torch.no_grad():
        cam_o, _ = mo.flat_forward(sd, helpers.oracle_graphs(gL), xs[2].cpu(), False, False)
    om, oj = lo.test_epilogue(cam_o, rev, nv, torch.from_numpy(jreg))
    mesh, joints, _ = step(xs[2])
    assert helpers.max_vertex_l2(mesh.cpu() / 1000.0, om / 1000.0) <= 1e-4
    assert (joints.cpu() - oj).abs().max() <= 1e-1                      # millimetres (1e-4 m)
    net.set_inference(real_only=False)


def test_graphed_inference_recaptures_when_the_weights_change(hip_libs):
    """the captured graph reads derived weight operands by raw pointer.  After the weights change
    (here: load_state_dict of other values) the next call must not replay the old operands: it re-captures, and the
    result is the general path's result for the NEW weights."""
    from pose2mesh_release_amd import infer, loss as L, synth
    net, sd, gL, rev, J = _flat("mano")
    nv, B = 778, 3
    jreg = synth.synthetic_regressor(J, nv)
    x = synth.pose2d_batch(B, J, seed=5).cuda()
    step = infer.GraphedInference(net, rev, nv, jreg, B, scale=1000.0)
    m0 = step(x)[0].clone()
    assert step.captures == 1
    step(x)
    assert step.captures == 1                                   # unchanged weights: plain replay
    net.load_state_dict(helpers.numpy_state(net.state_dict(), 11))
    m1 = step(x)[0].clone()
    assert step.captures == 2
    assert not torch.equal(m0, m1)
    net.set_inference(real_only=False)
    with torch.no_grad():
        cam, _ = net(x)
    ref = L.MeshEpilogue(rev, nv, jreg, scale=1000.0)(cam)[0]
    assert torch.equal(m1, ref)


def test_graphed_inference_recaptures_on_an_interior_in_place_edit(hip_libs):
    """ADVICE r4 (infer.py:127): an in-place edit of ONE interior tensor (a middle conv's weight, one layer's BatchNorm
    statistics) bumps neither ops.WEIGHT_EPOCH nor the first / last parameter or buffer.  The very next call must still
    re-capture (the check walks every parameter and buffer on every call) and return the new weights' result."""
    from pose2mesh_release_amd import infer, loss as L, synth
    net, sd, gL, rev, J = _flat("mano")
    nv, B = 778, 3
    jreg = synth.synthetic_regressor(J, nv)
    x = synth.pose2d_batch(B, J, seed=6).cuda()
    step = infer.GraphedInference(net, rev, nv, jreg, B, scale=1000.0)
    for _ in range(3):                                          # (any position in a would-be check period)
        m0 = step(x)[0].clone()
    assert step.captures == 1
    mid = len(net.pose2mesh.cl) // 2
    with torch.no_grad():
        net.pose2mesh.cl[mid].weight.mul_(1.25)                 # interior parameter, in place, no epoch bump
        # (edits through `.data` do not move Tensor._version - torch gives `.data` its own counter - and are invisible to
        #  every cache of this package as they are to autograd: such callers say so with ops.bump_weight_epoch())
    m1 = step(x)[0].clone()
    assert step.captures == 2, "an interior in-place weight edit must re-capture on the very next call"
    assert not torch.equal(m0, m1)
    with torch.no_grad():
        net.pose2mesh.bn[mid].running_var.mul_(2.0)             # interior buffer
    m2 = step(x)[0].clone()
    assert step.captures == 3 and not torch.equal(m1, m2)
    step(x)
    assert step.captures == 3                                   # nothing moved: plain replay
    net.set_inference(real_only=False)
    with torch.no_grad():
        cam, _ = net(x)
    ref = L.MeshEpilogue(rev, nv, jreg, scale=1000.0)(cam)[0]
    assert torch.equal(m2, ref)


def test_graphed_inference_at_configs1_size_vs_oracle(hip_libs):
    """BASELINE configs[1] on the bench's own path: batch 64, J=17, the captured hipGraph of the real-vertices-only
    forward + Tester epilogue; 4 samples of that batch against the oracle (eval-mode samples are independent)."""
    from pose2mesh_release_amd import infer, synth
    net, sd, gL, rev, J = _flat("human36")
    nv, B = 6890, 64
    jreg = synth.synthetic_regressor(J, nv)
    x = synth.pose2d_batch(B, J, seed=31).cuda()
    step = infer.GraphedInference(net, rev, nv, jreg, B, scale=1000.0)
    assert step.graph is not None
    mesh, joints, _ = step(x)
    torch.cuda.synchronize()
    idx = [0, 21, 42, 63]
    torch.set_num_threads(16)
    with torch.no_grad():
        cam_o, _ = mo.flat_forward(sd, helpers.oracle_graphs(gL), x[idx].cpu(), False, False)
    om, oj = lo.test_epilogue(cam_o, rev, nv, torch.from_numpy(jreg))
    assert helpers.max_vertex_l2(mesh[idx].cpu() / 1000.0, om / 1000.0) <= 1e-4
    assert (joints[idx].cpu() - oj).abs().max() <= 1e-1                      # millimetres (1e-4 m)
    net.set_inference(real_only=False)


def test_graph_handle_finalizer_is_capture_safe(hip_libs):
    """Round 5: DeviceGraph.__del__ frees device memory (hipFree), which invalidates any stream capture in progress, and the
    cyclic garbage collector may run it at any time (GraphedInference re-captures failed with hipErrorStreamCaptureInvalidated
    in long processes full of dead models).  Inside a capture the handle is parked and the capture survives; it is freed
    afterwards.  ops.capture_guard (used by GraphedInference / GraphedTrainStep) also keeps the collector off inside."""
    import gc
    from pose2mesh_release_amd import ops
    gL, _, _ = helpers.golden_graphs("mano")
    victim = ops.DeviceGraph(gL[0], "cuda:0")
    a = torch.ones(8, device="cuda")
    torch.cuda.synchronize()
    ops.drain_deferred_destroys()
    g = torch.cuda.CUDAGraph()
    with ops.capture_guard(), torch.cuda.graph(g):
        assert not gc.isenabled()
        b = a * 2.0
        del victim                                   # refcount -> 0: the finalizer runs HERE, inside the capture
        assert len(ops._deferred_destroy) == 1
        c = b + 1.0
    assert gc.isenabled() and len(ops._deferred_destroy) == 0        # freed after the capture
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(c, torch.full((8,), 3.0, device="cuda"))


def test_two_back_to_back_captures_of_the_op_do_not_share_amax_words(hip_libs, monkeypatch):
    """in f16x2 the stand-alone graph_conv_cheby draws its amax words from the device's current chunk.
    A chunk created inside a stream capture is that graph's private memory; two captures taken back to back (no eager call
    between them) used to share it (the chunk was keyed on a boolean "capturing").  Now the chunk is keyed on the capture's
    identity (p2m_stream_capture_id): the second capture gets its own chunk, each graph's replays re-zero their own words,
    and replaying either graph - also after the other one is destroyed - reproduces the eager result bitwise."""
    from pose2mesh_release_amd import ops
    from pose2mesh_release_amd.cheby_graph_conv import graph_conv_cheby
    monkeypatch.setattr(ops, "GEMM_ARITH", "f16x2")
    ops.bump_weight_epoch()
    gL, _, _ = helpers.golden_graphs("mano")
    L = gL[2]
    V, Fin, Fout, B = L.shape[0], 64, 32, 3
    torch.manual_seed(5)
    cl = torch.nn.Linear(Fin * 3, Fout).cuda()
    bn = torch.nn.BatchNorm1d(Fout).cuda().eval()
    g = ops.DeviceGraph(L, "cuda:0")
    xs = [torch.randn(B, V, Fin, device="cuda") * s for s in (1.0, 300.0)]       # different magnitudes: different amax words
    with torch.no_grad():
        eager = [graph_conv_cheby(x, cl, bn, g, Fout, 3).clone() for x in xs]
        # one capture straight after the other; the static inputs are filled later
        statics, graphs, outs, chunks = [torch.zeros_like(x) for x in xs], [], [], []
        torch.cuda.synchronize()
        for st in statics:
            gr = torch.cuda.CUDAGraph()
            with ops.capture_guard(), torch.cuda.graph(gr):
                outs.append(graph_conv_cheby(st, cl, bn, g, Fout, 3))
                chunks.append(ops._amax_chunks[0][0])
            graphs.append(gr)
    assert chunks[0].data_ptr() != chunks[1].data_ptr() or chunks[0] is not chunks[1]
    assert chunks[0] is not chunks[1]
    for rep in range(3):                                   # big magnitudes first: a word that only grew would poison the rest
        for i in (1, 0):
            statics[i].copy_(xs[i] * (1.0 if rep != 1 else 0.01))
            graphs[i].replay()
            torch.cuda.synchronize()
            if rep != 1:
                assert torch.equal(outs[i], eager[i]), (rep, i)
    # the first graph goes away (its pool with it); the second still replays correctly
    keep = outs[1]
    del graphs[0], outs[0], chunks
    torch.cuda.empty_cache()
    junk = torch.full((1 << 20,), 7.0e30, device="cuda")   # whatever lands in the freed pool is not a valid amax word
    statics[1].copy_(xs[1])
    graphs[0].replay()
    torch.cuda.synchronize()
    assert torch.equal(keep, eager[1])
    del junk
    ops.bump_weight_epoch()
