"""-m gpu: optimizer + the bench train step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_flat_adam_matches_torch_adam(hip_libs):
    from pose2mesh_release_amd import optim
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Tanh(), torch.nn.Linear(53, 3)).cuda()
    b = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Tanh(), torch.nn.Linear(53, 3)).cuda()
    b.load_state_dict(a.state_dict())
    oa = optim.FlatAdam(a.parameters(), lr=1e-2)
    ob = torch.optim.Adam(b.parameters(), lr=1e-2)
    x = torch.randn(64, 37, device="cuda")
    for _ in range(5):
        oa.zero_grad()
        ob.zero_grad()
        a(x).square().mean().backward()
        b(x).square().mean().backward()
        oa.step()
        ob.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert (p - q).abs().max() < 2e-6
    sd = oa.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    assert (sd["state"][0]["exp_avg"] - ob.state_dict()["state"][0]["exp_avg"]).abs().max() < 1e-6


def test_flat_rmsprop_matches_torch_rmsprop_with_multistep_lr(hip_libs):
    """The reference's recipe: RMSprop (lib/funcs_utils.py:87-91) + MultiStepLR (funcs_utils.py:101-104) + lr_check
    (funcs_utils.py:17-24) run against the fused flat optimizer unchanged."""
    from pose2mesh_release_amd import optim
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Tanh(), torch.nn.Linear(53, 3)).cuda()
    b = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Tanh(), torch.nn.Linear(53, 3)).cuda()
    b.load_state_dict(a.state_dict())
    oa = optim.get_optimizer(a, "rmsprop", lr=1e-2)
    ob = torch.optim.RMSprop(b.parameters(), lr=1e-2)
    assert isinstance(oa, torch.optim.Optimizer)
    sa = torch.optim.lr_scheduler.MultiStepLR(oa, milestones=[2], gamma=0.1)
    sb = torch.optim.lr_scheduler.MultiStepLR(ob, milestones=[2], gamma=0.1)
    x = torch.randn(64, 37, device="cuda")
    for epoch in range(4):
        for param_group in oa.param_groups:               # lr_check
            curr_lr = param_group["lr"]
        assert abs(curr_lr - ob.param_groups[0]["lr"]) < 1e-12
        for _ in range(2):
            oa.zero_grad()
            ob.zero_grad()
            a(x).square().mean().backward()
            b(x).square().mean().backward()
            oa.step()
            ob.step()
        sa.step()
        sb.step()
    assert abs(oa.param_groups[0]["lr"] - 1e-3) < 1e-12
    for p, q in zip(a.parameters(), b.parameters()):
        assert (p - q).abs().max() < 1e-5           # g / (sqrt(v) + eps) amplifies the last-bit differences of g
    sd = oa.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "square_avg"}
    assert (sd["state"][0]["square_avg"] - ob.state_dict()["state"][0]["square_avg"]).abs().max() < 1e-6
    # checkpoint round trip (main/train.py:51-58 saves optimizer.state_dict())
    oc = optim.FlatRMSprop(a.parameters(), lr=5.0)
    oc.load_state_dict(sd)
    assert abs(oc.param_groups[0]["lr"] - 1e-3) < 1e-12 and oc.step_count == oa.step_count
    assert torch.equal(oc.square_avg, oa.square_avg)


def test_train_step_runs_and_learns(hip_libs):
    """bench.py's TrainStep (reference train step, lib/core/base.py:122-148) at a small batch on the MANO-like mesh."""
    import bench
    step = bench.TrainStep(torch.device("cuda", 0), 8, "mano", 1)
    losses = [float(step()) for _ in range(8)]
    assert all(l == l for l in losses)
    assert losses[-1] < losses[0]
    for bn in step.model.pose2mesh.bn:
        if bn is not None:
            assert int(bn.num_batches_tracked) == 8


def test_fused_mesh_loss_matches_stock_losses(hip_libs):
    """p2m_mesh_loss (value + gradient) against the stock-torch losses that mirror lib/core/loss.py and
    lib/core/base.py:130-143."""
    import numpy as np
    from pose2mesh_release_amd import loss as L, synth
    import bench
    _, faces = synth.hull_mesh(500, 3)
    nv, V0, J, B = 500, 736, 17, 6
    rng = np.random.default_rng(0)
    perm_rev = rng.permutation(V0)
    jreg = synth.synthetic_regressor(J, nv)
    g = torch.Generator().manual_seed(1)
    cam = (torch.randn(B, V0, 3, generator=g) * 0.3).cuda().requires_grad_(True)
    gt_mesh = (torch.randn(B, nv, 3, generator=g) * 0.3).cuda()
    gt_pose = (torch.randn(B, J, 3, generator=g) * 300).cuda()
    vm = (torch.rand(B, 1, 1, generator=g) > 0.3).float().cuda()
    vp = (torch.rand(B, 1, 1, generator=g) > 0.3).float().cuda()
    stock = L.get_loss(faces)
    perm = torch.as_tensor(perm_rev[:nv], dtype=torch.long, device="cuda")
    pm = cam[:, perm, :]
    pose = torch.matmul(torch.from_numpy(jreg).cuda()[None], pm * 1000)
    parts = [stock[0](pm, gt_mesh, vm), 0.1 * stock[1](pm, gt_mesh), 20 * stock[2](pm, gt_mesh),
             1e-3 * stock[3](pose, gt_pose, vp)]
    sum(parts).backward()
    ref_grad = cam.grad.clone()
    cam2 = cam.detach().clone().requires_grad_(True)
    fused = L.FusedMeshLoss(faces, perm_rev, jreg)
    total, comp = fused(cam2, gt_mesh, gt_pose, vm, vp)
    total.backward()
    for a, b in zip(comp.tolist(), [float(p) for p in parts]):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b))
    assert (cam2.grad - ref_grad).abs().max() <= 2e-5 * ref_grad.abs().max()
    fake = np.setdiff1d(np.arange(V0), perm_rev[:nv])
    assert float(cam2.grad[:, torch.as_tensor(fake, device="cuda")].abs().max()) == 0.0


def test_fused_coord_loss_matches_the_stock_module(hip_libs):
    """p2m_coord_loss (the lifted-pose term, lib/core/loss.py:10-23 weighted as lib/core/base.py:139; value + gradient in one
    launch) against weight * CoordLoss for the mask shapes the reference uses, incl. exact zeros of the difference."""
    from pose2mesh_release_amd import loss as L
    B, J = 12, 17
    g = torch.Generator().manual_seed(5)
    pred0 = (torch.randn(B, J, 3, generator=g) * 300)
    tgt = (torch.randn(B, J, 3, generator=g) * 300)
    tgt[0, 0] = pred0[0, 0]                                        # d == 0: sign(0) = 0 on both sides
    masks = {"none": None, "sample": (torch.rand(B, 1, 1, generator=g) > 0.3).float(),
             "joint": (torch.rand(B, J, 1, generator=g) > 0.3).float(), "flat": (torch.rand(B, generator=g) > 0.3).float()}
    fused = L.FusedCoordLoss(1e-3)
    for name, m in masks.items():
        a = pred0.clone().cuda().requires_grad_(True)
        b = pred0.clone().cuda().requires_grad_(True)
        mm = None if m is None else m.cuda()
        mref = None if m is None else (mm.view(B, 1, 1) if mm.dim() == 1 else mm)
        ref = 1e-3 * (L.CoordLoss(has_valid=True)(a, tgt.cuda(), mref) if m is not None else L.CoordLoss()(a, tgt.cuda()))
        (ref * 3.0).backward()
        out = fused(b, tgt.cuda(), mm)
        assert out.dim() == 0
        (out * 3.0).backward()
        assert abs(float(out) - float(ref)) <= 2e-6 * abs(float(ref)), name
        assert float((a.grad - b.grad).abs().max()) <= 1e-6 * float(a.grad.abs().max()), name
        assert float(b.grad[0, 0].abs().max()) == 0.0 and float(a.grad[0, 0].abs().max()) == 0.0, name
    with pytest.raises(Exception):
        fused(pred0, tgt)                                          # CPU tensors: no silent fallback
    # a mask with B elements is per-sample only when its shape says so - a [J, 1] mask with J == B broadcasts
    # per JOINT in the reference's `pred * target_valid` (lib/core/loss.py:17-19), and so it does here
    Bq = J
    pq, tq = (torch.randn(Bq, J, 3, generator=g) * 300).cuda(), (torch.randn(Bq, J, 3, generator=g) * 300).cuda()
    mj = (torch.rand(J, 1, generator=g) > 0.4).float().cuda()
    assert 0 < float(mj.sum()) < J
    a = pq.clone().requires_grad_(True)
    b = pq.clone().requires_grad_(True)
    ref = 1e-3 * L.CoordLoss(has_valid=True)(a, tq, mj)
    ref.backward()
    out = fused(b, tq, mj)
    out.backward()
    assert abs(float(out) - float(ref)) <= 2e-6 * abs(float(ref))
    assert float((a.grad - b.grad).abs().max()) <= 1e-6 * float(a.grad.abs().max())
    with pytest.raises(RuntimeError):
        fused(pq, tq, torch.ones(J + 1, 1, device="cuda"))        # not broadcastable: raises, as torch would


def test_in_place_gradient_accumulation_is_bitwise_the_autograd_path(hip_libs):
    """Pose2Mesh.accumulate_grads_in_place: weight / bias / BatchNorm gradients written straight into the flat gradient
    buffer by the unpack / finalize kernels == the tensors autograd would have added, and they ADD to what is there."""
    import helpers
    from pose2mesh_release_amd import meshnet, optim
    gL, _, _ = helpers.golden_graphs("human36")
    flats = []
    for direct in (False, True):
        net = meshnet.get_model(5, 3, gL, mano=False)
        net.load_state_dict(helpers.numpy_state(net.state_dict(), 3))
        net = net.cuda().train()
        opt = optim.FlatAdam(net.parameters(), lr=1e-3)
        net.accumulate_grads_in_place(direct)
        x = helpers.meshnet_input(3, 17, seed=4).cuda()
        w = torch.randn(3, gL[0].shape[0], 3, generator=torch.Generator().manual_seed(1)).cuda()
        for rep in range(2):                       # second backward without zero_grad: gradients must accumulate
            (net(x) * w).sum().backward()
        torch.cuda.synchronize()
        flats.append(opt.flat_grad.clone())
    assert torch.equal(flats[0], flats[1])
    assert float(flats[0].abs().max()) > 0


@pytest.mark.parametrize("opt_name", ["adam", "rmsprop"])
def test_graphed_train_step_is_bitwise_the_eager_loop(hip_libs, opt_name):
    """train.GraphedTrainStep: 3 eager warm-up steps, capture, 3 replays (with an lr change in between, as MultiStepLR
    makes) == 6 eager steps, bit for bit.  MeshNet alone (every kernel of the step is this package's, all
    deterministic; with PoseNet in the step hipBLASLt may pick another algorithm under capture, which moves the last
    bit of its output)."""
    import helpers
    from pose2mesh_release_amd import meshnet, optim, train
    gL, _, _ = helpers.golden_graphs("mano")
    B, J = 6, int(gL[-1].shape[0])
    x = helpers.meshnet_input(B, J, seed=4).cuda()
    w = torch.randn(B, gL[0].shape[0], 3, generator=torch.Generator().manual_seed(1)).cuda()
    results = []
    for graphed in (False, True):
        net = meshnet.get_model(5, 3, gL, mano=True)
        net.load_state_dict(helpers.numpy_state(net.state_dict(), 3))
        net = net.cuda().train()
        opt = (optim.FlatAdam if opt_name == "adam" else optim.FlatRMSprop)(net.parameters(), lr=1e-3)
        net.accumulate_grads_in_place(True)

        def loss_fn():
            return (net(x) * w).sum() * 1e-3

        losses = []
        if graphed:
            step = train.GraphedTrainStep(net, opt, loss_fn, warmup=3)
            for i in range(6):
                if i == 4:
                    opt.param_groups[0]["lr"] = 1e-4
                losses.append(float(step()))
            assert step.graph is not None
        else:
            for i in range(6):
                if i == 4:
                    opt.param_groups[0]["lr"] = 1e-4
                opt.zero_grad()
                loss = loss_fn()
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        results.append((losses, opt.flat_param.clone(), [b.clone() for b in opt._bufs.values()],
                        {k: v.clone() for k, v in net.state_dict().items() if "running" in k or "tracked" in k}))
    (la, pa, ba, sa), (lb, pb, bb, sb) = results
    assert la == lb, (la, lb)
    assert torch.equal(pa, pb)
    for u, v in zip(ba, bb):
        assert torch.equal(u, v)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_graphed_flat_model_step_tracks_the_eager_loop(hip_libs):
    """The full FlatPose2Mesh step (PoseNet on hipBLASLt + MeshNet + fused losses) captured: the loss curve follows the
    eager one (not bitwise: see above)."""
    from pose2mesh_release_amd import loss as p2m_loss, optim, pose2mesh_net, synth, train
    faces, gL, perm_rev, J = synth.make_graphs("mano")
    nv = int(faces.max()) + 1
    B = 6
    gen = torch.Generator().manual_seed(3)
    pose2d = synth.pose2d_batch(B, J, 11).cuda()
    gt_mesh = (torch.randn(B, nv, 3, generator=gen) * 0.3).cuda()
    gt_pose = (torch.randn(B, J, 3, generator=gen) * 300).cuda()
    one = torch.ones(B, 1, 1).cuda()
    curves = []
    for graphed in (False, True):
        torch.manual_seed(5)
        model = pose2mesh_net.get_model(J, gL).cuda().train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        opt = optim.FlatRMSprop(model.parameters(), lr=1e-4)
        model.pose2mesh.accumulate_grads_in_place(True)
        mesh_loss = p2m_loss.FusedMeshLoss(faces, perm_rev, synth.synthetic_regressor(J, nv))
        lift_loss = p2m_loss.get_loss(faces)[4]

        def loss_fn():
            mesh, lift = model(pose2d)
            total, _ = mesh_loss(mesh, gt_mesh, gt_pose, one, one)
            return total + 1e-3 * lift_loss(lift, gt_pose, one)

        if graphed:
            step = train.GraphedTrainStep(model, opt, loss_fn, warmup=2)
            curves.append([float(step()) for _ in range(6)])
        else:
            c = []
            for _ in range(6):
                opt.zero_grad()
                loss = loss_fn()
                loss.backward()
                opt.step()
                c.append(float(loss.detach()))
            curves.append(c)
    a, b = curves
    assert all(abs(u - v) <= 2e-3 * abs(u) for u, v in zip(a, b)), (a, b)
    assert a[-1] < a[0] and b[-1] < b[0]


def test_eval_between_graph_replays_sees_the_current_weights(hip_libs):
    """the reference's loop alternates train epochs and validation epochs.  A replayed step updates
    parameters and BatchNorm running statistics by raw pointer, so the derived operands cached for eval-mode forwards
    (packed / split weights, eval BatchNorm coefficients) must be invalidated by every replay: replay, eval, replay, eval
    must give the eval outputs of the eager loop, bit for bit."""
    import helpers
    from pose2mesh_release_amd import meshnet, optim, train
    gL, _, _ = helpers.golden_graphs("mano")
    B, J = 4, int(gL[-1].shape[0])
    x = helpers.meshnet_input(B, J, seed=4).cuda()
    xe = helpers.meshnet_input(3, J, seed=9).cuda()
    w = torch.randn(B, gL[0].shape[0], 3, generator=torch.Generator().manual_seed(1)).cuda()
    outs = []
    for graphed in (False, True):
        net = meshnet.get_model(5, 3, gL, mano=True)
        net.load_state_dict(helpers.numpy_state(net.state_dict(), 3))
        net = net.cuda().train()
        opt = optim.FlatAdam(net.parameters(), lr=1e-3)
        net.accumulate_grads_in_place(True)

        def loss_fn():
            return (net(x) * w).sum() * 1e-3

        def eager():
            opt.zero_grad()
            loss_fn().backward()
            opt.step()
        step = train.GraphedTrainStep(net, opt, loss_fn, warmup=1) if graphed else eager
        evals = []
        for i in range(6):                       # graphed: 1 eager, capture + replay, then replays
            step()
            if i >= 2:
                net.eval()
                with torch.no_grad():
                    evals.append(net(xe).clone())
                net.train()
        if graphed:
            assert step.graph is not None
        outs.append(evals)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert not torch.equal(outs[1][0], outs[1][-1])          # the weights did move between the evals


def test_graph_replays_without_host_sync_keep_their_own_step_scalars(hip_libs):
    """with replay the host runs many steps ahead of the GPU; every step's {lr, bias corrections,
    grad_scale} must reach the device intact (a single pinned staging slot would be overwritten before its async copy
    ran).  40 replays issued back to back behind a long-running kernel, lr changed half way == the eager loop, bitwise."""
    import helpers
    from pose2mesh_release_amd import meshnet, optim, train
    gL, _, _ = helpers.golden_graphs("mano")
    B, J = 2, int(gL[-1].shape[0])
    x = helpers.meshnet_input(B, J, seed=4).cuda()
    w = torch.randn(B, gL[0].shape[0], 3, generator=torch.Generator().manual_seed(1)).cuda()
    N = 40
    finals = []
    for graphed in (False, True):
        net = meshnet.get_model(5, 3, gL, mano=True)
        net.load_state_dict(helpers.numpy_state(net.state_dict(), 3))
        net = net.cuda().train()
        opt = optim.FlatAdam(net.parameters(), lr=1e-3)
        net.accumulate_grads_in_place(True)

        def loss_fn():
            return (net(x) * w).sum() * 1e-3
        if graphed:
            step = train.GraphedTrainStep(net, opt, loss_fn, warmup=1)
            step()
            step()                                            # capture + first replay
            torch.cuda.synchronize()
            big = torch.randn(8192, 8192, device="cuda")
            for _ in range(6):                                # ~tens of ms of queued GPU work: the host gets far ahead
                big = big @ big * 1e-4
            for i in range(2, N):
                if i == N // 2:
                    opt.param_groups[0]["lr"] = 1e-4
                step()                                        # no host sync in this loop
        else:
            for i in range(N):
                if i == N // 2:
                    opt.param_groups[0]["lr"] = 1e-4
                opt.zero_grad()
                loss_fn().backward()
                opt.step()
        torch.cuda.synchronize()
        finals.append((opt.flat_param.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone()))
    for a, b in zip(*finals):
        assert torch.equal(a, b)


def test_bench_line_carries_the_contract_keys(hip_libs):
    """`python bench.py` at a small batch, as the driver runs it for N = 1: ONE JSON line with the contract's keys, the
    roofline object of the dominant kernel family (live HIP-event timing; PMC traffic with its code id and staleness flag), the
    untimed parity check against the CPU oracle, the CPU baseline object and - where the amdgpu hwmon files exist - the
    shader clock / package power sampled during the timed steps."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "2", "--batch", "32", "--also", "none",
                        "--no-arith-ab", "--cpu-seconds", "2"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_check"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["unit"] == "meshes/s" and d["value"] > 0
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "traffic_code_id", "code_id"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert d["parity_check"]["ok"] and d["parity_check"]["max_vertex_l2_m"] <= 1e-4
    if "power" in d:               # (no hwmon files: the key is absent, never a guess)
        pw = d["power"]
        assert 100 <= pw["sclk_mhz_median"] <= 3000 and 50 <= pw["power_w_median"] <= 2000 and pw["samples"] >= 3
        if rf["bound"] == "mfma":
            assert abs(rf["frac_at_step_sclk"] - rf["frac"] * 2400 / pw["sclk_mhz_median"]) < 1e-3
