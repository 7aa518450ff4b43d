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
