"""CPU, gloo, world_size 2: the data-parallel gradient exchange (dist.BucketedAllReduce) averages
gradients exactly like a single process over the concatenated batch; bench.py's launch contract
(RANK/WORLD_SIZE/MASTER_* from the environment) initialises correctly."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_setup(model):
    params = [p for p in model.parameters()]
    offsets, n = [], 0
    for p in params:
        offsets.append(n)
        n += (p.numel() + 3) // 4 * 4
    flat = torch.zeros(n)
    for p, o in zip(params, offsets):
        p.grad = flat[o:o + p.numel()].view_as(p)
    return params, offsets, flat


def _worker(rank, world, port, bucket_bytes, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from pose2mesh_release_amd import dist as pd
    r, w, _ = pd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    seen = pd.ranks_seen()                     # the diagnostics bench.py --gpus N prints: every rank sees every rank
    assert seen["backend"] == "gloo" and seen["world"] == world and seen["ranks"] == list(range(world))
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(20, 64), torch.nn.ReLU(), torch.nn.Linear(64, 33), torch.nn.ReLU(),
                                torch.nn.Linear(33, 5))
    params, offsets, flat = _flat_setup(model)
    red = pd.BucketedAllReduce(params, offsets, flat, bucket_bytes=bucket_bytes)
    g = torch.Generator().manual_seed(42)
    X, Y = torch.randn(16, 20, generator=g), torch.randn(16, 5, generator=g)
    xs, ys = X[rank * 8:(rank + 1) * 8], Y[rank * 8:(rank + 1) * 8]
    red.timing = True                          # a CPU buffer has no streams to time: the switch must be harmless
    for _ in range(2):                         # two steps: hooks/pending counters must re-arm
        flat.zero_()
        red.mark_step_start()
        ((model(xs) - ys) ** 2).mean().backward()
        assert len(red.launch_log) <= len(red.buckets)
        scale = red.finish()
        flat.mul_(scale)
        assert red.launch_log == [] and len(red.last_launch_log) == len(red.buckets)   # one step's entries, then cleared
    assert red.timing_report() is None
    if rank == 0:
        torch.save({"flat": flat.clone(), "nb": len(red.buckets)}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [1 << 30, 4096, 64])
def test_bucketed_allreduce_matches_full_batch(tmp_path, bucket_bytes):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), bucket_bytes, out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(20, 64), torch.nn.ReLU(), torch.nn.Linear(64, 33), torch.nn.ReLU(),
                                torch.nn.Linear(33, 5))
    params, offsets, flat = _flat_setup(model)
    g = torch.Generator().manual_seed(42)
    X, Y = torch.randn(16, 20, generator=g), torch.randn(16, 5, generator=g)
    ((model(X) - Y) ** 2).mean().backward()       # mean over 16 == average of the two 8-sample means
    assert (got["flat"] - flat).abs().max() < 1e-6
    if bucket_bytes == 64:
        assert got["nb"] > 3                       # really exercised several buckets
    if bucket_bytes == 1 << 30:
        assert got["nb"] == 1


def test_single_process_is_a_noop():
    from pose2mesh_release_amd import dist as pd
    model = torch.nn.Linear(4, 4)
    params, offsets, flat = _flat_setup(model)
    red = pd.BucketedAllReduce(params, offsets, flat)
    model(torch.ones(2, 4)).sum().backward()
    before = flat.clone()
    assert red.finish() == 1.0 and torch.equal(flat, before)


def _worker_two_branches(rank, world, port, out):
    """bench.py's data-parallel step: two branches without shared parameters, back-propagated one after the other
    (PoseNet branch first so that its buckets reduce under the MeshNet backward)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from pose2mesh_release_amd import dist as pd
    pd.init_from_env(backend="gloo")
    torch.manual_seed(0)
    model = torch.nn.ModuleDict({"a": torch.nn.Linear(20, 64), "b": torch.nn.Linear(20, 9)})
    params, offsets, flat = _flat_setup(model)
    red = pd.BucketedAllReduce(params, offsets, flat, bucket_bytes=2048)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(16, 20, generator=g)
    xs = X[rank * 8:(rank + 1) * 8]
    for _ in range(2):
        flat.zero_()
        la, lb = model["a"](xs).pow(2).mean(), model["b"](xs.detach()).abs().mean()
        la.backward()
        lb.backward()
        flat.mul_(red.finish())
    if rank == 0:
        torch.save({"flat": flat.clone(), "nb": len(red.buckets)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_backward_calls_share_one_reducer(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_two_branches, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    model = torch.nn.ModuleDict({"a": torch.nn.Linear(20, 64), "b": torch.nn.Linear(20, 9)})
    params, offsets, flat = _flat_setup(model)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(16, 20, generator=g)
    (model["a"](X).pow(2).mean() + model["b"](X).abs().mean()).backward()
    assert got["nb"] >= 2
    assert (got["flat"] - flat).abs().max() < 1e-6


def _worker_world8(rank, world, port, out):
    """BASELINE configs[3]'s topology on CPU: 8 ranks x B/8 samples, gradients of one flat buffer reduced in several
    buckets whose LAST one is smaller than the rest; per-rank BatchNorm statistics (DataParallel semantics,
    lib/core/base.py:108) - only the gradients are exchanged."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from pose2mesh_release_amd import dist as pd
    r, w, _ = pd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    seen = pd.ranks_seen()
    assert seen["world"] == world and seen["ranks"] == list(range(world))
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.BatchNorm1d(40), torch.nn.ReLU(),
                                torch.nn.Linear(40, 31), torch.nn.BatchNorm1d(31), torch.nn.ReLU(), torch.nn.Linear(31, 7))
    params, offsets, flat = _flat_setup(model)
    red = pd.BucketedAllReduce(params, offsets, flat, bucket_bytes=2000)       # 500-float buckets, 2 188 floats in all
    sizes = [e - s for s, e, _ in red.buckets]
    assert len(sizes) >= 3 and sizes[-1] < max(sizes), sizes                  # several buckets, an uneven last one
    assert [s for s, _, _ in red.buckets] == sorted(s for s, _, _ in red.buckets)     # buckets in buffer order
    g = torch.Generator().manual_seed(11)
    X, Y = torch.randn(64, 12, generator=g), torch.randn(64, 7, generator=g)
    n = 64 // world
    xs, ys = X[rank * n:(rank + 1) * n], Y[rank * n:(rank + 1) * n]
    for _ in range(2):
        flat.zero_()
        ((model(xs) - ys) ** 2).mean().backward()
        flat.mul_(red.finish())
        # gradients are written LAST layer first: the buckets must have been launched from the back of the buffer
        order = [b for b, _ in red.last_launch_log]
        assert sorted(order) == list(range(len(red.buckets))) and order[0] == len(red.buckets) - 1, order
    torch.save({"flat": flat.clone(), "rm": model[1].running_mean.clone()}, out + f"_{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_8_is_the_mean_of_the_shard_gradients(tmp_path):
    """the config-#4 topology (8 ranks) had only ever run with 2.  The averaged flat gradient on every
    rank == the mean of the 8 per-shard gradients computed by one process (each shard with its OWN BatchNorm batch
    statistics - what nn.DataParallel replicas do); running statistics stay per rank."""
    world = 8
    out = str(tmp_path / "w8")
    mp.spawn(_worker_world8, args=(world, _free_port(), out), nprocs=world, join=True)
    got = [torch.load(out + f"_{r}.pt") for r in range(world)]
    for r in range(1, world):
        assert torch.equal(got[r]["flat"], got[0]["flat"])                       # every rank holds the same average
    assert not torch.equal(got[0]["rm"], got[1]["rm"])                           # BatchNorm statistics: per rank
    g = torch.Generator().manual_seed(11)
    X, Y = torch.randn(64, 12, generator=g), torch.randn(64, 7, generator=g)
    mean = None
    for r in range(world):
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.BatchNorm1d(40), torch.nn.ReLU(),
                                    torch.nn.Linear(40, 31), torch.nn.BatchNorm1d(31), torch.nn.ReLU(), torch.nn.Linear(31, 7))
        params, offsets, flat = _flat_setup(model)
        for _ in range(2):                      # the second step's gradient does not depend on the first (no optimizer)
            flat.zero_()
            ((model(X[r * 8:(r + 1) * 8]) - Y[r * 8:(r + 1) * 8]) ** 2).mean().backward()
        mean = flat.clone() if mean is None else mean + flat
    mean /= world
    assert (got[0]["flat"] - mean).abs().max() < 1e-6


def test_bench_refuses_a_scaling_point_without_the_ranks():
    """`bench.py --gpus N` must not print a JSON line unless N ranks are really there (RCCL, N distinct
    GPUs: checked on the GPU box by tests/test_gpu_dist.py).  A launcher that set another world size than --gpus: no line,
    non-zero exit, and the message says how to launch.  (WORLD_SIZE unset: bench.py starts the ranks itself - next test.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK")}, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
    assert "WORLD_SIZE=1" in r.stderr and "torch.distributed.run" in r.stderr


def test_bench_gpus_n_without_launcher_starts_n_ranks():
    """bench.py's launch contract: `python bench.py --gpus 8` with WORLD_SIZE unset re-executes itself under
    torch.distributed.run with 8 processes on one node and a 127.0.0.1 rendezvous (the command is printed, not run: no GPU
    here); with WORLD_SIZE set by a launcher nothing is re-launched."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["P2M_BENCH_LAUNCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["self_launch"]
    assert cmd[1:5] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8"]
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7].endswith("bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]


def test_power_watch_reads_hwmon_files(tmp_path):
    """bench.PowerWatch over a fake amdgpu hwmon directory: the medians of what the files held while it ran, the cap, and no
    `power` object at all when nothing could be sampled (a box without the files must not break the bench line)."""
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    d = tmp_path / "hwmon0"
    d.mkdir()
    (d / "freq1_input").write_text("2100000000\n")
    (d / "power1_input").write_text("1197000000\n")
    (d / "power1_cap").write_text("1400000000\n")
    w = bench.PowerWatch(0, period=0.002)
    w.dirs, w.how = [str(d)], "test"
    with w:
        time.sleep(0.1)
    r = w.report()
    assert r["sclk_mhz_median"] == 2100 and r["power_w_median"] == 1197 and r["power_cap_w"] == 1400 and r["samples"] >= 3
    empty = bench.PowerWatch(0, period=0.002)
    empty.dirs = [str(tmp_path / "missing")]
    with empty:
        time.sleep(0.02)
    assert empty.report() is None
