#!/usr/bin/env python3
"""bench.py -- SMPL meshes/sec, forward + backward + Adam, batch 256 per GPU (BASELINE.json configs[2]).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --mode infer            (BASELINE.json configs[1]: batch 64, J=17, eval forward + Tester epilogue)

ARITHMETIC OF THE HEADLINE (round 5).  The reference computes in fp32.  `value` is measured with the fp32 contractions emulated
EXACTLY on the matrix pipe - every operand as three bf16 slices = 24-bit significands, six slice products per fp32 product
(`--arith bf16x3`, the default of this script; `dtype` says so).  The package's import-time default, `f16x2` (two scaled fp16
slices = 22-bit operands, three products: tolerance-compliant - vertex L2 1.4e-5 vs float64 at this batch - but narrower than
fp32), is the FAST MODE: measured in the same process at the same warm-up / step counts and reported under `arith_ab`, never as
`value`.  `--arith f16x2|f32` or $P2M_GEMM_ARITH select another main arithmetic explicitly.

Protocol: 20 warm-up + 50 timed steps by default; `value` / `ms_per_step` are wall-clock over the timed region,
`ms_per_step_stats` the per-step median / p10 / p90 from one HIP event per step.  The default 1-GPU train run also reports, from
the same process and after the timed region: `arith_ab` (f16x2 at full protocol, native-f32 MFMA briefly), `also` (configs[1]
inference and configs[4] MANO B=512 train, each in the main arithmetic and in the fast mode; `--also none` to skip), and for N > 1
`multi_gpu` (ranks seen over the job's backend, per-rank per-bucket all-reduce timings and the hidden fraction).  `python bench.py
--gpus N` with no launcher (WORLD_SIZE unset) starts its N ranks itself (round 6); either way NO line is printed unless N ranks,
the nccl (RCCL) backend and N distinct GPUs are seen.

One "step" (train mode) = one full reference train step (lib/core/base.py:122-148) on one synthetic batch that is
already resident in HBM: FlatPose2Mesh forward (PoseNet + coarse-to-fine GCN, both on hand-written kernels), perm-reverse gather,
joint regression, the five reference losses, backward, [gradient all-reduce], Adam.  Weak scaling: every rank owns `--batch`
samples.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline         the contraction family with the largest time in the step (bf16x3: the basis-inside-the-contraction tile
                   kernels): algorithmic fp32 FLOPs and algorithmic HBM bytes of its launches / their HIP-event time, against
                   the matrix-pipe peak expressed in algorithmic FLOPs (bf16x3: 2500 / 6 = 416.7 TFLOP/s; f16x2: 2500 / 3;
                   P2M_GEMM_ARITH=f32: 157.3 TFLOP/s) and 8 TB/s - `bound` names the binding one; `traffic` = HBM bytes per
                   launch from the committed rocprofv3 PMC passes of this same command (profiles/traffic_latest.json,
                   FETCH_SIZE x 2 + WRITE_SIZE, tools/rocprof_traffic.sh); the other families follow as roofline_<family>
  roofline_sparse  the Chebyshev-basis kernels (HBM-bound): BYTES THE LAUNCHES MOVE (real-vertex rows only, un-pooled
                   inputs read at the coarse resolution) / HIP-event time, vs 8 TB/s.  SURVEY 8(d)'s all-V-rows figure
                   is kept as `speedup_equivalent` (it credits the fake-vertex split, which is an algorithmic saving,
                   not bandwidth)
  cpu_baseline     the oracle port of the reference CPU path (oracle/meshnet_oracle.py + oracle/loss_oracle.py), same
                   train step, timed on this host's cores (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pose2mesh_release_amd import build as p2m_build  # noqa: E402
from pose2mesh_release_amd import dist as p2m_dist  # noqa: E402
from pose2mesh_release_amd import loss as p2m_loss  # noqa: E402
from pose2mesh_release_amd import ops, optim, pose2mesh_net, synth  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, dense FP32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide, dense BF16 matrix peak; the bf16x3 contraction spends 6 bf16 MFMA flops
                                  # per algorithmic fp32 flop -> 416.7 TFLOP/s fp32-equivalent
PEAK_HBM_GBPS = 8000.0            # HBM3E spec peak (achievable copy ceiling ~6300 GB/s)

synthetic_regressor = synth.synthetic_regressor


def _dense_gflop_fwd(mesh):
    """algorithmic dense FLOPs per mesh, forward = sum over convs 2*V*(K*Fin)*Fout + the fc lift."""
    fwd = sum(2.0 * mesh.graph_L[L.graph].shape[0] * 3 * L.Fin * L.Fout for L in mesh._layers)
    return (fwd + 2.0 * mesh.fc.in_features * mesh.fc.out_features) / 1e9


class TrainStep:
    """The reference train step (lib/core/base.py:122-148) on resident synthetic data."""

    def __init__(self, device, B, joint_set, world, edge_loss=True, seed=123, stock_losses=False, optimizer="adam"):
        self.device, self.B = device, B
        faces, graph_L, perm_rev, J = synth.make_graphs(joint_set)
        self.J, self.nv = J, int(faces.max()) + 1
        torch.manual_seed(seed)                                     # main/train.py:12
        self.model = pose2mesh_net.get_model(J, graph_L).to(device).train()
        if optimizer == "adam":
            self.opt = optim.FlatAdam(self.model.parameters(), lr=1e-3)          # funcs_utils.py:92-96
        else:
            self.opt = optim.FlatRMSprop(self.model.parameters(), lr=1e-3)       # funcs_utils.py:87-91 (the yaml recipe)
        self.model.accumulate_grads_in_place(True)                  # gradients land in opt.flat_grad directly (both halves)
        self.reducer = p2m_dist.BucketedAllReduce(self.opt.params, self.opt.offsets, self.opt.flat_grad) \
            if world > 1 else None
        if self.reducer is not None:       # in-place gradients are reported per layer, from inside the backward
            self.model.set_grad_sink(self.reducer.notify)
        self.losses = p2m_loss.get_loss(faces)
        self.edge_loss = edge_loss
        self.stock_losses = stock_losses
        self.mesh_loss = p2m_loss.FusedMeshLoss(faces, perm_rev, synthetic_regressor(J, int(faces.max()) + 1),
                                                w_normal=1e-1, w_edge=20.0 if edge_loss else 0.0, w_joint=1e-3)
        self.lift_loss = p2m_loss.FusedCoordLoss(1e-3)              # base.py:128,139 (the lifted-pose term), one launch
        self.perm_rev = np.asarray(perm_rev)
        self.perm = torch.as_tensor(self.perm_rev[:self.nv], dtype=torch.long, device=device)
        self.Jreg = torch.from_numpy(synthetic_regressor(J, self.nv)).to(device)
        g = torch.Generator().manual_seed(seed + int(os.environ.get("RANK", "0")))
        self.pose2d = synth.pose2d_batch(B, J, seed + int(os.environ.get("RANK", "0"))).to(device)
        self.gt_mesh = (torch.randn(B, self.nv, 3, generator=g) * 0.3).to(device)
        self.gt_reg = (torch.randn(B, J, 3, generator=g) * 300).to(device)
        self.gt_lift = (torch.randn(B, J, 3, generator=g) * 300).to(device)
        self.one = torch.ones(B, 1, 1, device=device)
        self.V0 = graph_L[0].shape[0]
        self.graph_L = graph_L
        self.faces = faces
        # forward + backward = 3x the forward (dX and dW cost one forward each)
        self.dense_gflop_fwd = _dense_gflop_fwd(self.model.pose2mesh)
        self.dense_gflop_fwd_bwd = 3.0 * self.dense_gflop_fwd

    def loss_fn(self):
        """Forward + the five reference losses (fused form) -> the total loss: what train.GraphedTrainStep captures
        between zero_grad and the optimizer step."""
        pred_mesh, lift_pose = self.model(self.pose2d)
        mesh_total, _ = self.mesh_loss(pred_mesh, self.gt_mesh, self.gt_reg, self.one, self.one)
        return mesh_total + self.lift_loss(lift_pose, self.gt_lift, self.one)

    def __call__(self):
        m = self.model
        if self.reducer is not None:
            self.reducer.mark_step_start()          # (no-op unless the reducer's timing is switched on)
        self.opt.zero_grad()
        pred_mesh, lift_pose = m(self.pose2d)
        if not self.stock_losses:
            # base.py:130-143 (gather, J-regression, vertex/normal/edge/joint losses) in one fused HIP call
            mesh_total, _ = self.mesh_loss(pred_mesh, self.gt_mesh, self.gt_reg, self.one, self.one)
            lift = self.lift_loss(lift_pose, self.gt_lift, self.one)
            loss = mesh_total + lift
            if self.reducer is not None:
                # the two branches share no parameters (pose2mesh_net.py:20 detaches the lifted pose): back-propagate
                # the PoseNet branch first so that the all-reduce of its 270 MB of gradients runs under the MeshNet
                # backward instead of after it (autograd alone would run the later-built MeshNet node first)
                lift.backward()
                mesh_total.backward()
            else:
                loss.backward()
            scale = self.reducer.finish() if self.reducer is not None else 1.0
            self.opt.step(scale)
            return loss
        pred_mesh = pred_mesh[:, self.perm, :]                                        # base.py:130
        pred_pose = torch.matmul(self.Jreg[None, :, :], pred_mesh * 1000)             # base.py:131
        L = self.losses
        loss = L[0](pred_mesh, self.gt_mesh, self.one) + 1e-1 * L[1](pred_mesh, self.gt_mesh) \
            + 1e-3 * L[3](pred_pose, self.gt_reg, self.one) + 1e-3 * L[4](lift_pose, self.gt_lift, self.one)
        if self.edge_loss:
            loss = loss + 20 * L[2](pred_mesh, self.gt_mesh)                          # base.py:141-143
        loss.backward()
        scale = self.reducer.finish() if self.reducer is not None else 1.0
        self.opt.step(scale)
        return loss


class InferStep:
    """BASELINE.json configs[1]: the Tester's step (lib/core/base.py:196-204) on resident synthetic data --
    FlatPose2Mesh in eval() under no_grad, then mesh = pred[:, perm_reverse[:nv]] * 1000 and joints = J_reg @ mesh.
    Default: the captured hipGraph of the real-vertices-only inference path (pose2mesh_release_amd/infer.py);
    --infer-path general runs the drop-in module + the separate epilogue kernel, launch by launch."""

    def __init__(self, device, B, joint_set, seed=123, path="graph"):
        from pose2mesh_release_amd import infer
        self.device, self.B, self.path = device, B, path
        faces, graph_L, perm_rev, J = synth.make_graphs(joint_set)
        self.J, self.nv = J, int(faces.max()) + 1
        torch.manual_seed(seed)
        self.model = pose2mesh_net.get_model(J, graph_L).to(device).eval()
        self.pose2d = synth.pose2d_batch(B, J, seed).to(device)
        self.V0 = graph_L[0].shape[0]
        self.dense_gflop_fwd = _dense_gflop_fwd(self.model.pose2mesh)
        self.dense_gflop_fwd_bwd = 3.0 * self.dense_gflop_fwd
        if path == "general":
            self.epilogue = p2m_loss.MeshEpilogue(perm_rev, self.nv, synthetic_regressor(J, self.nv), scale=1000.0)
            self.step = None
        else:
            self.step = infer.GraphedInference(self.model, perm_rev, self.nv, synthetic_regressor(J, self.nv), B,
                                               scale=1000.0, use_graph=(path == "graph"))

    def __call__(self):
        if self.step is not None:
            return self.step(self.pose2d)
        with torch.no_grad():
            pred_mesh, _ = self.model(self.pose2d)
            return self.epilogue(pred_mesh)


def _cpu_info():
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return os.cpu_count() or 1, model


def cpu_baseline(joint_set, budget_s, edge_loss=True, batch=32, mode="train"):
    """Oracle port of the reference CPU path (oracle/meshnet_oracle.py, oracle/loss_oracle.py), same step, on this
    host's cores.  BASELINE.md section 3: B=32 train steps (B=64 forward for --mode infer), 1 warm-up, up to 3 timed
    steps inside a time budget."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import loss_oracle as lo
    import meshnet_oracle as mo
    nproc, cpu_model = _cpu_info()
    # torch's CPU sparse/BN kernels stop scaling (and then collapse) well before a 256-thread host is full:
    # 16 threads is the sweet spot measured for this path (0.05 meshes/s with all 256 threads of the GPU box,
    # ~5 with 8-16); `cores` in the JSON reports the threads actually used, `nproc` what the host has.
    threads = min(nproc, int(os.environ.get("P2M_CPU_THREADS", "16")))
    torch.set_num_threads(threads)
    faces, graph_L, perm_rev, J = synth.make_graphs(joint_set)
    nv = int(faces.max()) + 1
    torch.manual_seed(123)
    ref_model = pose2mesh_net.get_model(J, graph_L)                 # only to get identically-shaped weights
    sd = {k: v.detach().clone() for k, v in ref_model.state_dict().items()}
    del ref_model
    params = [v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k]
    opt = torch.optim.Adam(params, lr=1e-3)
    glt = [mo.scipy_to_torch_coo(L) for L in mo.trim_graph_list(graph_L)]
    mano = joint_set == "mano"

    def make(B):
        g = torch.Generator().manual_seed(123)
        return {"B": B, "pose2d": synth.pose2d_batch(B, J), "gt_mesh": torch.randn(B, nv, 3, generator=g) * 0.3,
                "gt_j": torch.randn(B, J, 3, generator=g) * 300, "one": torch.ones(B, 1, 1)}
    Jreg = torch.from_numpy(synthetic_regressor(J, nv))

    def step(d):
        if mode == "infer":
            with torch.no_grad():
                mesh, _ = mo.flat_forward(sd, glt, d["pose2d"], mano, False)
                lo.test_epilogue(mesh, perm_rev, nv, Jreg)
            return
        opt.zero_grad()
        mesh, lift = mo.flat_forward(sd, glt, d["pose2d"], mano, True)
        loss, _ = lo.train_losses(mesh, lift, perm_rev, nv, faces, Jreg, d["gt_mesh"], d["gt_j"], d["gt_j"], d["one"],
                                  d["one"], d["one"], with_edge=edge_loss)
        loss.backward()
        opt.step()
    # calibrate on a tiny batch so that a slow host cannot blow the budget with the first big step
    d = make(2)
    step(d)
    t = time.time()
    step(d)
    per_mesh = (time.time() - t) / 2
    B = batch
    while B > 2 and per_mesh * B * 2 > budget_s:      # warm-up + >= 1 timed step must fit
        B //= 2
    d = make(B)
    tw = time.time()
    step(d)                                            # warm-up at the real batch (allocator, thread pool)
    tw = time.time() - tw
    t0 = time.time()
    n = 0
    while True:
        step(d)
        n += 1
        el = time.time() - t0
        if n >= 3 or tw + el + el / n > budget_s:
            break
    dt = time.time() - t0
    what = "train steps (fwd + 5 losses + bwd + Adam)" if mode == "train" else "eval forwards + Tester epilogue"
    return {"value": round(B * n / dt, 3), "unit": "meshes/s", "cores": threads, "kind": "port", "nproc": nproc,
            "cpu_model": cpu_model, "threads": threads, "torch": torch.__version__,
            # how much slower this port is than the real reference code for the same step (identical loss)
            **_port_ratio(),
            "sample": f"{n} {what} at batch {B} after 1 warm-up, same synthetic "
                      f"{'MANO' if mano else 'SMPL'}-like mesh, inputs and losses; oracle port of the reference CPU "
                      f"path (same torch.sparse.mm -> cat -> permute -> Linear -> BatchNorm1d sequence), "
                      f"{threads} of {nproc} host threads"}


def _port_ratio():
    """Time of the oracle port / time of the real reference code for the same train step, measured by
    tools/cpu_port_vs_reference.py where /root/reference exists and committed as profiles/cpu_port_vs_reference.json."""
    try:
        r = json.load(open(os.path.join(ROOT, "profiles", "cpu_port_vs_reference.json")))
        return {"port_over_reference_time": r["port_over_reference_time"],
                "port_over_reference_measured_at": f"batch {r['batch']}, {r['threads']} threads, {r['joint_set']}, "
                                                   f"{r['where']}"}
    except (OSError, ValueError, KeyError):
        return {"port_over_reference_time": None}


def self_launch_command(n, argv, port=None):
    """The command `python bench.py --gpus N ...` re-executes itself as when no launcher set WORLD_SIZE: torch.distributed.run,
    one node, N processes (one per GPU), rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv):
    """Runs the N-rank job and returns its exit code; rank 0's JSON line goes to this process's stdout unchanged.
    $P2M_BENCH_LAUNCH_DRYRUN=1 prints the command instead (CPU test of the launch contract)."""
    import subprocess
    cmd = self_launch_command(n, argv)
    if os.environ.get("P2M_BENCH_LAUNCH_DRYRUN") == "1":
        print(json.dumps({"self_launch": cmd}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def _traffic_for(*kernel_prefixes):
    """HBM bytes per launch of a kernel family (all template instantiations whose name contains the prefix) from the
    committed PMC passes of this command (tools/rocprof_traffic.sh -> profiles/traffic_latest.json)."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    tot, n = 0.0, 0
    for name, rec in t.get("kernels", {}).items():
        if any(p in name for p in kernel_prefixes):
            tot += rec["total_hbm_bytes"]
            n += rec["launches"]
    if n == 0:
        return None, None
    return {"hbm_bytes_per_launch": tot / n, "launches": n, "code_id": t.get("code_id")}, t.get("source")


def parity_check(step, n=4):
    """Untimed, after the timed region (rank 0, N = 1, beside the CPU baseline): the model the bench just trained, put in
    eval(), on the first `n` samples of the TIMED batch - FlatPose2Mesh forward on the GPU against the CPU oracle
    (oracle/meshnet_oracle.py: the reference's operator sequence, lib/models/pose2mesh_net.py:16-22) with the same weights
    and running statistics.  Max per-vertex L2 in metres; BASELINE's bar is 1e-4."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import meshnet_oracle as mo
    model = step.model
    was_training = model.training
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x = step.pose2d[:n].contiguous()
    try:
        model.eval()
        with torch.no_grad():
            mesh, lift = model(x)
        torch.cuda.synchronize()
    finally:
        model.train(was_training)
    glt = [mo.scipy_to_torch_coo(L) for L in mo.trim_graph_list(step.graph_L)]
    mano = step.V0 < 2000
    with torch.no_grad():
        rmesh, rlift = mo.flat_forward(sd, glt, x.cpu(), mano, False)
    e = float((mesh.cpu().double() - rmesh.double()).norm(dim=-1).max())
    el = float((lift.cpu() - rlift).abs().max() / max(1.0, float(rlift.abs().max())))
    return {"max_vertex_l2_m": e, "tol": 1e-4, "ok": bool(e <= 1e-4), "lifted_pose_rel_err": el, "samples": n,
            "arith": ops.GEMM_ARITH,
            "what": f"eval forward of the first {n} samples of the timed batch with the weights the timed steps left, GPU "
                    f"(this package, {ops.GEMM_ARITH}) vs the CPU oracle port of the reference path; untimed"}


class PowerWatch:
    """Samples the GPU's shader clock and package power (amdgpu hwmon in sysfs: freq1_input, power1_input, power1_cap)
    every `period` seconds from a side thread while the timed steps run: two small file reads per sample, no
    subprocess.  The MFMA kernels of this path run INTO the package power cap: the clock they
    sustain, not the nominal 2.4 GHz, is what their pipe peak scales with."""

    def __init__(self, device_index=0, period=0.02):
        import glob
        self.period, self.samples, self._stop, self._th = period, [], False, None
        dirs = []
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            dirs = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
        except Exception:  # noqa: BLE001  (older torch: no pci ids) -> the busiest card below
            dirs = []
        self.how = "hwmon of cuda device %d" % device_index if dirs else "hwmon of the card drawing the most power"
        self.dirs = dirs or glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def _loop(self):
        while not self._stop:
            best = None
            for d in self.dirs:
                pw = self._read(d + "/power1_input")
                if pw is None:
                    pw = self._read(d + "/power1_average")
                ck = self._read(d + "/freq1_input")
                if pw is not None and ck is not None and (best is None or pw > best[1]):
                    best = (ck / 1e6, pw / 1e6, d)
            if best is not None:
                self.samples.append(best)
            time.sleep(self.period)

    def __enter__(self):
        if self.dirs:
            import threading
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._th is not None:
            self._th.join()

    def report(self):
        if len(self.samples) < 3:
            return None
        s = self.samples[len(self.samples) // 5:]              # the first fifth: clocks still ramping
        ck = np.asarray([x[0] for x in s])
        pw = np.asarray([x[1] for x in s])
        cap = self._read(s[-1][2] + "/power1_cap")
        return {"sclk_mhz_median": round(float(np.median(ck)), 0), "sclk_mhz_min": round(float(ck.min()), 0),
                "sclk_mhz_max": round(float(ck.max()), 0), "sclk_nominal_mhz": 2400,
                "power_w_median": round(float(np.median(pw)), 0), "power_w_max": round(float(pw.max()), 0),
                "power_cap_w": None if cap is None else round(cap / 1e6, 0), "samples": len(s), "source": self.how,
                "note": "sampled every %d ms DURING the timed steps; every peak in the roofline objects is the guide's "
                        "figure at the nominal 2.4 GHz - at the median clock here the matrix-pipe peaks are "
                        "sclk_mhz_median / 2400 of that" % int(self.period * 1e3)}


def timed_run(step, warmup, steps, barrier):
    """`warmup` untimed steps, then EXACTLY `steps` steps between two barrier + synchronize pairs -> (wall seconds,
    per-step milliseconds).  The timed region is clean: no per-launch events in it, only one HIP event per STEP on the
    main stream (recorded, never waited on inside the region) from which the per-step median / spread are read."""
    for _ in range(warmup):
        step()
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    return dt, [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]


def step_stats(per_step_ms):
    a = np.sort(np.asarray(per_step_ms, dtype=np.float64))
    return {"median": round(float(np.median(a)), 3), "min": round(float(a[0]), 3), "max": round(float(a[-1]), 3),
            "p10": round(float(np.percentile(a, 10)), 3), "p90": round(float(np.percentile(a, 90)), 3),
            "note": "per-step time between HIP events recorded on the main stream after every step of the timed region "
                    "(GPU-side; `value` and `ms_per_step` are wall-clock over the whole region, as the contract asks)"}


DTYPE_NOTE = {
    "f16x2": "f32 (emulated on the FP16 matrix pipe: every operand as 2 scaled fp16 slices = 22-bit significands, 3 slice "
             "products per fp32 product, fp32 accumulate; activations, weights, gradients and optimizer state are stored "
             "in fp32)",
    "bf16x3": "f32 (emulated on the BF16 matrix pipe: every operand as 3 exact bf16 slices = 24-bit significands, 6 slice "
              "products per fp32 product, fp32 accumulate)",
    "f32": "f32 (native v_mfma_f32_32x32x2_f32)",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (train: 256, infer: 64 = BASELINE.json)")
    ap.add_argument("--joint-set", default=None, choices=["coco", "human36", "mano"])
    ap.add_argument("--optimizer", default="adam", choices=["adam", "rmsprop"])
    ap.add_argument("--no-edge-loss", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=45.0)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--stock-losses", action="store_true", help="use the stock-torch loss modules instead of p2m_mesh_loss")
    ap.add_argument("--train-graph", action="store_true",
                    help="train mode, 1 GPU: the whole step as one captured hipGraph (train.GraphedTrainStep)")
    ap.add_argument("--infer-path", default="graph", choices=["graph", "eager", "general"],
                    help="--mode infer: captured hipGraph of the real-only path (default), the same launch by launch, "
                         "or the general drop-in module + epilogue kernel")
    ap.add_argument("--also", default=None,
                    help="comma list of extra legs measured after the main one and reported under `also` (1 GPU, default "
                         "train run only): infer = BASELINE configs[1] (batch 64, J=17, eval forward + Tester epilogue, "
                         "captured hipGraph), mano = configs[4] (MANO-like, batch 512, train step).  Default: infer,mano; "
                         "`none` switches them off")
    ap.add_argument("--no-arith-ab", action="store_true",
                    help="skip the same-process A/B of the other contraction arithmetics (`arith_ab`)")
    ap.add_argument("--arith", default=None, choices=["bf16x3", "f16x2", "f32"],
                    help="contraction arithmetic of the MAIN measurement (`value`).  Default: $P2M_GEMM_ARITH if set, else "
                         "bf16x3 - the exact fp32 emulation (24-bit operands): the reference computes in fp32, so the headline "
                         "is quoted on an fp32-equivalent arithmetic; the package's fast mode f16x2 (22-bit operands, its "
                         "import-time default) is measured beside it under `arith_ab`, at the same protocol")
    args = ap.parse_args()
    main_arith = args.arith or os.environ.get("P2M_GEMM_ARITH") or "bf16x3"
    if main_arith != ops.GEMM_ARITH:
        ops.GEMM_ARITH = main_arith
        ops.bump_weight_epoch()
    infer = args.mode == "infer"
    default_run = (not infer and args.batch is None and args.joint_set is None and not args.train_graph
                   and not args.stock_losses and args.optimizer == "adam")
    if args.batch is None:
        args.batch = 64 if infer else 256
    if args.joint_set is None:
        args.joint_set = "human36" if infer else "coco"
    # >= 20 warm-up, >= 50 timed iterations, median (the line carries total-time throughput as `value`, as the
    # driver's contract asks, and the per-step median / spread from HIP events beside it)
    if args.steps is None:
        args.steps = 100 if infer else 50
    if args.warmup is None:
        args.warmup = 20
    also = [] if args.also in ("none", "") else (args.also.split(",") if args.also else (["infer", "mano"] if default_run else []))
    for a in also:
        if a not in ("infer", "mano"):
            ap.error(f"--also: unknown leg {a!r} (infer, mano, none)")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU, exactly
        # the command of the module docstring) and let the checks below - N ranks, RCCL, N distinct GPUs - run in them
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank, world, local = p2m_dist.init_from_env()
    if world != args.gpus:
        # a line that says n_gpus = N must come from N ranks: another world size than --gpus is refused instead of
        # silently measuring something else
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks with `python -m "
                         f"torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                         f"--master-port P bench.py --gpus {args.gpus} ...` (no line printed)")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    seen, functional_only = None, False
    if world > 1:
        # every rank sees every rank over the job's backend, which must be RCCL ("nccl") with one distinct GPU per rank -
        # else no line: a 1-rank fallback or two ranks on one GPU cannot pass for a scaling point.  The one exception is
        # spelled out by the caller: P2M_DIST_BACKEND=gloo (tests/test_gpu_dist.py: two ranks sharing the only GPU),
        # and the line then says `valid_scaling_point: false`.
        seen = p2m_dist.ranks_seen(device)
        functional_only = os.environ.get("P2M_DIST_BACKEND", "") not in ("", "nccl")
        problems = []
        if seen["world"] != args.gpus or sorted(seen["ranks"]) != list(range(args.gpus)):
            problems.append(f"ranks seen {seen['ranks']} (world {seen['world']}) != 0..{args.gpus - 1}")
        if not functional_only:
            if seen["backend"] != "nccl":
                problems.append(f"backend {seen['backend']!r} is not nccl (RCCL)")
            if len(set(seen["local_devices"])) != args.gpus or min(seen["local_devices"]) < 0:
                problems.append(f"local devices {seen['local_devices']}: not {args.gpus} distinct GPUs")
        if problems:
            raise SystemExit("[bench] refusing to print a line: " + "; ".join(problems))
    if infer:
        step = InferStep(device, args.batch, args.joint_set, path=args.infer_path)
    else:
        step = TrainStep(device, args.batch, args.joint_set, world, edge_loss=not args.no_edge_loss,
                         stock_losses=args.stock_losses, optimizer=args.optimizer)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graphed = infer and args.infer_path == "graph"
    eager_step = step
    train_graph = args.train_graph and not infer and world == 1 and not args.stock_losses
    if train_graph:
        from pose2mesh_release_amd import train as p2m_train
        step = p2m_train.GraphedTrainStep(eager_step.model, eager_step.opt, eager_step.loss_fn, warmup=2)
        args.warmup = max(args.warmup, 4)              # 2 eager steps, the capture, one replay before the clock starts
        graphed = True
    with PowerWatch(local) as pwatch:
        dt, per_step_ms = timed_run(step, args.warmup, args.steps, barrier)
    ksteps = min(args.steps, 10)          # steps of the per-kernel timing pass (untimed; ~1 200 HIP events per step)
    if not args.no_kernel_timing:
        # per-kernel rooflines: the SAME steps once more, launch by launch, every launch bracketed by HIP events on its
        # stream (a replayed graph cannot be bracketed node by node, so the graphed paths re-issue their launches eagerly)
        ops.TIMER = ops.KernelTimer()
        if train_graph:
            for _ in range(ksteps):
                eager_step()
        elif graphed:
            with torch.no_grad():
                for _ in range(ksteps):
                    step.step._eager()
        else:
            for _ in range(ksteps):
                step()
        torch.cuda.synchronize()
    timer_serial = None
    if not args.no_kernel_timing and not infer and world == 1:
        # ... and twice more with the weight-gradient launches on the MAIN stream (ops.DW_SIDE_STREAM off): in the step they
        # overlap the main stream's kernels, so their bracketed times above measure a shared GPU; here every launch has
        # the GPU to itself (`exclusive_serial` in the roofline objects)
        timer_overlap, ops.TIMER = ops.TIMER, ops.KernelTimer()
        side, ops.DW_SIDE_STREAM = ops.DW_SIDE_STREAM, False
        try:
            for _ in range(2):
                eager_step()
            torch.cuda.synchronize()
        finally:
            ops.DW_SIDE_STREAM = side
        timer_serial, ops.TIMER = ops.TIMER, timer_overlap
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    timer, ops.TIMER = ops.TIMER, None

    # ---- multi-GPU diagnostics (untimed, after the timed region): two more steps with HIP events around every collective
    multi = None
    if world > 1 and not infer and eager_step.reducer is not None:
        red = eager_step.reducer
        red.timing = True
        for _ in range(2):
            eager_step()
        rep = red.timing_report()
        red.timing = False
        mine = {"rank": rank, "device": local, "n_buckets": len(red.buckets), **(rep or {})}
        allrep = [None] * world
        torch.distributed.all_gather_object(allrep, mine)
        multi = {"ranks_seen": seen, "bucket_MiB": 64, "per_rank": allrep,
                 "note": "one extra, untimed step with HIP events on the communication stream around every all-reduce and on "
                         "the main stream around finish(): times in ms since the start of the step; exposed_ms = what the "
                         "main stream waited for collectives after its own backward work; hidden_frac = 1 - exposed / sum "
                         "of the collectives' durations"}

    # ---- same-process A/B of the contraction arithmetics (1 GPU, train): a fresh TrainStep per arithmetic, few steps
    arith_ab = None
    if world == 1 and not infer and not args.no_arith_ab and not train_graph:
        arith_ab = {}
        for ar in ("bf16x3", "f16x2", "f32"):
            if ar == main_arith:
                continue
            # the two matrix-pipe arithmetics at the protocol of the main leg; the native-f32 reference point briefly
            w2, k2 = (args.warmup, args.steps) if ar != "f32" else (3, 10)
            ops.GEMM_ARITH = ar
            ops.bump_weight_epoch()
            try:
                st2 = TrainStep(device, args.batch, args.joint_set, world, edge_loss=not args.no_edge_loss,
                                stock_losses=args.stock_losses, optimizer=args.optimizer)
                d2, ps2 = timed_run(st2, w2, k2, barrier)
                stt = step_stats(ps2)
                arith_ab[ar] = {"value": round(args.batch * k2 / d2, 2), "unit": "meshes/s", "steps": k2, "warmup": w2,
                                "ms_per_step_median": stt["median"], "ms_per_step_p10_p90": [stt["p10"], stt["p90"]],
                                "dtype": DTYPE_NOTE[ar]}
                del st2
            finally:
                ops.GEMM_ARITH = main_arith
                ops.bump_weight_epoch()
            torch.cuda.empty_cache()

    # ---- the other single-GPU configurations of BASELINE.json, measured by the same command (`--also`)
    also_out = {}
    if world == 1 and also:
        def also_leg(leg):
            if leg == "infer":
                st3 = InferStep(device, 64, "human36", path="graph")
                d3, ps3 = timed_run(st3, 20, 100, barrier)
                return {"metric": "SMPL meshes/sec fwd at batch 64", "value": round(64 * 100 / d3, 2), "unit": "meshes/s",
                        "steps": 100, "warmup": 20, "ms_per_step": round(1000 * d3 / 100, 3),
                        "ms_per_step_stats": step_stats(ps3), "dtype": DTYPE_NOTE[ops.GEMM_ARITH],
                        "workload": f"configs[1]: batch=64 synthetic human36 2D poses (J={st3.J}), SMPL-like hull mesh {st3.nv} "
                                    f"verts (padded {st3.V0}), FlatPose2Mesh eval forward + Tester epilogue, captured hipGraph of "
                                    f"the real-vertices-only inference path (python bench.py --mode infer)"}
            st3 = TrainStep(device, 512, "mano", 1, edge_loss=not args.no_edge_loss)
            d3, ps3 = timed_run(st3, 5, 20, barrier)
            return {"metric": "MANO meshes/sec fwd+bwd", "value": round(512 * 20 / d3, 2), "unit": "meshes/s",
                    "steps": 20, "warmup": 5, "ms_per_step": round(1000 * d3 / 20, 3),
                    "ms_per_step_stats": step_stats(ps3), "dtype": DTYPE_NOTE[ops.GEMM_ARITH],
                    "workload": f"configs[4]: batch=512 synthetic mano 2D poses (J={st3.J}), MANO-like hull mesh {st3.nv} "
                                f"verts (padded {st3.V0}), FlatPose2Mesh fwd + 5 reference losses + bwd + Adam "
                                f"(python bench.py --joint-set mano --batch 512)"}
        for leg in also:
            key = "configs[1]_infer" if leg == "infer" else "configs[4]_mano"
            also_out[key] = also_leg(leg)                    # in the arithmetic of the main measurement
            torch.cuda.empty_cache()
            if main_arith != "f16x2":                        # ... and in the package's fast mode, beside it
                ops.GEMM_ARITH = "f16x2"
                ops.bump_weight_epoch()
                try:
                    fast = also_leg(leg)
                finally:
                    ops.GEMM_ARITH = main_arith
                    ops.bump_weight_epoch()
                also_out[key]["fast_mode_f16x2"] = {k: fast[k] for k in ("value", "unit", "ms_per_step", "dtype")}
                torch.cuda.empty_cache()
    if train_graph:
        step = eager_step                    # the attributes the report reads live on the TrainStep

    if rank == 0:
        total = args.batch * world * args.steps
        mano = args.joint_set == "mano"
        if infer:
            metric = f"{'MANO' if mano else 'SMPL'} meshes/sec fwd at batch {args.batch}"
            how = {"graph": "captured hipGraph of the real-vertices-only inference path (kernel timings: same launches "
                            "issued one by one after the timed region)",
                   "eager": "real-vertices-only inference path, launch by launch",
                   "general": "general drop-in module + epilogue kernel, launch by launch"}[args.infer_path]
            workload = (f"configs[1]: batch={args.batch}/GPU synthetic {args.joint_set} 2D poses (J={step.J}), "
                        f"{'MANO' if mano else 'SMPL'}-like hull mesh {step.nv} verts (padded {step.V0}), FlatPose2Mesh "
                        f"eval forward + Tester epilogue (perm-reverse gather x1000, joint regression); {how}")
        else:
            metric = "SMPL meshes/sec fwd+bwd at batch 256" if not mano else "MANO meshes/sec fwd+bwd"
            workload = (f"configs[{4 if mano else 2}]: batch={args.batch}/GPU synthetic "
                        f"{args.joint_set} 2D poses (J={step.J}), "
                        f"{'MANO' if mano else 'SMPL'}-like hull mesh {step.nv} verts "
                        f"(padded {step.V0}), FlatPose2Mesh fwd + 5 reference losses + bwd + "
                        f"{'Adam' if args.optimizer == 'adam' else 'RMSprop'}")
        line = {
            "metric": metric,
            "value": round(total / dt, 2), "unit": "meshes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True,
            "ms_per_step_stats": step_stats(per_step_ms),
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NOTE[ops.GEMM_ARITH], "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "gemm_arith": {"bf16x3": "fp32 contraction as 3 exact bf16 slices x 6 products on the BF16 MFMA pipe, "
                                                "fp32 accumulate (error vs float64 <= native f32 MFMA)",
                                      "f16x2": "fp32 contraction as 2 fp16 slices of each operand tensor scaled by a "
                                               "power of two (device-side amax words) x 3 products on the FP16 MFMA "
                                               "pipe, fp32 accumulate: 22-bit operands, error vs float64 within 4x of the "
                                               "native f32 MFMA's (tests/test_gpu_ops.py::test_f16x2_error_is_fp32_class)",
                                      "f32": "native f32 MFMA"}[ops.GEMM_ARITH],
                       "grad_allreduce_MB": round(step.opt.numel * 4 / 1e6, 1) if (world > 1 and not infer) else 0,
                       **({"backend": seen["backend"], "local_devices": seen["local_devices"],
                           "valid_scaling_point": not functional_only} if seen is not None else {}),
                       **({"train_step": "one captured hipGraph (train.GraphedTrainStep); kernel timings from the same "
                                         "launches issued one by one after the timed region"} if train_graph else {})},
        }
        pw = pwatch.report()
        if pw is not None:
            line["power"] = pw
        # "MFMA util = meshes/s * FLOPs / 157.3e12" over the WHOLE step (all kernels, not only GEMM time)
        gf = step.dense_gflop_fwd if infer else step.dense_gflop_fwd_bwd
        line["step_dense"] = {("gflop_per_mesh_fwd" if infer else "gflop_per_mesh_fwd_bwd"): round(gf, 2),
                              "tflops": round(line["value"] * gf / 1e3 / world, 2),
                              "speedup_equivalent_frac_of_f32_mfma_peak": round(line["value"] * gf / 1e3 / world
                                                                                / PEAK_FP32_MFMA_TFLOPS, 4),
                              "note": "per GPU; algorithmic dense FLOPs of the REFERENCE network (every padding vertex "
                                      "counted, although their contractions are never executed at full width here) x "
                                      "meshes/s, against the f32 MFMA peak the reference arithmetic maps to: a "
                                      "speed-up-equivalent, NOT a utilisation -- `roofline` counts executed FLOPs"}
        if timer is not None:
            summ = timer.summary()
            summ_serial = timer_serial.summary() if timer_serial is not None else None

            def merged(*names):
                ks = [summ[n] for n in names if n in summ]
                return None if not ks else {"launches": sum(k["launches"] for k in ks), "ms": sum(k["ms"] for k in ks),
                                            "work": sum(k["work"] for k in ks),
                                            "work_alg": sum(k["work_alg"] for k in ks),
                                            "bytes": sum(k.get("bytes", 0.0) for k in ks)}
            # One roofline object per contraction family.  Each is bounded by the LARGER of (algorithmic fp32 FLOPs / matrix
            # pipe peak in those FLOPs) and (algorithmic HBM bytes / 8 TB/s) - SURVEY 8(d)'s rule; `bound` names the binding
            # resource, achieved / peak / frac refer to it, and both fractions are always listed.  `roofline` is the family
            # with the largest measured time in the step; the others follow as roofline_<family>.
            slices = {"f16x2": 3.0, "bf16x3": 6.0}.get(ops.GEMM_ARITH)
            pipe_peak = PEAK_BF16_MFMA_TFLOPS / slices if slices else PEAK_FP32_MFMA_TFLOPS
            arith_note = {"f16x2": "fp32 as 2 scaled fp16 slices, 3 x v_mfma_f32_32x32x16_f16 per product",
                          "bf16x3": "fp32 as 3 bf16 slices, 6 x v_mfma_f32_32x32x16_bf16 per product",
                          "f32": "v_mfma_f32_32x32x2_f32"}[ops.GEMM_ARITH]

            def merged_serial(*names):
                if summ_serial is None:
                    return None
                ks = [summ_serial[n] for n in names if n in summ_serial]
                return None if not ks else {"launches": sum(k["launches"] for k in ks), "ms": sum(k["ms"] for k in ks),
                                            "work": sum(k["work"] for k in ks), "bytes": sum(k.get("bytes", 0.0) for k in ks)}

            def family(rec, kernel, prefixes, fwd=None, serial=None):
                if not rec or rec["ms"] <= 0:
                    return None
                sec = rec["ms"] * 1e-3
                tf, gbs = rec["work"] / sec / 1e12, rec["bytes"] / sec / 1e9
                f_m, f_h = tf / pipe_peak, gbs / PEAK_HBM_GBPS
                tr, tr_src = _traffic_for(*prefixes)
                alg = rec["bytes"] / rec["launches"]
                cur_id = p2m_build.source_id()
                o = {"bound": "hbm" if f_h >= f_m else "mfma", "kernel": f"{kernel} ({arith_note})",
                     "achieved": round(gbs, 1) if f_h >= f_m else round(tf, 2),
                     "peak": PEAK_HBM_GBPS if f_h >= f_m else round(pipe_peak, 1),
                     "unit": "GB/s" if f_h >= f_m else "TFLOP/s", "frac": round(max(f_h, f_m), 4),
                     "traffic": None if tr is None else round(tr["hbm_bytes_per_launch"]),
                     "algorithmic_bytes_per_launch": round(alg),
                     "traffic_over_algorithmic": None if tr is None else round(tr["hbm_bytes_per_launch"] / alg, 3),
                     "mfma": {"achieved_TFLOPs": round(tf, 2), "peak": round(pipe_peak, 1), "frac": round(f_m, 4),
                              "frac_of_f32_mfma_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 4)},
                     "hbm": {"achieved_GBps": round(gbs, 1), "peak": PEAK_HBM_GBPS, "frac": round(f_h, 4),
                             "frac_of_copy_ceiling_6300": round(gbs / 6300.0, 4)},
                     "launches": rec["launches"], "avg_launch_ms": round(rec["ms"] / rec["launches"], 4),
                     "ms_per_step": round(rec["ms"] / ksteps, 3),
                     "note": "achieved = algorithmic work / HIP-event time over ALL launches of the family in the timed "
                             "steps (backward launches share the GPU with the side-stream weight gradient); matrix-pipe peak "
                             "in algorithmic fp32 FLOPs = 2500 / slice products (or the f32 MFMA's 157.3); algorithmic "
                             "bytes = every operand row and every output row once"}
                if tr is not None:
                    # the PMC passes are a separate (profiled) run: say which code they measured, and whether it is this code
                    o["traffic_code_id"], o["code_id"] = tr.get("code_id"), cur_id
                    o["traffic_stale"] = tr.get("code_id") != cur_id
                    o["traffic_note"] = (f"HBM bytes per launch (FETCH_SIZE x 2 [gfx950 correction] + WRITE_SIZE) averaged "
                                         f"over the {tr['launches']} launches of this family in the rocprofv3 --pmc passes of "
                                         f"{tr_src}; NOT measured in this run: traffic_code_id = the source hash "
                                         f"(pose2mesh_release_amd.build.source_id) of the code those passes ran, "
                                         f"traffic_stale = it differs from the code of this run")
                if fwd and fwd["ms"] > 0 and not infer:
                    fs = fwd["ms"] * 1e-3
                    o["exclusive"] = {"note": "forward launches only (nothing else on the GPU)",
                                      "mfma_frac": round(fwd["work"] / fs / 1e12 / pipe_peak, 4),
                                      "hbm_frac": round(fwd["bytes"] / fs / 1e9 / PEAK_HBM_GBPS, 4),
                                      "launches": fwd["launches"], "avg_launch_ms": round(fwd["ms"] / fwd["launches"], 4)}
                if serial and serial["ms"] > 0:
                    ss = serial["ms"] * 1e-3
                    o["exclusive_serial"] = {
                        "note": "the same launches in two extra steps with the weight gradient on the main stream: every "
                                "launch has the GPU to itself",
                        "mfma_frac": round(serial["work"] / ss / 1e12 / pipe_peak, 4),
                        "hbm_frac": round(serial["bytes"] / ss / 1e9 / PEAK_HBM_GBPS, 4),
                        "hbm_GBps": round(serial["bytes"] / ss / 1e9, 1),
                        "launches": serial["launches"], "avg_launch_ms": round(serial["ms"] / serial["launches"], 4)}
                return o
            fams = {
                "planes": family(merged("gemm_planes_mfma", "gemm_planes_mfma_bwd"),
                                 "k_gemm_planes_ws: C = [A0|A1|A2] W, planes from HBM" if slices else "k_gemm_planes",
                                 (ops.gemm_kernel_name(),), summ.get("gemm_planes_mfma"),
                                 merged_serial("gemm_planes_mfma", "gemm_planes_mfma_bwd")),
                "weight_gradient": family(merged("gemm_tn_mfma", "gemm_tn_mfma_bwd"),
                                          "k_gemm_tn_ws: P = X^T [g|Lg|L2g] over row chunks" if slices else "k_gemm_tn",
                                          ("k_gemm_tn_ws",) if slices else ("k_gemm_tn<",), None,
                                          merged_serial("gemm_tn_mfma", "gemm_tn_mfma_bwd")),
                "fused": family(merged("cheb_tile_gemm", "cheb_tile_gemm_bwd"),
                                "k_cheb_tile_gemm (VALU gather: bf16x3, N = 256) / k_cheb_mg_gemm (gather on the matrix cores: f16x2): "
                                "Chebyshev planes formed per tile on chip, no T1/T2 in HBM on the way in", ("k_cheb_mg_gemm", "k_cheb_tile_gemm"), summ.get("cheb_tile_gemm"),
                                merged_serial("cheb_tile_gemm", "cheb_tile_gemm_bwd")),
            }
            fams = {k: v for k, v in fams.items() if v is not None}
            if pw is not None and pw["sclk_mhz_median"] > 0:
                # the same fraction against the pipe peak AT THE CLOCK THE STEP RAN AT (`power`): the package power cap, not
                # the pipe, is what these kernels run into; `frac` stays the nominal-clock figure
                for v in fams.values():
                    if v.get("bound") == "mfma":
                        v["frac_at_step_sclk"] = round(v["frac"] * pw["sclk_nominal_mhz"] / pw["sclk_mhz_median"], 4)
            if fams:
                top = max(fams, key=lambda k: fams[k]["ms_per_step"])
                line["roofline"] = dict(fams[top], family=top)
                for k, v in fams.items():
                    if k != top:
                        line["roofline_" + k] = v
            sp = merged("cheb_basis_fwd", "cheb_basis_fwd_bwd", "cheb_basis_bwd", "cheb_basis_bwd_bwd")
            if sp and sp["ms"] > 0:
                # achieved = the bytes the launches have to move (real-vertex rows only: read X once at the stored
                # resolution, write T1, T2) / HIP-event time.  SURVEY 8(d)'s figure (4*K*V*Fin over ALL V rows) credits
                # rows the fake-vertex split never touches: kept as speedup_equivalent, not as a roofline fraction.
                mov = sp["work"] / (sp["ms"] * 1e-3) / 1e9
                alg = sp["work_alg"] / (sp["ms"] * 1e-3) / 1e9
                # traffic over the SAME launch set as `algorithmic_bytes_per_launch`: every basis kernel of the step
                tr, tr_src = _traffic_for("k_basis_tile", "k_basis_fwd<")
                line["roofline_sparse"] = {"bound": "hbm", "kernel": "k_basis_tile (split levels) + k_basis_fwd",
                                           "achieved": round(mov, 1),
                                           "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(mov / PEAK_HBM_GBPS, 4),
                                           "frac_of_copy_ceiling_6300": round(mov / 6300.0, 4),
                                           "traffic": None if tr is None else round(tr["hbm_bytes_per_launch"]),
                                           "algorithmic_bytes_per_launch": round(sp["work"] / sp["launches"]),
                                           "traffic_over_algorithmic": None if tr is None else round(
                                               tr["hbm_bytes_per_launch"] / (sp["work"] / sp["launches"]), 3),
                                           "launches": sp["launches"],
                                           "avg_launch_ms": round(sp["ms"] / sp["launches"], 4),
                                           "speedup_equivalent": {
                                               "GBps_if_all_V_rows_counted": round(alg, 1),
                                               "note": "SURVEY 8(d) bytes (4*K*V*Fin over ALL V rows incl. padding "
                                                       "vertices) / the same time: what an unsplit stage would need "
                                                       "to sustain to be as fast; an algorithmic saving, NOT bandwidth"},
                                           "note": "achieved = bytes the launches move (real-vertex rows; un-pooled "
                                                   "inputs read at the coarse resolution) / HIP-event time, all launches "
                                                   "(backward ones share the GPU with the side-stream weight-gradient "
                                                   "GEMMs)"}
                f = summ.get("cheb_basis_fwd")
                if f and not infer:
                    movf = f["work"] / (f["ms"] * 1e-3) / 1e9
                    line["roofline_sparse"]["exclusive"] = {
                        "note": "forward launches only (nothing else on the GPU)",
                        "achieved": round(movf, 1), "frac": round(movf / PEAK_HBM_GBPS, 4),
                        "frac_of_copy_ceiling_6300": round(movf / 6300.0, 4),
                        "launches": f["launches"], "avg_launch_ms": round(f["ms"] / f["launches"], 4)}
            line["kernel_ms_per_step"] = {k: round(v["ms"] / ksteps, 3) for k, v in summ.items()}
        if arith_ab is not None:
            arith_ab[ops.GEMM_ARITH] = {"value": line["value"], "unit": "meshes/s", "steps": args.steps,
                                        "warmup": args.warmup, "ms_per_step_median": line["ms_per_step_stats"]["median"],
                                        "dtype": DTYPE_NOTE[ops.GEMM_ARITH], "note": "the main measurement of this line"}
            line["arith_ab"] = dict(arith_ab, note="same process, same box, same synthetic batch: a fresh TrainStep per "
                                                    "contraction arithmetic; `value` of this line is bf16x3 = the exact fp32 "
                                                    "emulation (24-bit operands, what the reference's fp32 arithmetic maps to); "
                                                    "f16x2 = the package's tolerance-compliant FAST MODE (22-bit operands, "
                                                    "vertex L2 vs float64 1.4e-5 at this batch - NOT the headline), measured at "
                                                    "the same warm-up / step counts; f32 = native f32 MFMA, 3 + 10 steps (it "
                                                    "also switches the tile kernels off: they exist in the slice arithmetics "
                                                    "only)")
        if also_out:
            line["also"] = also_out
        if multi is not None:
            line["multi_gpu"] = multi
        if world == 1 and not args.no_cpu_baseline and not infer:
            line["parity_check"] = parity_check(step)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.joint_set, args.cpu_seconds, edge_loss=not args.no_edge_loss,
                                                batch=64 if infer else 32, mode=args.mode)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
