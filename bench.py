#!/usr/bin/env python3
"""bench.py -- SMPL meshes/sec, forward + backward + Adam, batch 256 per GPU (BASELINE.json configs[2]).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one full reference train step (lib/core/base.py:122-148) on one synthetic batch that is
already resident in HBM: FlatPose2Mesh forward (PoseNet + coarse-to-fine GCN), perm-reverse gather,
joint regression, the five reference losses, backward, [gradient all-reduce], Adam.  Weak scaling:
every rank owns `--batch` samples.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant kernel (the dense contraction k_gemm_planes[_bx]): algorithmic fp32 FLOPs of its
                launches / their HIP-event time, against the MFMA peak of the pipe it runs on expressed in
                algorithmic FLOPs (bf16x3: 2500 / 6 = 416.7 TFLOP/s; P2M_GEMM_ARITH=f32: 157.3 TFLOP/s)
  roofline_sparse  the Chebyshev-basis gather kernels (HBM-bound): algorithmic bytes / event time vs 8 TB/s
  cpu_baseline  the oracle port of the reference CPU path (oracle/meshnet_oracle.py), same train step,
                small batch, timed on this host's cores (rank 0, N=1 only)
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

from pose2mesh_release_amd import dist as p2m_dist  # noqa: E402
from pose2mesh_release_amd import loss as p2m_loss  # noqa: E402
from pose2mesh_release_amd import ops, optim, pose2mesh_net, synth  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, dense FP32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide, dense BF16 matrix peak; the bf16x3 contraction spends 6 bf16 MFMA flops
                                  # per algorithmic fp32 flop -> 416.7 TFLOP/s fp32-equivalent
PEAK_HBM_GBPS = 8000.0            # HBM3E spec peak (achievable copy ceiling ~6300 GB/s)


synthetic_regressor = synth.synthetic_regressor


class TrainStep:
    """The reference train step (lib/core/base.py:122-148) on resident synthetic data."""

    def __init__(self, device, B, joint_set, world, edge_loss=True, seed=123, stock_losses=False):
        self.device, self.B = device, B
        faces, graph_L, perm_rev, J = synth.make_graphs(joint_set)
        self.J, self.nv = J, int(faces.max()) + 1
        torch.manual_seed(seed)                                     # main/train.py:12
        self.model = pose2mesh_net.get_model(J, graph_L).to(device).train()
        self.opt = optim.FlatAdam(self.model.parameters(), lr=1e-3)          # funcs_utils.py:92-96
        self.reducer = p2m_dist.BucketedAllReduce(self.opt.params, self.opt.offsets, self.opt.flat_grad) \
            if world > 1 else None
        self.losses = p2m_loss.get_loss(faces)
        self.edge_loss = edge_loss
        self.stock_losses = stock_losses
        self.mesh_loss = p2m_loss.FusedMeshLoss(faces, perm_rev, synthetic_regressor(J, int(faces.max()) + 1),
                                                w_normal=1e-1, w_edge=20.0 if edge_loss else 0.0, w_joint=1e-3)
        self.perm = torch.as_tensor(np.asarray(perm_rev)[:self.nv], dtype=torch.long, device=device)
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
        # algorithmic dense FLOPs per mesh, forward = sum over convs 2*V*(K*Fin)*Fout + the fc lift;
        # forward + backward = 3x (dX and dW cost one forward each)
        mesh = self.model.pose2mesh
        fwd = sum(2.0 * mesh.graph_L[L.graph].shape[0] * 3 * L.Fin * L.Fout for L in mesh._layers)
        fwd += 2.0 * mesh.fc.in_features * mesh.fc.out_features
        self.dense_gflop_fwd_bwd = 3.0 * fwd / 1e9

    def __call__(self):
        m = self.model
        self.opt.zero_grad()
        pred_mesh, lift_pose = m(self.pose2d)
        if not self.stock_losses:
            # base.py:130-143 (gather, J-regression, vertex/normal/edge/joint losses) in one fused HIP call
            mesh_total, _ = self.mesh_loss(pred_mesh, self.gt_mesh, self.gt_reg, self.one, self.one)
            lift = 1e-3 * self.losses[4](lift_pose, self.gt_lift, self.one)
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


def cpu_baseline(joint_set, budget_s, edge_loss=True):
    """Oracle port of the reference CPU path, same train step, on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import meshnet_oracle as mo
    # torch's CPU sparse/BN kernels stop scaling (and then collapse) well before a 256-core host is full:
    # 16 threads is the sweet spot measured for this path; `cores` in the JSON reports what was used.
    ncores = min(os.cpu_count() or 1, int(os.environ.get("P2M_CPU_THREADS", "16")))
    torch.set_num_threads(ncores)
    faces, graph_L, perm_rev, J = synth.make_graphs(joint_set)
    nv = int(faces.max()) + 1
    torch.manual_seed(123)
    ref_model = pose2mesh_net.get_model(J, graph_L)                 # only to get identically-shaped weights
    sd = {k: v.detach().clone() for k, v in ref_model.state_dict().items()}
    del ref_model
    params = [v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k]
    opt = torch.optim.Adam(params, lr=1e-3)
    glt = [mo.scipy_to_torch_coo(L) for L in mo.trim_graph_list(graph_L)]
    losses = p2m_loss.get_loss(faces)
    B = 4
    perm = torch.as_tensor(np.asarray(perm_rev)[:nv], dtype=torch.long)
    Jreg = torch.from_numpy(synthetic_regressor(J, nv))
    g = torch.Generator().manual_seed(123)
    pose2d = synth.pose2d_batch(B, J)
    gt_mesh = torch.randn(B, nv, 3, generator=g) * 0.3
    gt_j = torch.randn(B, J, 3, generator=g) * 300
    one = torch.ones(B, 1, 1)

    def step():
        opt.zero_grad()
        mesh, lift = mo.flat_forward(sd, glt, pose2d, joint_set == "mano", True)
        mesh = mesh[:, perm, :]
        pose = torch.matmul(Jreg[None], mesh * 1000)
        loss = losses[0](mesh, gt_mesh, one) + 1e-1 * losses[1](mesh, gt_mesh) + 1e-3 * losses[3](pose, gt_j, one) \
            + 1e-3 * losses[4](lift, gt_j, one)
        if edge_loss:
            loss = loss + 20 * losses[2](mesh, gt_mesh)
        loss.backward()
        opt.step()
    tw = time.time()
    step()                                    # warm-up (allocator, thread pool); also calibrates the budget
    tw = time.time() - tw
    t0 = time.time()
    n = 0
    while True:
        step()
        n += 1
        if time.time() - t0 + tw > budget_s or n >= 8:
            break
    dt = time.time() - t0
    return {"value": round(B * n / dt, 3), "unit": "meshes/s", "cores": ncores, "kind": "port",
            "sample": f"{n} train steps (fwd+bwd+Adam) at batch {B}, same synthetic SMPL-like mesh and losses, "
                      f"torch {torch.__version__} CPU, {ncores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="samples per GPU (BASELINE.json: 256)")
    ap.add_argument("--joint-set", default="coco", choices=["coco", "human36", "mano"])
    ap.add_argument("--no-edge-loss", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--stock-losses", action="store_true", help="use the stock-torch loss modules instead of p2m_mesh_loss")
    args = ap.parse_args()

    rank, world, local = p2m_dist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} != --gpus {args.gpus}: using WORLD_SIZE", file=sys.stderr)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    step = TrainStep(device, args.batch, args.joint_set, world, edge_loss=not args.no_edge_loss,
                     stock_losses=args.stock_losses)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if not args.no_kernel_timing:
        ops.TIMER = ops.KernelTimer()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    timer, ops.TIMER = ops.TIMER, None

    if rank == 0:
        total = args.batch * world * args.steps
        line = {
            "metric": "SMPL meshes/sec fwd+bwd at batch 256" if args.joint_set != "mano"
                      else "MANO meshes/sec fwd+bwd",
            "value": round(total / dt, 2), "unit": "meshes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"configs[{4 if args.joint_set == 'mano' else 2}]: batch={args.batch}/GPU synthetic "
                                    f"{args.joint_set} 2D poses (J={step.J}), "
                                    f"{'MANO' if args.joint_set == 'mano' else 'SMPL'}-like hull mesh {step.nv} verts "
                                    f"(padded {step.V0}), FlatPose2Mesh fwd + 5 reference losses + bwd + Adam"),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "gemm_arith": ("fp32 contraction as 3 exact bf16 slices x 6 products on the BF16 MFMA pipe, "
                                      "fp32 accumulate (error vs float64 <= native f32 MFMA)"
                                      if ops.GEMM_ARITH == "bf16x3" else "native f32 MFMA"),
                       "grad_allreduce_MB": round(step.opt.numel * 4 / 1e6, 1) if world > 1 else 0},
        }
        # "MFMA util = meshes/s * FLOPs / 157.3e12" over the WHOLE step (all kernels, not only GEMM time)
        line["step_dense"] = {"gflop_per_mesh_fwd_bwd": round(step.dense_gflop_fwd_bwd, 2),
                              "tflops": round(line["value"] * step.dense_gflop_fwd_bwd / 1e3 / world, 2),
                              "frac_of_f32_mfma_peak": round(line["value"] * step.dense_gflop_fwd_bwd / 1e3 / world
                                                             / PEAK_FP32_MFMA_TFLOPS, 4),
                              "note": "per GPU; algorithmic dense FLOPs of the reference network (fake vertices "
                                      "included) x meshes/s, against the f32 MFMA peak the reference arithmetic maps to"}
        if timer is not None:
            summ = timer.summary()

            def merged(*names):
                ks = [summ[n] for n in names if n in summ]
                return None if not ks else {"launches": sum(k["launches"] for k in ks), "ms": sum(k["ms"] for k in ks),
                                            "work": sum(k["work"] for k in ks),
                                            "work_alg": sum(k["work_alg"] for k in ks)}
            g = merged("gemm_planes_mfma", "gemm_planes_mfma_bwd")
            if g:
                if ops.GEMM_ARITH == "bf16x3":
                    kname = ("k_gemm_planes_ws" if os.environ.get("P2M_GEMM_WS", "2") != "0" else "k_gemm_planes_bx") + \
                        " (fp32 as 3 bf16 slices, 6 x v_mfma_f32_32x32x16_bf16 per product)"
                    peak = PEAK_BF16_MFMA_TFLOPS / 6.0
                else:
                    kname, peak = "k_gemm_planes (v_mfma_f32_32x32x2_f32)", PEAK_FP32_MFMA_TFLOPS
                ach = g["work"] / (g["ms"] * 1e-3) / 1e12
                line["roofline"] = {"bound": "mfma", "kernel": kname,
                                    "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                                    "frac": round(ach / peak, 4), "traffic": None,
                                    "launches": g["launches"], "avg_launch_ms": round(g["ms"] / g["launches"], 4),
                                    "note": "achieved = algorithmic fp32 FLOPs / HIP-event time; peak = pipe peak "
                                            "in algorithmic fp32 FLOPs (BF16 dense 2500 / 6 slice products, or the "
                                            "f32 MFMA 157.3)",
                                    "frac_of_f32_mfma_peak": round(ach / PEAK_FP32_MFMA_TFLOPS, 4)}
                f = summ.get("gemm_planes_mfma")       # forward launches: no other kernel shares the GPU with them
                if f:
                    achf = f["work"] / (f["ms"] * 1e-3) / 1e12
                    line["roofline"]["exclusive"] = {
                        "note": "forward launches only; backward launches overlap the side-stream k_gemm_tn",
                        "achieved": round(achf, 2), "frac": round(achf / peak, 4),
                        "frac_of_f32_mfma_peak": round(achf / PEAK_FP32_MFMA_TFLOPS, 4),
                        "launches": f["launches"], "avg_launch_ms": round(f["ms"] / f["launches"], 4)}
            sp = merged("cheb_basis_fwd", "cheb_basis_fwd_bwd", "cheb_basis_bwd", "cheb_basis_bwd_bwd")
            if sp and sp["ms"] > 0:
                # achieved = SURVEY.md 8(d)'s algorithmic bytes of the stage (4*K*V*Fin per sample and conv, ALL V rows)
                # / HIP-event time; bytes_moved = what the launches really have to move after the fake-vertex split
                # (41 % of the finest level's rows are isolated padding vertices whose planes are folded into W)
                ach = sp["work_alg"] / (sp["ms"] * 1e-3) / 1e9
                mov = sp["work"] / (sp["ms"] * 1e-3) / 1e9
                line["roofline_sparse"] = {"bound": "hbm", "kernel": "k_basis_tile (split levels) + k_basis_fwd",
                                           "achieved": round(ach, 1),
                                           "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBPS, 4),
                                           "frac_of_copy_ceiling_6300": round(ach / 6300.0, 4),
                                           "bytes_moved_rate": round(mov, 1), "traffic": None,
                                           "note": "achieved = SURVEY 8(d) algorithmic bytes (4*K*V*Fin, all V rows) / "
                                                   "HIP-event time; bytes_moved_rate counts only the real-vertex rows "
                                                   "the launches touch after the fake-vertex split"}
                f = summ.get("cheb_basis_fwd")
                if f:
                    achf = f["work_alg"] / (f["ms"] * 1e-3) / 1e9
                    line["roofline_sparse"]["exclusive"] = {
                        "note": "forward launches only (nothing else on the GPU)",
                        "achieved": round(achf, 1), "frac": round(achf / PEAK_HBM_GBPS, 4),
                        "frac_of_copy_ceiling_6300": round(achf / 6300.0, 4),
                        "bytes_moved_rate": round(f["work"] / (f["ms"] * 1e-3) / 1e9, 1)}
            line["kernel_ms_per_step"] = {k: round(v["ms"] / args.steps, 3) for k, v in summ.items()}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.joint_set, args.cpu_seconds, edge_loss=not args.no_edge_loss)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
