#!/usr/bin/env python3
"""Times the oracle PORT (oracle/meshnet_oracle.py + oracle/loss_oracle.py, what bench.py's cpu_baseline leg runs on the
GPU box) against the REAL reference (/root/reference through oracle/ref_loader.py) on the same host, same inputs:
forward+backward+Adam train step, SMPL-like mesh.  Only runs where /root/reference exists (this container).
The measured ratio is written to profiles/cpu_port_vs_reference.json (tracked); bench.py's cpu_baseline object reads it as
`port_over_reference_time` instead of carrying a literal.
usage: python tools/cpu_port_vs_reference.py [B] [threads] [joint_set]"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "oracle"), os.path.join(R, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import loss_oracle as lo  # noqa: E402
import meshnet_oracle as mo  # noqa: E402
import ref_loader  # noqa: E402
from pose2mesh_release_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
torch.set_num_threads(threads)
joint_set = sys.argv[3] if len(sys.argv) > 3 else "coco"      # bench.py's default train workload (configs[2])
faces, graph_L, perm_rev, J = synth.make_graphs(joint_set)
nv = 6890
ns = ref_loader.load(joint_set)
L = ref_loader.load_loss()
torch.manual_seed(123)
net = ns.pose2mesh_net.get_model(J, [g.copy() for g in graph_L])
for m in net.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
g = torch.Generator().manual_seed(123)
pose2d = synth.pose2d_batch(B, J)
gt_mesh = torch.randn(B, nv, 3, generator=g) * 0.3
gt_j = torch.randn(B, J, 3, generator=g) * 300
one = torch.ones(B, 1, 1)
Jreg = torch.from_numpy(synth.synthetic_regressor(J, nv))
perm = torch.as_tensor(np.asarray(perm_rev)[:nv], dtype=torch.long)

# ---- real reference: lib/core/base.py:122-148
net.train()
opt_r = torch.optim.Adam(net.parameters(), lr=1e-3)
losses = L.get_loss(faces)


def ref_step():
    with ref_loader.cpu_cuda_shim():
        pred_mesh, lift = net(pose2d)
        pm = pred_mesh[:, perm, :]
        pp = torch.matmul(Jreg[None], pm * 1000)
        loss = losses[0](pm, gt_mesh, one) + 1e-1 * losses[1](pm, gt_mesh) + 1e-3 * losses[3](pp, gt_j, one) \
            + 1e-3 * losses[4](lift, gt_j, one) + 20 * losses[2](pm, gt_mesh)
    opt_r.zero_grad()
    loss.backward()
    opt_r.step()
    return float(loss)


# ---- port
params = [v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k]
opt_p = torch.optim.Adam(params, lr=1e-3)
glt = [mo.scipy_to_torch_coo(g_) for g_ in mo.trim_graph_list(graph_L)]


def port_step():
    opt_p.zero_grad()
    mesh, lift = mo.flat_forward(sd, glt, pose2d, False, True)
    loss, _ = lo.train_losses(mesh, lift, perm_rev, nv, faces, Jreg, gt_mesh, gt_j, gt_j, one, one, one, with_edge=True)
    loss.backward()
    opt_p.step()
    return float(loss)


def bench_interleaved(fa, fb, n=3):
    """Steps of the two implementations ALTERNATE: timing one after the other charged the second one with the first
    one's allocator / page-cache state (round 2's 1.33x and a first 2.03x at B=32 were that artefact, not the code)."""
    la, lb = fa(), fb()                     # warm-up (and the first-step losses)
    ta = tb = 0.0
    for _ in range(n):
        t = time.time()
        fa()
        ta += time.time() - t
        t = time.time()
        fb()
        tb += time.time() - t
    return ta / n, la, tb / n, lb


tr, lr, tp, lp = bench_interleaved(ref_step, port_step)
out = {"port_over_reference_time": round(tp / tr, 4), "batch": B, "threads": threads, "nproc": os.cpu_count(),
       "joint_set": joint_set, "reference_s_per_step": round(tr, 3), "port_s_per_step": round(tp, 3),
       "first_step_loss_reference": lr, "first_step_loss_port": lp, "torch": torch.__version__,
       "where": "build container (the only place /root/reference exists)", "order": "steps interleaved (ref, port, ...)",
       "command": "python tools/cpu_port_vs_reference.py " + " ".join(sys.argv[1:])}
with open(os.path.join(R, "profiles", "cpu_port_vs_reference.json"), "w") as f:
    json.dump(out, f, indent=1)
print(f"B={B} threads={threads} nproc={os.cpu_count()}: real reference {tr:.3f} s/step ({B / tr:.2f} meshes/s), "
      f"oracle port {tp:.3f} s/step ({B / tp:.2f} meshes/s), port/reference time ratio {tp / tr:.3f}; "
      f"first-step loss reference {lr:.6f} port {lp:.6f}")
