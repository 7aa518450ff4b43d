#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REAL reference (imported read-only from /root/reference
through oracle/ref_loader.py's three shims) on CPU.  Run here (the reference tree does not exist on
the GPU box):   python tests/golden/make_golden.py

Fixtures (all deterministic: numpy PCG64 weights/inputs, fixed hull meshes):
  graphs_<set>.npz   build_coarse_graphs output of the real lib/graph_utils.py: per-level CSR of L
                     (float64, as the reference returns it), perm[0], perm_reverse
  meshnet_<set>.npz  real lib/models/meshnet.Pose2Mesh: eval + train outputs, updated running stats,
                     gradient norms (+ a few full gradients) for a seeded upstream gradient
  flat_<set>.npz     real lib/models/pose2mesh_net.FlatPose2Mesh in eval(): cam_mesh, pose3d
  chebconv.npz       real graph_conv_cheby (cheby_graph_conv.py:5-42) fwd/bwd at several shapes
  loss_<set>.npz     real lib/core/loss.py classes driven exactly as lib/core/base.py:130-143 does: the five weighted
                     loss values and d(loss)/d(cam_mesh) (inputs are re-created from seeds by helpers.loss_case)
  state_<set>.npz    key / shape / checksum list of the real FlatPose2Mesh.state_dict() under torch.manual_seed(123)
                     (+ the small tensors themselves), for the strict load_state_dict test
  demo_h36m.npz      BASELINE configs[0]: demo/h36m_joint_input.npy through the real get_bbox / process_bbox /
                     j2d_processing (demo/run.py:149-160), the real FlatPose2Mesh in eval(), and the
                     perm-reverse + joint-regression epilogue (demo/run.py:169-171)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import helpers  # noqa: E402
import ref_loader  # noqa: E402
from pose2mesh_release_amd import synth  # noqa: E402

SETS = {"mano": (778, 4), "human36": (6890, 2), "coco": (6890, 2)}


def ref_graphs(joint_set):
    J, skel, flip, levels = synth.JOINT_SETS[joint_set]
    nv = SETS[joint_set][0]
    _, faces = synth.hull_mesh(nv, 0)
    _, gL, perm, rev = ref_loader.build_coarse_graphs(faces, J, skel, flip, levels)
    return faces, gL, perm, rev, J, levels


def save_graphs(joint_set):
    faces, gL, perm, rev, J, levels = ref_graphs(joint_set)
    d = {"num_levels": np.int64(len(gL)), "perm0": np.asarray(perm[0], np.int32), "perm_reverse": np.asarray(rev, np.int32),
         "faces_sha": np.frombuffer(__import__("hashlib").sha256(faces.tobytes()).digest(), np.uint8)}
    for i, L in enumerate(gL):
        L = L.tocsr()
        L.sort_indices()
        d[f"L{i}_indptr"] = L.indptr.astype(np.int32)
        d[f"L{i}_indices"] = L.indices.astype(np.int32)
        d[f"L{i}_data"] = L.data.astype(np.float64)
        d[f"L{i}_n"] = np.int64(L.shape[0])
    np.savez_compressed(os.path.join(HERE, f"graphs_{joint_set}.npz"), **d)
    return gL, J


def save_meshnet(joint_set, gL, J):
    ns = ref_loader.load("mano" if joint_set == "mano" else "human36")
    B = SETS[joint_set][1]
    mano = joint_set == "mano"
    out = {}
    x = helpers.meshnet_input(B, J, seed=7)
    gseed = 3
    for mode in ("eval", "train"):
        net = ns.meshnet.get_model(5, 3, [L.copy() for L in gL])
        sd = helpers.numpy_state(net.state_dict(), 1)
        net.load_state_dict(sd)
        net.train(mode == "train")
        xin = x.clone().requires_grad_(True)
        with ref_loader.cpu_cuda_shim():
            y = net(xin)
        g = torch.Generator().manual_seed(gseed)
        w = torch.randn(y.shape, generator=g)
        (y * w).sum().backward()
        out[f"{mode}_out"] = y.detach().numpy()
        out[f"{mode}_gin"] = xin.grad.numpy()
        names, norms = [], []
        for k, p in net.named_parameters():
            names.append(k)
            norms.append(float(p.grad.double().norm()))
            if p.numel() <= 256:
                out[f"{mode}_grad::{k}"] = p.grad.numpy()
        out[f"{mode}_grad_names"] = np.array(names)
        out[f"{mode}_grad_norms"] = np.array(norms)
        if mode == "train":
            for k, v in net.state_dict().items():
                if "running" in k:
                    out[f"train_state::{k}"] = v.numpy()
    out["B"], out["J"], out["mano"] = np.int64(B), np.int64(J), np.bool_(mano)
    np.savez_compressed(os.path.join(HERE, f"meshnet_{joint_set}.npz"), **out)


def save_flat(joint_set, gL, J):
    ns = ref_loader.load("mano" if joint_set == "mano" else "human36")
    B = SETS[joint_set][1]
    net = ns.pose2mesh_net.get_model(J, [L.copy() for L in gL])
    sd = helpers.numpy_state(net.state_dict(), 2)
    net.load_state_dict(sd)
    net.eval()
    pose2d = synth.pose2d_batch(B, J, seed=11)
    with torch.no_grad(), ref_loader.cpu_cuda_shim():
        mesh, pose3d = net(pose2d)
    np.savez_compressed(os.path.join(HERE, f"flat_{joint_set}.npz"), cam_mesh=mesh.numpy(), pose3d=pose3d.numpy(),
                        B=np.int64(B), J=np.int64(J))


def save_chebconv(gL_mano):
    ns = ref_loader.load("mano")
    out = {}
    cases = [(2, 128, 128, True), (3, 64, 128, True), (4, 32, 64, True), (-1, 5, 32, True), (1, 64, 3, False)]
    for ci, (lvl, Fin, Fout, has_bn) in enumerate(cases):
        L = gL_mano[lvl]
        V = L.shape[0]
        B = 2
        rng = np.random.default_rng(100 + ci)
        x = torch.from_numpy(rng.standard_normal((B, V, Fin)).astype(np.float32)).requires_grad_(True)
        cl = torch.nn.Linear(Fin * 3, Fout)
        bn = torch.nn.BatchNorm1d(Fout) if has_bn else None
        with torch.no_grad():
            cl.weight.copy_(torch.from_numpy(rng.uniform(-0.1, 0.1, cl.weight.shape).astype(np.float32)))
            cl.bias.copy_(torch.from_numpy(rng.uniform(-0.1, 0.1, cl.bias.shape).astype(np.float32)))
            if bn is not None:
                bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, (Fout,)).astype(np.float32)))
                bn.bias.copy_(torch.from_numpy(rng.uniform(-0.2, 0.2, (Fout,)).astype(np.float32)))
        Lt = ns.graph_utils.sparse_python_to_torch(L)
        y = ns.cheby.graph_conv_cheby(x, cl, bn, Lt, Fout, 3)
        w = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
        (y * w).sum().backward()
        out[f"c{ci}_meta"] = np.array([lvl, Fin, Fout, int(has_bn), B], np.int64)
        out[f"c{ci}_y"] = y.detach().numpy()
        out[f"c{ci}_dx"] = x.grad.numpy()
        out[f"c{ci}_dW"] = cl.weight.grad.numpy()
        out[f"c{ci}_db"] = cl.bias.grad.numpy()
        if bn is not None:
            out[f"c{ci}_dgamma"] = bn.weight.grad.numpy()
            out[f"c{ci}_dbeta"] = bn.bias.grad.numpy()
            out[f"c{ci}_rm"] = bn.running_mean.numpy()
            out[f"c{ci}_rv"] = bn.running_var.numpy()
    np.savez_compressed(os.path.join(HERE, "chebconv.npz"), **out)


def real_regressor():
    """data/Human36M/J_regressor_h36m_correct.npy: the one regressor the reference ships ((17, 6890) float64, 107 nnz)."""
    R = np.load(os.path.join(ref_loader.REF_ROOT, "data", "Human36M", "J_regressor_h36m_correct.npy"))
    assert R.shape == (17, 6890)
    return R.astype(np.float32)


def regressor_triplets(R):
    r, c = np.nonzero(R)
    return {"jreg_rows": r.astype(np.int32), "jreg_cols": c.astype(np.int32), "jreg_vals": R[r, c].astype(np.float32),
            "jreg_shape": np.asarray(R.shape, np.int64)}


def save_loss(joint_set):
    """The reference's loss classes, called in the order and with the weights of lib/core/base.py:130-143.
    human36: with the reference's REAL joint regressor (embedded in the fixture), the others with a synthetic one."""
    L = ref_loader.load_loss()
    jreg = real_regressor() if joint_set == "human36" else None
    c = helpers.loss_case(joint_set, jreg=jreg)
    losses = L.get_loss(c["faces"])                                        # base.py:60 / loss.py:117-120
    out = {}
    for tag, with_edge in (("edge", True), ("noedge", False)):
        cam = c["cam_mesh"].clone().requires_grad_(True)
        lift = c["lift_pose"].clone().requires_grad_(True)
        with ref_loader.cpu_cuda_shim():
            pred_mesh = cam[:, c["perm_reverse"][:c["nv"]], :]                                     # base.py:130
            pred_pose = torch.matmul(c["J_regressor"][None, :, :], pred_mesh * 1000)               # base.py:131
            l1 = losses[0](pred_mesh, c["gt_mesh"], c["val_mesh"])                                 # base.py:134-137
            l2 = 1e-1 * losses[1](pred_mesh, c["gt_mesh"])
            l4 = 1e-3 * losses[3](pred_pose, c["gt_reg3dpose"], c["val_reg3dpose"])
            l5 = 1e-3 * losses[4](lift, c["gt_lift3dpose"], c["val_lift3dpose"])
            loss = l1 + l2 + 0 + l4 + l5                                                           # base.py:138-139
            l3 = torch.zeros(())
            if with_edge:                                                                          # base.py:141-143
                l3 = 20 * losses[2](pred_mesh, c["gt_mesh"])
                loss = loss + l3
        loss.backward()
        out[f"{tag}_losses"] = np.array([float(v) for v in (l1, l2, l3, l4, l5)], np.float64)
        out[f"{tag}_total"] = np.float64(float(loss))
        out[f"{tag}_grad_cam"] = cam.grad.numpy()
        out[f"{tag}_grad_lift"] = lift.grad.numpy()
    out["cam_sha"] = np.frombuffer(__import__("hashlib").sha256(c["cam_mesh"].numpy().tobytes()).digest(), np.uint8)
    if jreg is not None:
        out.update(regressor_triplets(jreg))
    np.savez_compressed(os.path.join(HERE, f"loss_{joint_set}.npz"), **out)


def save_state_spec(joint_set, gL, J):
    """Key / shape inventory of the REAL reference module's state dict (lib/models/pose2mesh_net.py:25-28 under the
    reference's default seed, main/train.py:12), so that load_state_dict(strict=True) of the new module is tested
    against the real key set; tensors with <= 4096 elements are stored, larger ones as (sum, abs-sum) checksums."""
    ns = ref_loader.load("mano" if joint_set == "mano" else "human36")
    torch.manual_seed(123)
    net = ns.pose2mesh_net.get_model(J, [L.copy() for L in gL])
    sd = net.state_dict()
    keys = list(sd.keys())
    out = {"keys": np.array(keys), "shapes": np.array([",".join(str(d) for d in v.shape) for v in sd.values()]),
           "dtypes": np.array([str(v.dtype) for v in sd.values()]),
           "num_params": np.int64(sum(p.numel() for p in net.parameters()))}
    for i, (k, v) in enumerate(sd.items()):
        if v.numel() <= 4096:
            out[f"t{i}"] = v.numpy()
        else:
            out[f"c{i}"] = np.array([float(v.double().sum()), float(v.double().abs().sum())])
    np.savez_compressed(os.path.join(HERE, f"state_{joint_set}.npz"), **out)


def save_demo(gL, rev):
    """configs[0]: the demo's single-pose path with the real preprocessing functions and the real model."""
    aug = ref_loader.load_aug()
    ns = ref_loader.load("human36")
    cfgm = ns.cfg.MODEL
    joint_input = np.load(os.path.join(ref_loader.REF_ROOT, "demo", "h36m_joint_input.npy"))
    bbox = aug.coord_utils.get_bbox(joint_input)                                                  # run.py:150
    bbox2 = aug.coord_utils.process_bbox(bbox.copy())                                             # run.py:152
    joint_img, trans = aug.aug_utils.j2d_processing(joint_input.copy(), (cfgm.input_shape[1], cfgm.input_shape[0]),
                                                    bbox2, 0, 0, None)                            # run.py:154
    joint_img = joint_img[:, :2]                                                                  # run.py:156-159
    joint_img /= np.array([[cfgm.input_shape[1], cfgm.input_shape[0]]])
    mean, std = np.mean(joint_img, axis=0), np.std(joint_img, axis=0)
    joint_img = (joint_img.copy() - mean) / std
    x = torch.Tensor(joint_img[None, :, :])                                                       # run.py:160
    J, nv = 17, 6890
    net = ns.pose2mesh_net.get_model(J, [L.copy() for L in gL])
    net.load_state_dict(helpers.numpy_state(net.state_dict(), 2))
    net.eval()                                                                                    # run.py:166
    with torch.no_grad(), ref_loader.cpu_cuda_shim():
        pred_mesh, pose3d = net(x)                                                                # run.py:167
        mesh = pred_mesh[:, rev[:nv], :]                                                          # run.py:170
        R = real_regressor()                          # demo/run.py:127 loads exactly this file (mesh_model.joint_regressor_h36m)
        joints = torch.matmul(torch.from_numpy(R), mesh)                                          # run.py:171
    np.savez_compressed(os.path.join(HERE, "demo_h36m.npz"), joint_input=joint_input, bbox=bbox, bbox2=bbox2,
                        trans=trans, model_input=x.numpy(), cam_mesh=pred_mesh.numpy(), pose3d=pose3d.numpy(),
                        mesh=mesh.numpy(), joints=joints.numpy(), **regressor_triplets(R))


if __name__ == "__main__":
    assert ref_loader.available(), "reference tree missing"
    torch.set_num_threads(os.cpu_count() or 1)
    if sys.argv[1:] == ["regressor-fixtures"]:         # only the two fixtures that carry the reference's real regressor
        save_loss("human36")
        _, gLh, _, revh, _, _ = ref_graphs("human36")
        save_demo(gLh, revh)
        sys.exit(0)
    graphs = {}
    for js in ("mano", "human36", "coco"):
        graphs[js] = save_graphs(js)
        print("graphs", js, [L.shape[0] for L in graphs[js][0]])
    for js in ("mano", "human36"):
        save_meshnet(js, *graphs[js])
        print("meshnet", js)
    save_flat("mano", *graphs["mano"])
    save_flat("coco", *graphs["coco"])
    save_chebconv(graphs["mano"][0])
    for js in ("mano", "coco", "human36"):
        save_loss(js)
        print("loss", js)
    for js in ("mano", "human36", "coco"):
        save_state_spec(js, *graphs[js])
        print("state", js)
    _, gLh, _, revh, _, _ = ref_graphs("human36")
    save_demo(gLh, revh)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
