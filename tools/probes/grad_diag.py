"""Gradient-parity diagnosis: HIP vs the fp32 oracle vs the fp64 oracle, per tensor (rel-L2).
python tools/probes/grad_diag.py [joint_set] [B] [mode]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "oracle"), os.path.join(R, "tests")]
import torch  # noqa: E402

import helpers  # noqa: E402
import meshnet_oracle as mo  # noqa: E402
import _child_meshnet_run as child  # noqa: E402

js = sys.argv[1] if len(sys.argv) > 1 else "mano"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else "train"
torch.set_num_threads(16)
hip = child.run(js, B, mode, 21, 99, 5, keep_on_gpu=True)
gL, _, _ = helpers.golden_graphs(js)
J = int(gL[-1].shape[0])
mano = js == "mano"
sd = helpers.numpy_state(mo.init_state(J, mo.trim_graph_list(gL), mano), 21)
x = helpers.meshnet_input(B, J, seed=99)
glt = helpers.oracle_graphs(gL)
o32, g32, _ = helpers.oracle_run(sd, glt, x, mano, mode == "train", grad_seed=5)
sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
for k, v in sd64.items():
    if v.dtype.is_floating_point and "running" not in k:
        v.requires_grad_(True)
x64 = x.double().requires_grad_(True)
o64 = mo.meshnet_forward(sd64, [g.double() for g in glt], x64, mano, mode == "train")
w = torch.randn(o64.shape, generator=torch.Generator().manual_seed(5)).double()
(o64 * w).sum().backward()
g64 = {k: v.grad for k, v in sd64.items() if v.requires_grad}
g64["__input__"] = x64.grad
print(f"{js} B={B} {mode}: fwd max vertex L2 hip-o32 {helpers.max_vertex_l2(hip['out'].cpu(), o32):.2e} "
      f"hip-o64 {helpers.max_vertex_l2(hip['out'].cpu(), o64.detach()):.2e} o32-o64 {helpers.max_vertex_l2(o32, o64.detach()):.2e}")
print(f"{'tensor':<16}{'hip-o32':>10}{'hip-o64':>10}{'o32-o64':>10}   |ref|")
for k, r64 in g64.items():
    h = hip["grad::" + k].cpu()
    print(f"{k:<16}{helpers.rel_l2(h, g32[k]):>10.2e}{helpers.rel_l2(h, r64):>10.2e}{helpers.rel_l2(g32[k], r64):>10.2e}"
          f"   {float(r64.norm()):.3e}")
