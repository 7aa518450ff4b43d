"""Phase times of k_cheb_tile_gemm from inside the kernel (probe build -DP2M_TILE_TRACE=<block>): s_memtime stamps of one
producer wave and one MFMA wave at the phase boundaries of every unit of one block (csrc/chebtile.hip P2M_TRC).

    bash tools/tile_trace.sh build      (here)        ->  lib/abl/libp2m_hip_TRACE.so
    bash tools/tile_trace.sh run        (GPU box)     ->  prints the per-unit phase table

PROBE_CASE=level,Ka,N,plan (default 0,128,128,0), PROBE_FORM=fwd|planes."""
import ctypes
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pose2mesh_release_amd import _lib, ops, synth  # noqa: E402

ops.TILE_GEMM = True
lvl, Ka, N, plan = (int(v) for v in os.environ.get("PROBE_CASE", "0,128,128,0").split(","))
form = os.environ.get("PROBE_FORM", "fwd")
B = int(os.environ.get("PROBE_B", "256"))
_, gL, _, J = synth.make_graphs("human36")
g = ops.DeviceGraph(gL[lvl], "cuda:0")
shift = 1 if plan == 1 else 0
X = torch.randn(B * (g.V >> shift), Ka, device="cuda")
W = torch.randn(3 * Ka, N, device="cuda") / (3 * Ka) ** 0.5
Bx = ops.weight_split(W)
C = torch.empty(B * g.V, N, device="cuda")
xa = ops.amax_of(X)
kw = dict(want_planes=True) if form == "planes" else dict(stats=True)
for _ in range(3):
    ops.cheb_tile_gemm(g, plan, X, X, Ka, Bx, None, None, C, N, B, amax=xa, **kw)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["P2M_HIP_LIB"])
buf = np.zeros((2, 40, 8), dtype=np.uint64)
rc = lib.p2m_tile_trace_dump(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
p, m = buf[0].astype(np.int64), buf[1].astype(np.int64)
nu = int((p[:, 0] > 0).sum())
print(f"case level={lvl} Ka={Ka} N={N} plan={plan} form={form} arith={ops.GEMM_ARITH}: {nu} units traced")
V2 = os.environ.get("P2M_TILE_V2", "0") == "1" and N <= 128 and ops.GEMM_ARITH == "bf16x3"
if V2:       # k_cheb_tile_gemm_v2 (round 6): stamps 0 head, 1 half 0 gathered, 2 past MID, 3 half 1 gathered, 4 past B2, 5 image stored, 6 past B1
    print("producer wave:  unit | gather + split h0 | wait MID | gather + split h1 | wait B2 | (copy-out of the closed group +) "
          "image store | wait B1 | unit total")
else:        # k_cheb_tile_gemm since round 6 (two barriers per unit): 0 head, 1 gathered, 2 past B2, 3 image stored, 4 xs stored + loads issued, 5 past B1
    print("producer wave:  unit | gather+split | wait B2 | (copy-out of the closed group +) image store | xs store + loads | wait B1 | "
          "- | unit total")
tot = np.zeros(7)
for w in range(nu):
    d = [p[w, 1] - p[w, 0], p[w, 2] - p[w, 1], p[w, 3] - p[w, 2], p[w, 4] - p[w, 3], p[w, 5] - p[w, 4], p[w, 6] - p[w, 5]]
    nxt = (p[w + 1, 0] - p[w, 0]) if w + 1 < nu else (p[w, 6] - p[w, 0])
    print(f"   {w:3d} | " + " | ".join(f"{int(v):7d}" for v in d) + f" | {int(nxt):7d}")
    if 1 <= w < nu - 1:
        tot += np.array(d + [nxt])
if nu > 2:
    print("   avg | " + " | ".join(f"{v / (nu - 2):7.0f}" for v in tot))
print("MFMA wave:  unit | wait B2 | (DMA of H1, stage + copy-out +) wait B1 | 6 k-steps (incl. MID / B3) | epilogue values / partials | unit total")
tot = np.zeros(5)
for w in range(nu):
    d = [m[w, 1] - m[w, 0], m[w, 2] - m[w, 1], m[w, 3] - m[w, 2], m[w, 4] - m[w, 3]]
    nxt = (m[w + 1, 0] - m[w, 0]) if w + 1 < nu else (m[w, 4] - m[w, 0])
    print(f"   {w:3d} | " + " | ".join(f"{int(v):7d}" for v in d) + f" | {int(nxt):7d}")
    if 1 <= w < nu - 1:
        tot += np.array(d + [nxt])
if nu > 2:
    print("   avg | " + " | ".join(f"{v / (nu - 2):7.0f}" for v in tot))
print(f"(s_memtime ticks; first stamp of the two roles: producer {p[0, 0]}, MFMA {m[0, 0]}; block span "
      f"{int(max(p[nu - 1, 6], m[nu - 1, 4]) - min(p[0, 0], m[0, 0]))} ticks)")

# per-wave arrival at B2 (round 6): which wave is the block waiting for?
wbuf = np.zeros((12, 40, 2), dtype=np.uint64)
if hasattr(lib, "p2m_tile_trace_dump_waves") and lib.p2m_tile_trace_dump_waves(wbuf.ctypes.data_as(ctypes.c_void_p)) == 0:
    wv = wbuf.astype(np.int64)
    print("arrival at B2(w), cycles after producer wave 4's unit head; waves 0-3 MFMA (k-steps of unit w - 1), 4-11 producers "
          "(gather of unit w):")
    for w in range(1, min(nu, 8)):
        base = wv[4, w, 0]
        print(f"   unit {w}: " + " ".join(f"{int(wv[k, w, 1] - base):6d}" for k in range(12)))
    print("the producers' own gather time (arrival at B2 - their own unit head), waves 4-11:")
    for w in range(1, min(nu, 8)):
        print(f"   unit {w}: " + " ".join(f"{int(wv[k, w, 1] - wv[k, w, 0]):6d}" for k in range(4, 12)))
