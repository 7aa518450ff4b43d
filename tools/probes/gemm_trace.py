"""Where do the cycles of one k_gemm_planes_ws iteration go?  Needs the probe build of the library:
  P2M_HIPCC_FLAGS=-DP2M_GEMM_TRACE=1 python -m pose2mesh_release_amd.build   (into a scratch copy, see gemm_trace.sh)
s_memtime stamps of producer wave 4 / consumer wave 0 of 8 mid-launch blocks:
  producer: 0 top | 1 A loads arrived | 5 A slices stored | 2 B slices stored | 3 next loads issued | 4 barrier passed
  consumer: 0 top | 1 fragments read | 2 MFMAs issued | 3 barrier passed"""
import ctypes
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pose2mesh_release_amd import _lib, ops  # noqa: E402

V, Ka, N = [int(x) for x in os.environ.get("PROBE_SHAPE", "5888,128,128").split(",")]
B = 256
M = B * V
A = [torch.randn(M, Ka, device="cuda") for _ in range(3)]
W = torch.randn(3 * Ka, N, device="cuda") / (3 * Ka) ** 0.5
bias = torch.randn(N, device="cuda")
for _ in range(2):
    ops.gemm_planes(A, Ka, 0, W, bias, M, N, 1, False)
torch.cuda.synchronize()
lib = _lib.hip()
IT, ST = 40, 6
n = 8 * 2 * IT * ST
buf = (ctypes.c_ulonglong * n)()
lib.p2m_debug_gemm_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.p2m_debug_gemm_trace(buf, n)
assert rc == 0
t = np.array(buf, dtype=np.uint64).reshape(8, 2, IT, ST).astype(np.int64)
for blk in range(8):
    p, c = t[blk, 0], t[blk, 1]
    ok = (p[4:36, 0] > 0).all() and (c[4:36, 0] > 0).all()
    if not ok:
        print("block", blk, "incomplete trace")
        continue
    it = np.diff(p[4:36, 0])
    print(f"block {blk}: iteration {it.mean():7.0f} cycles (min {it.min()}, max {it.max()})")
    d = p[4:35]
    print(f"   producer: wait A {np.mean(d[:, 1] - d[:, 0]):6.0f} | split A + issue A stores {np.mean(d[:, 5] - d[:, 1]):6.0f} | "
          f"wait B + issue B stores {np.mean(d[:, 2] - d[:, 5]):6.0f} | "
          f"(drain LDS +) issue loads {np.mean(d[:, 3] - d[:, 2]):6.0f} | barrier wait {np.mean(d[:, 4] - d[:, 3]):6.0f}")
    d = c[4:35]
    print(f"   consumer: frag reads {np.mean(d[:, 1] - d[:, 0]):6.0f} | 24 MFMAs issue {np.mean(d[:, 2] - d[:, 1]):6.0f} | "
          f"barrier wait {np.mean(d[:, 3] - d[:, 2]):6.0f} | next top {np.mean(c[5:36, 0] - d[:, 3]):6.0f}")
