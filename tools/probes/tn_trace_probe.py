"""Phase stamps of one block of k_gemm_tn_ws<128, ROWS, 3> (probe build: tools/tn_trace.sh): the weight gradient of the
finest level's 128 -> 128 conv in row-set mode, 32 steady-state stages, every wave of the block.
    P2M_HIP_LIB=.../libp2m_hip_TNTRACE.so python tools/probes/tn_trace_probe.py [level Ka N]"""
import ctypes
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pose2mesh_release_amd import _lib, ops, synth  # noqa: E402

lvl, Ka, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (0, 128, 128)
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
_, gL, _, J = synth.make_graphs("human36")
g = ops.DeviceGraph(gL[lvl], "cuda:0")
X = torch.randn(B * g.V, Ka, device="cuda")
Gy = torch.randn(B * g.V, N, device="cuda")
E1, E2 = ops.cheb_basis_fwd_real(g, Gy, B, N, 0)
xa, ga = ops.amax_of(X), ops.amax_of(Gy)


def run():
    return ops.gemm_tn_rows(g, 1, B, X, Ka, 0, [Gy, E1, E2], N, True, a_amax=xa, g_amax=ga, g_bits=g.plane_bits)


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
nst = -(-g.n_real // 16)
print(f"B={B} ({B * 3 * (-(-Ka // 128))} blocks, {nst} stages: {ms * 1e6 / nst:.0f} ns per stage if one round) V={g.V} rows={g.n_real} Ka={Ka} N=3x{N}: {ms:.3f} ms, {2.0 * B * g.n_real * 3 * Ka * N / ms / 1e9:.1f} TF"
      f"   [{ops.GEMM_ARITH}]")
lib = _lib.hip()
if not hasattr(lib, "p2m_tn_trace_dump"):
    sys.exit("not a trace build (tools/tn_trace.sh build)")
buf = np.zeros((8, 32, 4), dtype=np.uint64)
assert lib.p2m_tn_trace_dump(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.astype(np.int64)
if not t.any():
    sys.exit("no stamps (block id out of range?)")
names = ["mfma0", "mfma1", "mfma2", "mfma3", "stgA0", "stgA1", "stgG0", "stgG1"]
print("cycles per stage, mean over 31 steady stages (s_memtime ticks):")
print("  wave    stage   operands-wait  work(issue)  tail-to-barrier  barrier-wait")
for w in range(8):
    tw = t[w]
    stage = np.diff(tw[:, 0])                      # barrier exit to barrier exit
    wait = (tw[:, 1] - tw[:, 0])[:-1]
    work = (tw[:, 2] - tw[:, 1])[:-1]
    tail = (tw[:, 3] - tw[:, 2])[:-1]
    bar = tw[1:, 0] - tw[:-1, 3]
    print(f"  {names[w]}  {stage.mean():7.0f}  {wait.mean():10.0f}    {work.mean():8.0f}    {tail.mean():8.0f}       {bar.mean():8.0f}"
          f"    (stage min/max {stage.min()}/{stage.max()})")
# who arrives last at each barrier?
arr = t[:, :, 3]
last = arr.argmax(axis=0)
print("last wave at the barrier, per stage:", " ".join(names[i] for i in last))
print("arrival spread (last - first) mean:", float((arr.max(axis=0) - arr.min(axis=0)).mean()))
print("mean arrival relative to the earliest wave, per wave:",
      " ".join(f"{names[w]}:{float((arr[w] - arr.min(axis=0)).mean()):.0f}" for w in range(8)))
