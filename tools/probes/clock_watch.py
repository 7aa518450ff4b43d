"""Runs one kernel family in a loop for a few seconds while a side thread samples rocm-smi (sclk, power): does the GPU hold its
clock under this kernel?   python tools/probes/clock_watch.py tn|planes|tile|bn|idle [seconds]"""
import os
import subprocess
import sys
import threading
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch  # noqa: E402

from pose2mesh_release_amd import ops, synth  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "tn"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
B, Ka, N = 256, 128, 128
_, gL, _, J = synth.make_graphs("human36")
g = ops.DeviceGraph(gL[0], "cuda:0")
X = torch.randn(B * g.V, Ka, device="cuda")
Gy = torch.randn(B * g.V, N, device="cuda")
E1, E2 = ops.cheb_basis_fwd_real(g, Gy, B, N, 0)
T1, T2 = ops.cheb_basis_fwd_real(g, X, B, Ka, 0)
W = torch.randn(3 * Ka, N, device="cuda") / (3 * Ka) ** 0.5
Bx = ops.weight_split(W)
C = torch.empty(B * g.V, N, device="cuda")
xa, ga = ops.amax_of(X), ops.amax_of(Gy)
fns = {
    "tn": lambda: ops.gemm_tn_rows(g, 1, B, X, Ka, 0, [Gy, E1, E2], N, True, a_amax=xa, g_amax=ga, g_bits=g.plane_bits),
    "planes": lambda: ops.gemm_planes_rows(g, 1, B, [X, T1, T2], Ka, 0, True, W, None, None, C, N, False, Bx=Bx, amax=xa,
                                           amax_bits=g.plane_bits),
    "tile": lambda: ops.cheb_tile_gemm(g, 0, X, X, Ka, Bx, None, None, C, N, B, stats=True, amax=xa),
    "copy": lambda: C.copy_(Gy),
    "idle": lambda: time.sleep(0.01),
}
fn = fns[what]
samples = []
stop = False


def watch():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = [ln.split(":")[-1].strip() for ln in out.splitlines() if "sclk" in ln.lower()]
            pw = [ln.split(":")[-1].strip() for ln in out.splitlines() if "power" in ln.lower() and "W" in ln]
            samples.append((sclk[:1], pw[:1]))
        except Exception as e:  # noqa: BLE001
            samples.append((str(e)[:60], None))
        time.sleep(0.3)


th = threading.Thread(target=watch)
th.start()
fn()
torch.cuda.synchronize()
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    n += 20
el = time.time() - t0
stop = True
th.join()
print(f"{what}: {n} calls in {el:.2f} s = {el / n * 1e3:.3f} ms per call")
for s in samples:
    print("   ", s)
