"""Does the whole train step capture into one hipGraph, and what does a replay cost?
   python tools/probes/train_graph_probe.py [joint_set] [batch]"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch  # noqa: E402

import bench  # noqa: E402

JS = sys.argv[1] if len(sys.argv) > 1 else "coco"
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
step = bench.TrainStep(dev, BATCH, JS, 1)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
eager_ms = None
with torch.cuda.stream(s):
    eager_ms = timed(step, 8)
print(f"{JS} B={BATCH}: eager {eager_ms:.3f} ms/step", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    step.opt.zero_grad()
try:
    with torch.cuda.graph(g, stream=s):
        loss = step()
    torch.cuda.synchronize()
    l0 = float(loss)
    ms = timed(g.replay, 8)
    print(f"captured: replay {ms:.3f} ms/step, loss after replays {float(loss):.6f} (at capture {l0:.6f})", flush=True)
except Exception as e:  # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:400], flush=True)
