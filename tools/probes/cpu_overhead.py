"""How long does the host take to enqueue one train step (no sync)?  If < GPU step time the run is GPU-bound.
   python tools/probes/cpu_overhead.py [joint_set] [batch]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch, bench
JS = sys.argv[1] if len(sys.argv) > 1 else "coco"
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 256
step = bench.TrainStep(torch.device("cuda", 0), BATCH, JS, 1)
for _ in range(3): step()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("enqueue ms:", [round(a * 1e3, 1) for a, _ in ts], "total ms:", [round(b * 1e3, 1) for _, b in ts])
