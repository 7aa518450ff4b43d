"""How long does the host take to enqueue one train step (no sync)?  If < GPU step time the run is GPU-bound."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R]
import torch, bench
step = bench.TrainStep(torch.device("cuda", 0), 256, "coco", 1)
for _ in range(3): step()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("enqueue ms:", [round(a * 1e3, 1) for a, _ in ts], "total ms:", [round(b * 1e3, 1) for _, b in ts])
