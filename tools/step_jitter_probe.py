"""Per-step GPU time (HIP events on the caller's stream) over a long run of eager training steps: is the 8 ms/step mode a
uniform slowdown or a few long stalls?   python tools/step_jitter_probe.py [steps] [graph]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench
from volume_segmantics_amd.engine import VolSegUnet
from volume_segmantics_amd.data.losses import HipDiceLoss

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
use_graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
dev = torch.device("cuda", 0)
x, lab = bench.synth_batch(32, 256, 2, seed=1234)
x = x.to(dev)
t = torch.nn.functional.one_hot(lab, 2).permute(0, 3, 1, 2).to(dev, torch.uint8).contiguous()
crit = HipDiceLoss()
m = VolSegUnet(2, device=dev, precision="bf16", seed=0)
o = m.fused_adamw(lr=1e-4, fuse_step_into_backward=True)
m.train()
def step():
    if use_graph:
        m.fused_train_step(x, t, o, clone_loss=False)
    else:
        o.zero_grad(); loss = crit(m(x), t); loss.backward(); o.step()
t_start = time.perf_counter()
for _ in range(3):
    step()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
ev[0].record()
for i in range(n):
    step(); ev[i + 1].record(); host.append(time.perf_counter())
    if i % 8 == 7:
        ev[i + 1].synchronize()      # keep the host at most 8 steps ahead (like a loop that reads the loss now and then)
torch.cuda.synchronize()
d = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(n)])
print(f"{'graph' if use_graph else 'eager'}: {n} steps: median {np.median(d):.3f} ms, mean {d.mean():.3f}, p99 {np.percentile(d, 99):.3f}, max {d.max():.3f}")
slow = np.where(d > 1.25 * np.median(d))[0]
print(f"steps slower than 1.25 x median: {len(slow)}")
for i in slow[:40]:
    print(f"   step {i}: {d[i]:.3f} ms at t = {host[i] - t_start:.3f} s")
