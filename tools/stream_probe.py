"""Does the caller's stream matter?  Eager training steps timed on torch's default (null) stream and on an explicit stream;
run under different GPU_MAX_HW_QUEUES to see how HIP streams map onto hardware queues.
    python tools/stream_probe.py [rounds]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench
from volume_segmantics_amd import _lib
from volume_segmantics_amd.engine import VolSegUnet
from volume_segmantics_amd.data.losses import HipDiceLoss

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
x, lab = bench.synth_batch(32, 256, 2, seed=1234)
x = x.to(dev)
t = torch.nn.functional.one_hot(lab, 2).permute(0, 3, 1, 2).to(dev, torch.uint8).contiguous()
crit = HipDiceLoss()
m = VolSegUnet(2, device=dev, precision="bf16", seed=0)
o = m.fused_adamw(lr=1e-4, fuse_step_into_backward=True)
m.train()
AHEAD = int(os.environ.get("PROBE_AHEAD", "0"))   # > 0: the host stays at most this many steps ahead of the GPU
pend = []
def step():
    o.zero_grad(); loss = crit(m(x), t); loss.backward(); o.step()
    if AHEAD:
        e = torch.cuda.Event(); e.record(); pend.append(e)
        if len(pend) > AHEAD:
            pend.pop(0).synchronize()
explicit = torch.cuda.Stream()
detail = []
def run(steps=20):
    step(); torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for _ in range(steps):
        step(); ts.append(time.perf_counter())
    torch.cuda.synchronize()
    d = np.diff(ts) * 1e3
    detail.append(f"host enqueue/step min {d.min():.2f} med {np.median(d):.2f} max {d.max():.2f}; first 6: {[round(q, 2) for q in d[:6]]}")
    return (time.perf_counter() - ts[0]) / steps * 1e3
for _ in range(5):
    step()
PREHEAT = float(os.environ.get("PROBE_PREHEAT_S", "0"))
if PREHEAT:
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < PREHEAT:
        step(); torch.cuda.synchronize()
res = {"default": [], "explicit": []}
for r in range(rounds):
    res["default"].append(run())
    with torch.cuda.stream(explicit):
        res["explicit"].append(run())
for i, dline in enumerate(detail):
    print(f"  run {i} ({'default' if i % 2 == 0 else 'explicit'}): {dline}")
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), "PROBE_AHEAD =", AHEAD, "PREHEAT_S =", PREHEAT)
for k, v in res.items():
    print(f"  caller stream {k}: median {np.median(v):.3f} ms/step  {[round(q, 3) for q in v]}")
