"""Chasing the 8 ms/step mode (driver run of round 1; one run in six on an explicit stream): per-step host enqueue times and
allocator activity for runs of 20 eager steps on an explicit stream."""
import gc, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench
from volume_segmantics_amd.engine import VolSegUnet
from volume_segmantics_amd.data.losses import HipDiceLoss

dev = torch.device("cuda", 0)
x, lab = bench.synth_batch(32, 256, 2, seed=1234)
x = x.to(dev)
t = torch.nn.functional.one_hot(lab, 2).permute(0, 3, 1, 2).to(dev, torch.uint8).contiguous()
crit = HipDiceLoss()
m = VolSegUnet(2, device=dev, precision="bf16", seed=0)
o = m.fused_adamw(lr=1e-4, fuse_step_into_backward=True)
m.train()
def step():
    o.zero_grad(); loss = crit(m(x), t); loss.backward(); o.step()
mode = sys.argv[1] if len(sys.argv) > 1 else "explicit"
stream = torch.cuda.Stream() if mode == "explicit" else torch.cuda.current_stream()
with torch.cuda.stream(stream):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    for r in range(10):
        ms0 = torch.cuda.memory_stats()
        gc0 = gc.get_count()
        step(); torch.cuda.synchronize()
        ts = [time.perf_counter()]
        for _ in range(20):
            step(); ts.append(time.perf_counter())
        torch.cuda.synchronize()
        wall = (time.perf_counter() - ts[0]) / 20 * 1e3
        d = np.diff(ts) * 1e3
        ms1 = torch.cuda.memory_stats()
        print(f"run {r}: wall {wall:.3f} ms/step; host enqueue per step min {d.min():.2f} med {np.median(d):.2f} max {d.max():.2f}; "
              f"device allocs +{ms1['num_device_alloc'] - ms0['num_device_alloc']} frees +{ms1['num_device_free'] - ms0['num_device_free']} "
              f"alloc retries +{ms1['num_alloc_retries'] - ms0['num_alloc_retries']}; gc {gc0}->{gc.get_count()}", flush=True)
