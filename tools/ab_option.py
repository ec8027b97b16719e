"""A/B of a library option on the eager training step, interleaved rounds in ONE process (cdna guide rule 24):
    python tools/ab_option.py <option> <v0> <v1> [rounds] [steps]   - options read at plan creation need a fresh model per arm"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench
from volume_segmantics_amd import _lib
from volume_segmantics_amd.engine import VolSegUnet
from volume_segmantics_amd.data.losses import HipDiceLoss

opt_name, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda", 0)
x, lab = bench.synth_batch(32, 256, 2, seed=1234)
x = x.to(dev)
t = torch.nn.functional.one_hot(lab, 2).permute(0, 3, 1, 2).to(dev, torch.uint8).contiguous()
crit = HipDiceLoss()
arms = {}
for v in (v0, v1):
    _lib.set_option(opt_name, v)
    m = VolSegUnet(2, device=dev, precision="bf16", seed=0)
    o = m.fused_adamw(lr=1e-4, fuse_step_into_backward=True)
    m.train()
    def step(m=m, o=o):
        o.zero_grad(); loss = crit(m(x), t); loss.backward(); o.step()
    for _ in range(5):
        step()
    arms[v] = step
torch.cuda.synchronize()
res = {v: [] for v in arms}
for r in range(rounds):
    for v, step in arms.items():
        _lib.set_option(opt_name, v)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / steps * 1e3)
for v, ts in res.items():
    print(f"{opt_name}={v}: median {np.median(ts):.3f} ms/step, min {min(ts):.3f}, all {[round(q, 3) for q in ts]}")
