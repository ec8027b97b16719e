"""A/B of several option settings on the eager training step, interleaved in one process: python tools/ab_multi.py "a=1,b=2" "a=0" "env:VS_X=0" ..."""
import os, sys, time
import pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench
from volume_segmantics_amd import _lib
from volume_segmantics_amd.engine import VolSegUnet
from volume_segmantics_amd.data.losses import HipDiceLoss
arms_spec = sys.argv[1:]
dev = torch.device("cuda", 0)
x, lab = bench.synth_batch(32, 256, 2, seed=1234)
x = x.to(dev)
t = torch.nn.functional.one_hot(lab, 2).permute(0, 3, 1, 2).to(dev, torch.uint8).contiguous()
crit = HipDiceLoss()
def setopts(spec):
    for kv in spec.split(","):
        k, v = kv.split("=")
        if k.startswith("env:"): os.environ[k[4:]] = v      # (switches the library reads from the environment at every launch)
        else: _lib.set_option(k, int(v))
arms = {}
for spec in arms_spec:
    setopts(spec)
    m = VolSegUnet(2, device=dev, precision="bf16", seed=0)
    o = m.fused_adamw(lr=1e-4, fuse_step_into_backward=True)
    m.train()
    def step(m=m, o=o):
        o.zero_grad(); loss = crit(m(x), t); loss.backward(); o.step()
    for _ in range(5): step()
    arms[spec] = step
torch.cuda.synchronize()
res = {s: [] for s in arms}
for r in range(5):
    for spec, step in arms.items():
        setopts(spec)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize()
        res[spec].append((time.perf_counter() - t0) / 20 * 1e3)
for spec, ts in res.items():
    print(f"{spec}: median {np.median(ts):.3f} ms/step, min {min(ts):.3f}")
