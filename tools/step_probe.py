"""Decomposition of the training step on ONE GPU (diagnostics): forward only, whole step, whole step with the weight gradients
serialised on the caller's stream, whole step with the encoder frozen (no encoder weight gradients) - interleaved rounds."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench
from volume_segmantics_amd import _lib
from volume_segmantics_amd.engine import VolSegUnet
from volume_segmantics_amd.data.losses import HipDiceLoss

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
x, lab = bench.synth_batch(32, 256, 2, seed=1234)
x = x.to(dev)
t = torch.nn.functional.one_hot(lab, 2).permute(0, 3, 1, 2).to(dev, torch.uint8).contiguous()
crit = HipDiceLoss()


def make(frozen=False):
    m = VolSegUnet(2, device=dev, precision="bf16", seed=0)
    if frozen:
        for n, p in m.named_parameters():
            if "encoder" in n and "conv" in n:
                p.requires_grad = False
    o = m.fused_adamw(lr=1e-4, fuse_step_into_backward=True)
    m.train()
    return m, o


m, o = make()
mf, of = make(True)


def full(m=m, o=o):
    o.zero_grad(); loss = crit(m(x), t); loss.backward(); o.step()


def fwd_only():
    with torch.no_grad():
        m(x)


def frozen():
    full(mf, of)


def serial():
    _lib.set_option("side_stream", 0)
    full()
    _lib.set_option("side_stream", 1)


arms = {"forward only (train-mode BN)": fwd_only, "whole step": full, "whole step, wgrad on the caller's stream": serial, "whole step, encoder frozen": frozen}
for fn in arms.values():
    for _ in range(5):
        fn()
torch.cuda.synchronize()
res = {k: [] for k in arms}
for r in range(rounds):
    for k, fn in arms.items():
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / steps * 1e3)
for k, ts in res.items():
    print(f"{k:45s} median {np.median(ts):.3f} ms, min {min(ts):.3f}")
