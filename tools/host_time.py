"""Host (enqueue) time vs GPU time of one training step (diagnostics; needs a GPU)."""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from volume_segmantics_amd.engine import VolSegUnet
from volume_segmantics_amd.data.losses import HipDiceLoss
dev = "cuda:0"
model = VolSegUnet(2, device=dev, precision="bf16", seed=0)
x = torch.randn(32, 1, 256, 256, device=dev)
t = torch.nn.functional.one_hot((torch.rand(32, 256, 256, device=dev) > 0.5).long(), 2).permute(0, 3, 1, 2).float()
opt = model.fused_adamw(lr=1e-4, fuse_step_into_backward=True)
sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=200, pct_start=0.3)
crit = HipDiceLoss()
model.train()
def step(timers):
    t0 = time.perf_counter(); opt.zero_grad(); logits = model(x); t1 = time.perf_counter()
    loss = crit(logits, t); t2 = time.perf_counter()
    loss.backward(); t3 = time.perf_counter()
    opt.step(); sched.step(); t4 = time.perf_counter()
    for k, v in zip(("fwd", "loss", "bwd", "opt+sched"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)): timers[k] = timers.get(k, 0.0) + v
for _ in range(10): step({})
torch.cuda.synchronize()
timers = {}
n = 30
w0 = time.perf_counter()
for _ in range(n): step(timers)
h1 = time.perf_counter()
torch.cuda.synchronize()
w1 = time.perf_counter()
print(f"host enqueue total {(h1 - w0) / n * 1e3:.3f} ms/step, wall {(w1 - w0) / n * 1e3:.3f} ms/step")
print({k: round(v / n * 1e3, 3) for k, v in timers.items()})
