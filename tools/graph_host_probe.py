"""Host cost of a training step: call by call vs hipGraph replay, with and without the side stream (a linear graph).
    python tools/graph_host_probe.py [steps]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from volume_segmantics_amd import _lib
from volume_segmantics_amd.engine import VolSegUnet
from volume_segmantics_amd.data.losses import HipDiceLoss

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
x, lab = bench.synth_batch(32, 256, 2, seed=1234)
x = x.to(dev)
t = torch.nn.functional.one_hot(lab, 2).permute(0, 3, 1, 2).to(dev, torch.uint8).contiguous()
crit = HipDiceLoss()
for side, gmode in ((1, "seg"), (1, "branch"), (0, "branch")):
    _lib.set_option("side_stream", side)
    os.environ["VOLSEG_STEP_GRAPH"] = gmode
    model = VolSegUnet(2, device=dev, precision="bf16", seed=0)
    opt = model.fused_adamw(lr=1e-4, fuse_step_into_backward=True)
    model.train()
    def eager():
        opt.zero_grad(); loss = crit(model(x), t); loss.backward(); opt.step()
    def graph():
        model.fused_train_step(x, t, opt, clone_loss=False)
    for name, fn in (("eager", eager), ("graph", graph)):
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"side_stream={side} mode={gmode} {name}: host enqueue {(t1 - t0) / steps * 1e3:.3f} ms/step, wall {(t2 - t0) / steps * 1e3:.3f} ms/step", flush=True)
    # host cost of one replay when the GPU is idle (no back-pressure from a full queue)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); graph(); ts.append(time.perf_counter() - t0); torch.cuda.synchronize()
    print(f"side_stream={side} mode={gmode} graph, idle GPU: host {min(ts) * 1e3:.3f} ms per replay", flush=True)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); eager(); ts.append(time.perf_counter() - t0); torch.cuda.synchronize()
    print(f"side_stream={side} eager, idle GPU: host {min(ts) * 1e3:.3f} ms per step", flush=True)
    del model, opt
