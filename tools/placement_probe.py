"""Step time against WHERE the model's buffers landed: several identical models in one process, their buffer addresses and step times.
python tools/placement_probe.py [n_models]"""
import sys, time
import pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench
from volume_segmantics_amd import _lib
from volume_segmantics_amd.engine import VolSegUnet
from volume_segmantics_amd.data.losses import HipDiceLoss
n_models = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
dev = torch.device("cuda", 0)
x, lab = bench.synth_batch(32, 256, 2, seed=1234)
x = x.to(dev)
t = torch.nn.functional.one_hot(lab, 2).permute(0, 3, 1, 2).to(dev, torch.uint8).contiguous()
crit = HipDiceLoss()
models = []
for i in range(n_models):
    m = VolSegUnet(2, device=dev, precision="bf16", seed=0)
    o = m.fused_adamw(lr=1e-4, fuse_step_into_backward=True)
    m.train()
    def step(m=m, o=o):
        o.zero_grad(); loss = crit(m(x), t); loss.backward(); o.step()
    for _ in range(5): step()
    torch.cuda.synchronize()
    models.append((m, o, step))
for i, (m, o, step) in enumerate(models):
    ts = []
    for r in range(3):
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    ws = [p["ws"] for p in m._plans.values()] if hasattr(m, "_plans") else []
    desc = " ".join(f"ws=0x{w.data_ptr():x}+{w.numel() / 2**20:.0f}MiB" for w in ws)
    if "--classes" in sys.argv:
        _lib.set_option("side_stream", 0)
        step(); torch.cuda.synchronize()
        _lib.check(_lib.lib.vs_profile_enable(1))
        for _ in range(2): step()
        torch.cuda.synchronize()
        raw = _lib.profile_read_raw()
        _lib.check(_lib.lib.vs_profile_enable(0))
        _lib.set_option("side_stream", 1)
        cls = {}
        for kind, tag, ms, fl, by, _v in raw: cls[kind] = cls.get(kind, 0.0) + ms / 2
        print("   classes (serialised, ms): " + " ".join(f"{k}={v:.3f}" for k, v in sorted(cls.items())), flush=True)
    print(f"model {i}: {np.median(ts):.3f} ms  flat=0x{m._flat.data_ptr():x} grad=0x{m._flat_grad.data_ptr():x} m=0x{o.exp_avg.data_ptr():x} v=0x{o.exp_avg_sq.data_ptr():x} {desc}", flush=True)
