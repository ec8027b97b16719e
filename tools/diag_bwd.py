"""Diagnostic (not a test): compare dz (grad wrt every conv output) of the HIP backward with oracle hooks."""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))  # repo root
import torch
from oracle import predictor_numpy as P
from oracle.unet_resnet34_torch import seeded_oracle
from volume_segmantics_amd import _lib as L
from volume_segmantics_amd.engine import VolSegUnet

DEV = "cuda:0"
precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B, hw = 4, 64
oracle = seeded_oracle(2, 3, perturb_bn=False)
model = VolSegUnet(2, device=DEV, precision=precision, init="none")
model.load_state_dict(oracle.state_dict())
g = torch.Generator().manual_seed(5)
x = torch.randn(B, 1, hw, hw, generator=g)
mask = (torch.rand(B, hw, hw, generator=g) > 0.65).to(torch.uint8)
_, t = P.prepare_training_batch(x, mask, 2)
grads, acts = {}, {}
for name, mod in oracle.named_modules():
    if isinstance(mod, torch.nn.Conv2d):
        mod.register_full_backward_hook(lambda m, gi, go, name=name: grads.__setitem__(name, go[0].detach()))
        mod.register_forward_hook(lambda m, i, o, name=name: acts.__setitem__(name, o.detach()))
oracle.train(); model.train()
P.dice_loss_none(oracle(x), t.float()).backward()
out = model(x.to(DEV)); P.dice_loss_none(out, t.to(DEV).float()).backward()
torch.cuda.synchronize()
plan = model._plans[(hw, hw)]
ws = plan["ws"]
dt = torch.float32 if precision == "fp32" else torch.bfloat16
esz = 4 if precision == "fp32" else 2
nu = L.lib.vs_unet_num_units(plan["handle"])
name = C.create_string_buffer(128); c, h, w = C.c_int(), C.c_int(), C.c_int()
oa, oz, oda, odz = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
for u in reversed(range(nu)):
    L.check(L.lib.vs_unet_debug_unit(plan["handle"], u, name, 128, C.byref(c), C.byref(h), C.byref(w), C.byref(oa), C.byref(oz), C.byref(oda), C.byref(odz)))
    wn = name.value.decode()
    if wn == "maxpool" or wn.startswith("segmentation_head"):
        continue
    mod = wn.rsplit(".weight", 1)[0]
    n_el = B * c.value * h.value * w.value
    def get(off):
        return ws[off:off + n_el * esz].view(dt).view(B, h.value, w.value, c.value).float().cpu().permute(0, 3, 1, 2)
    dz, z = get(odz.value), get(oz.value)
    rdz, rz = grads[mod], acts[mod]
    nz_ref, nz = rdz != 0, dz != 0
    mism = (nz_ref != nz).sum().item()
    agree = nz_ref == nz
    e_agree = ((dz - rdz)[agree].norm() / (rdz[agree].norm() + 1e-30)).item()
    print(f"   zero-pattern mismatches {mism} of {rdz.numel()}  dz err on agreeing positions {e_agree:.1e}")
    print(f"{u:3d} {mod:40s} z err {((z - rz).norm() / rz.norm()).item():.1e}  dz err {((dz - rdz).norm() / (rdz.norm() + 1e-30)).item():.1e}  |dz| {rdz.norm().item():.2e}")
