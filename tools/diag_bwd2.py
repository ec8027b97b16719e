"""Diagnostic: self-consistency of dgrad + bn_bwd for the last decoder block, recomputed in fp64 on the CPU
from the engine's own saved tensors."""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))  # repo root
import torch
import torch.nn.functional as F
from oracle import predictor_numpy as P
from oracle.unet_resnet34_torch import seeded_oracle
from volume_segmantics_amd import _lib as L
from volume_segmantics_amd.engine import VolSegUnet

DEV = "cuda:0"
B, hw = 4, 64
oracle = seeded_oracle(2, 3, perturb_bn=False)
model = VolSegUnet(2, device=DEV, precision="fp32", init="none")
model.load_state_dict(oracle.state_dict())
g = torch.Generator().manual_seed(5)
x = torch.randn(B, 1, hw, hw, generator=g)
mask = (torch.rand(B, hw, hw, generator=g) > 0.65).to(torch.uint8)
_, t = P.prepare_training_batch(x, mask, 2)
model.train()
out = model(x.to(DEV)); P.dice_loss_none(out, t.to(DEV).float()).backward()
torch.cuda.synchronize()
plan = model._plans[(hw, hw)]; ws = plan["ws"]
name = C.create_string_buffer(128); c, h, w = C.c_int(), C.c_int(), C.c_int()
oa, oz, oda, odz = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
def unit(u):
    L.check(L.lib.vs_unet_debug_unit(plan["handle"], u, name, 128, C.byref(c), C.byref(h), C.byref(w), C.byref(oa), C.byref(oz), C.byref(oda), C.byref(odz)))
    n_el = B * c.value * h.value * w.value
    def get(off):
        return ws[off:off + n_el * 4].view(torch.float32).view(B, h.value, w.value, c.value).cpu().double().permute(0, 3, 1, 2).contiguous()
    return name.value.decode(), get(oa.value), get(oz.value), get(oda.value), get(odz.value)
sd = {k: v.double().cpu() for k, v in model.state_dict().items()}
n46, a46, z46, da46, dz46 = unit(46)   # dec4.conv2
n45, a45, z45, da45, dz45 = unit(45)   # dec4.conv1
print(n46, n45)
def bn_bwd_ref(da, a, z, gamma):
    dzm = da * (a > 0)
    mean = z.mean((0, 2, 3), keepdim=True); var = z.var((0, 2, 3), unbiased=False, keepdim=True)
    invstd = 1 / torch.sqrt(var + 1e-5); xh = (z - mean) * invstd
    M = z.numel() / z.shape[1]
    db = dzm.sum((0, 2, 3), keepdim=True); dg = (dzm * xh).sum((0, 2, 3), keepdim=True)
    return gamma.view(1, -1, 1, 1) * invstd * (dzm - db / M - xh * dg / M), dzm
ref46, dzm46 = bn_bwd_ref(da46, a46, z46, sd["decoder.blocks.4.conv2.1.weight"])
print("bn_bwd(dec4.conv2) self-consistency:", ((dz46 - ref46).norm() / ref46.norm()).item(), " kept fraction", (ref46.norm() / (dzm46.norm() * (sd["decoder.blocks.4.conv2.1.weight"].abs().mean() / torch.sqrt(z46.var() + 1e-5)))).item())
# dgrad of conv2: da45 should equal conv_transpose(dz46, W)
W = sd["decoder.blocks.4.conv2.0.weight"]
ref_da45 = F.conv_transpose2d(dz46, W, padding=1)
print("dgrad(dec4.conv2) self-consistency:", ((da45 - ref_da45).norm() / ref_da45.norm()).item())
ref45, dzm45 = bn_bwd_ref(da45, a45, z45, sd["decoder.blocks.4.conv1.1.weight"])
print("bn_bwd(dec4.conv1) self-consistency:", ((dz45 - ref45).norm() / ref45.norm()).item(), " |P da|/|da| ~", (ref45.norm() / dzm45.norm()).item())
# sensitivity: perturb da45 by 1e-5 relative noise and see the relative change of the BN backward output
noise = torch.randn_like(da45) * da45.norm() / da45.numel() ** 0.5 * 1e-5
pert, _ = bn_bwd_ref(da45 + noise, a45, z45, sd["decoder.blocks.4.conv1.1.weight"])
print("bn_bwd(dec4.conv1) response to 1e-5 random input noise:", ((pert - ref45).norm() / ref45.norm()).item())
pz, _ = bn_bwd_ref(da45, a45, z45 * (1 + 1e-5 * torch.randn_like(z45)), sd["decoder.blocks.4.conv1.1.weight"])
print("bn_bwd(dec4.conv1) response to 1e-5 relative noise on z:", ((pz - ref45).norm() / ref45.norm()).item())
