"""Diagnostic (not a test): per-parameter gradient error of the HIP engine vs the CPU oracle."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))  # repo root
import torch
from oracle import predictor_numpy as P
from oracle.unet_resnet34_torch import seeded_oracle
from volume_segmantics_amd.engine import VolSegUnet

DEV = "cuda:0"
for precision in ("fp32", "bf16"):
    for B, hw in ((4, 64), (2, 64), (4, 128)):
        oracle = seeded_oracle(2, 3, perturb_bn=False)
        model = VolSegUnet(2, device=DEV, precision=precision, init="none")
        model.load_state_dict(oracle.state_dict())
        g = torch.Generator().manual_seed(5)
        x = torch.randn(B, 1, hw, hw, generator=g)
        mask = (torch.rand(B, hw, hw, generator=g) > 0.65).to(torch.uint8)
        _, t = P.prepare_training_batch(x, mask, 2)
        oracle.eval(); model.eval()
        with torch.no_grad():
            e_eval = (model(x.to(DEV)).cpu() - oracle(x)).abs().max().item()
        oracle.train(); model.train()
        ro = oracle(x); rl = P.dice_loss_none(ro, t.float()); rl.backward()
        out = model(x.to(DEV)); loss = P.dice_loss_none(out, t.to(DEV).float()); loss.backward()
        torch.cuda.synchronize()
        ref = dict(oracle.named_parameters())
        errs = sorted((((p.grad.cpu() - ref[n].grad).norm() / (ref[n].grad.norm() + 1e-30)).item(), n) for n, p in model.named_parameters())
        print(f"{precision} B={B} {hw}x{hw}: eval logits max err {e_eval:.2e}; train logits max err "
              f"{(out.detach().cpu() - ro.detach()).abs().max().item():.2e} rel {((out.detach().cpu()-ro.detach()).norm()/ro.detach().norm()).item():.2e}; loss diff {abs(loss.item() - rl.item()):.2e}")
        print("   grad rel err: min %.2e median %.2e max %.2e (%s)" % (errs[0][0], errs[len(errs) // 2][0], errs[-1][0], errs[-1][1]))
        print("   worst 5:", [(f"{e:.1e}", n) for e, n in errs[-5:]])
