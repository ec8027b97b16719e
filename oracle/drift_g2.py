"""How far may a CORRECT implementation drift from the g2 golden (three reference training steps, fp32)?

Test infrastructure (run in the build container; writes tests/golden/g2_drift.npz).  The golden's losses after the first
AdamW step depend on gradients that flip with the sign of pre-activations lying ~1e-6 from zero (ReLU masks), and AdamW
normalises every gradient element to +-lr: two correct implementations therefore diverge by far more than rounding after one
step.  This script MEASURES that divergence instead of guessing it: the same three steps (oracle network, reference loop:
vol_seg_2d_trainer.py:419-432, AdamW + OneCycleLR) are run

  * in float64 (the "true" trajectory),
  * in float32 with every input pixel moved by one ulp up / down (a perturbation smaller than any implementation difference),
  * in float32 with a different intra-op thread count (another summation order, as another correct implementation has),

and the spread of the per-step losses and of the updated parameters against the plain float32 run (= the golden) is stored.
and - because another correct implementation differs from the oracle by more than one input ulp - along a LADDER of
implementation-difference sizes: every weight multiplied by 1 + eps * U(-1, 1) for eps = 1e-7 .. 3e-5, recording how far the
step-0 logits move (delta) and how far the later losses / updated parameters then drift.  tests/test_hip_unet.py measures the
engine's own delta against the oracle and takes its tolerances from the ladder rung with at least that delta."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle import predictor_numpy as P  # noqa: E402
from oracle.unet_resnet34_torch import seeded_oracle  # noqa: E402

KEEP = ["segmentation_head.0.weight", "decoder.blocks.4.conv2.0.weight", "encoder.conv1.weight", "encoder.layer4.2.bn2.weight",
        "encoder.layer2.0.downsample.0.weight"]


def run(x, m, dtype=torch.float32, threads=8, weight_eps=0.0):
    torch.set_num_threads(threads)
    net = seeded_oracle(2, 3, perturb_bn=False).to(dtype)
    if weight_eps:   # every weight moved by a relative weight_eps * U(-1, 1): the size of accumulated rounding differences
        gen = torch.Generator().manual_seed(77)
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1 + weight_eps * (2 * torch.rand(p.shape, generator=gen, dtype=p.dtype) - 1))
    net.train()
    with torch.no_grad():
        logits0 = net(x.to(dtype)).double().numpy()      # (train-mode forward; running statistics move - they feed nothing below)
    net = seeded_oracle(2, 3, perturb_bn=False).to(dtype) if not weight_eps else net
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=2e-3, steps_per_epoch=4, epochs=1, pct_start=0.3)
    net.train()
    losses = []
    _, t = P.prepare_training_batch(x, m, 2)
    for _ in range(3):
        opt.zero_grad()
        loss = P.dice_loss_none(net(x.to(dtype)), t.to(dtype))
        loss.backward()
        opt.step()
        sched.step()
        losses.append(float(loss))
    sd = net.state_dict()
    return np.array(losses), {k: sd[k].double().numpy().copy() for k in KEEP}, logits0


def main():
    g = np.load(REPO / "tests" / "golden" / "g2_train3_b4_64.npz")
    x, m = torch.tensor(g["x"]), torch.tensor(g["mask"])
    base_l, base_p, base_logits = run(x, m)
    assert np.allclose(base_l, g["losses"], atol=2e-6), (base_l, g["losses"])
    variants = {
        "fp64": run(x, m, torch.float64),
        "ulp_up": run(torch.nextafter(x, torch.full_like(x, float("inf"))), m),
        "ulp_down": run(torch.nextafter(x, torch.full_like(x, float("-inf"))), m),
        "threads2": run(x, m, threads=2),
        "threads1": run(x, m, threads=1),
    }
    # a ladder of implementation-difference sizes: what a forward pass that agrees with the oracle's logits to delta may do later
    ladder = {eps: run(x, m, weight_eps=eps) for eps in (1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5)}
    out = {"base_losses": base_l}
    loss_dev = np.zeros(3)
    rel_dev, cos_min = {k: 0.0 for k in KEEP}, {k: 1.0 for k in KEEP}
    w0 = {k: v.double().numpy() for k, v in seeded_oracle(2, 3, False).state_dict().items() if k in KEEP}
    lad = []
    for eps, (l, p, lg) in ladder.items():
        delta = float(np.abs(lg - base_logits).max())
        rel = max(np.linalg.norm(p[k] - base_p[k]) / (np.linalg.norm(base_p[k]) + 1e-12) for k in KEEP)
        w0_ = {k: v.double().numpy() for k, v in seeded_oracle(2, 3, False).state_dict().items() if k in KEEP}
        # the perturbed run starts from perturbed weights: compare UPDATES (end - own start), start = w0 * (1 + eps U) ~ w0
        cos = min(float(((p[k] - w0_[k]).ravel() @ (base_p[k] - w0_[k]).ravel()) /
                        (np.linalg.norm(p[k] - w0_[k]) * np.linalg.norm(base_p[k] - w0_[k]) + 1e-30)) for k in KEEP)
        lad.append((eps, delta, *np.abs(l - base_l), rel, cos))
        print(f"weights * (1 + {eps:g} U): logits move by {delta:.3e}; |dloss| {np.abs(l - base_l)}; worst param rel dev {rel:.3e}, worst update cosine {cos:.4f}")
    out["ladder"] = np.array(lad)     # columns: eps, max |dlogit| at step 0, |dloss| steps 0..2, worst relative parameter deviation, worst update cosine
    for name, (l, p, _) in variants.items():
        out[f"losses__{name}"] = l
        loss_dev = np.maximum(loss_dev, np.abs(l - base_l))
        for k in KEEP:
            rel = np.linalg.norm(p[k] - base_p[k]) / (np.linalg.norm(base_p[k]) + 1e-12)
            du, dv = (p[k] - w0[k]).ravel(), (base_p[k] - w0[k]).ravel()
            cos = float(du @ dv / (np.linalg.norm(du) * np.linalg.norm(dv) + 1e-30))
            rel_dev[k], cos_min[k] = max(rel_dev[k], rel), min(cos_min[k], cos)
        print(f"{name:9s} losses {l}  |dloss| {np.abs(l - base_l)}")
    out["loss_dev_max"] = loss_dev
    out["param_names"] = np.array(KEEP)
    out["param_rel_dev_max"] = np.array([rel_dev[k] for k in KEEP])
    out["param_update_cos_min"] = np.array([cos_min[k] for k in KEEP])
    print("max |dloss| per step over the variants:", loss_dev)
    for k in KEEP:
        print(f"  {k:45s} rel dev {rel_dev[k]:.3e}  min cos of the update {cos_min[k]:.4f}")
    np.savez_compressed(REPO / "tests" / "golden" / "g2_drift.npz", **out)


if __name__ == "__main__":
    main()
