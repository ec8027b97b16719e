"""-m gpu: the whole network through the engine (C ABI underneath) against the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import fingerprint
from hip_helpers import DEV, sync
from oracle import predictor_numpy as P
from oracle.unet_resnet34_torch import seeded_oracle

pytestmark = pytest.mark.gpu


def _pair(classes, seed, precision, perturb_bn=True):
    from volume_segmantics_amd.engine import VolSegUnet
    oracle = seeded_oracle(classes, seed, perturb_bn)
    model = VolSegUnet(classes, device=DEV, precision=precision, init="none")
    model.load_state_dict(oracle.state_dict())
    return oracle, model


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 96, 64), (3, 32, 32)])
def test_eval_logits_fp32_within_1e3(shape):
    oracle, model = _pair(3, 0, "fp32")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(shape[0], 1, *shape[1:], generator=g)
    oracle.eval(); model.eval()
    with torch.no_grad():
        ref = oracle(x)
        got = model(x.to(DEV)).cpu()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 1e-3, (got - ref).abs().max()   # north_star: logits within 1e-3 fp32


def test_eval_logits_bf16_close_and_labels_agree():
    oracle, model = _pair(4, 0, "bf16")
    x = torch.randn(2, 1, 64, 64, generator=torch.Generator().manual_seed(2))
    oracle.eval(); model.eval()
    with torch.no_grad():
        ref = oracle(x)
        got = model(x.to(DEV)).cpu()
    rel = (got - ref).norm() / ref.norm()
    assert rel < 0.05, rel
    agree = (got.argmax(1) == ref.argmax(1)).float().mean().item()
    assert agree > 0.9, agree


@pytest.mark.parametrize("precision,rtol", [("fp32", 2e-3), ("bf16", 0.12)])
def test_train_forward_backward_matches_autograd(precision, rtol):
    oracle, model = _pair(2, 3, precision, perturb_bn=False)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 1, 64, 64, generator=g)
    mask = (torch.rand(4, 64, 64, generator=g) > 0.65).to(torch.uint8)
    _, t = P.prepare_training_batch(x, mask, 2)
    oracle.train(); model.train()
    ref_out = oracle(x)
    ref_loss = P.dice_loss_none(ref_out, t.float())
    ref_loss.backward()
    out = model(x.to(DEV))
    loss = P.dice_loss_none(out, t.to(DEV).float())
    loss.backward()
    sync()
    assert abs(loss.item() - ref_loss.item()) < (1e-4 if precision == "fp32" else 2e-2)
    assert (out.detach().cpu() - ref_out.detach()).abs().max() < (2e-3 if precision == "fp32" else 0.35)
    ref_grads = dict(oracle.named_parameters())
    worst = 0.0
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        r = ref_grads[name].grad
        err = ((p.grad.cpu() - r).norm() / (r.norm() + 1e-12)).item()
        worst = max(worst, err)
        assert err < rtol, (name, err)
    # running statistics moved exactly like torch's
    osd, msd = oracle.state_dict(), model.state_dict()
    for k in osd:
        if "running" in k:
            assert torch.allclose(msd[k].cpu(), osd[k], rtol=1e-3 if precision == "fp32" else 5e-2, atol=1e-4 if precision == "fp32" else 2e-2), k
        if "num_batches" in k:
            assert int(msd[k]) == int(osd[k]) == 1


def test_three_reference_training_steps_fp32(golden):
    """Golden from the reference's own _train_one_batch + AdamW + OneCycleLR (oracle/gen_goldens.py G2)."""
    g = golden("g2_train3_b4_64.npz")
    oracle, model = _pair(2, 3, "fp32", perturb_bn=False)
    if not np.array_equal(fingerprint(oracle), g["fingerprint0"]):
        pytest.skip("torch RNG stream differs from the build container")
    opt = model.fused_adamw(lr=1e-3)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=2e-3, steps_per_epoch=4, epochs=1, pct_start=0.3)
    x, m = torch.tensor(g["x"]).to(DEV), torch.tensor(g["mask"])
    _, t = P.prepare_training_batch(None, m, 2)
    t = t.to(DEV).float()
    model.train()
    for step in range(3):
        assert np.isclose(opt.param_groups[0]["lr"], g["lrs"][step], rtol=1e-12)
        assert np.isclose(opt.param_groups[0]["betas"][0], g["beta1"][step], rtol=1e-12)
        opt.zero_grad()
        loss = P.dice_loss_none(model(x), t)
        loss.backward()
        opt.step()
        sched.step()
        assert abs(loss.item() - g["losses"][step]) < 5e-4, (step, loss.item(), g["losses"][step])
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("after__"):
            assert torch.allclose(sd[k[7:]].cpu(), torch.tensor(g[k]), rtol=2e-2, atol=2e-3), k


def test_frozen_encoder_matches_reference_predicate():
    oracle, model = _pair(2, 3, "fp32", perturb_bn=False)
    for net in (oracle, model):
        for name, p in net.named_parameters():   # vol_seg_2d_trainer.py:106-108
            if all(["encoder" in name, "conv" in name]) and p.requires_grad:
                p.requires_grad = False
    assert sum(not p.requires_grad for p in model.parameters()) == 33
    x = torch.randn(2, 1, 64, 64, generator=torch.Generator().manual_seed(8))
    t = torch.nn.functional.one_hot((torch.rand(2, 64, 64) > 0.5).long(), 2).permute(0, 3, 1, 2).float()
    oracle.train(); model.train()
    P.dice_loss_none(oracle(x), t).backward()
    P.dice_loss_none(model(x.to(DEV)), t.to(DEV)).backward()
    ref = dict(oracle.named_parameters())
    for name, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None
        else:
            r = ref[name].grad
            assert ((p.grad.cpu() - r).norm() / (r.norm() + 1e-12)) < 2e-3, name
