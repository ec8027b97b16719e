"""-m gpu: the hot path at BASELINE.json's full sizes (config 2: 256^3 / 256^2 slices / batch 32 / 2 classes; config 3:
512^3 / 4 classes / 12 directions) checked through size-independent properties - the oracle cannot run these sizes in
seconds, so correctness of the big-grid code paths (K splits, direct kernels, XCD-aware tile maps, side-stream ordering,
key merge) is pinned by invariants that must hold exactly:
  * the same step twice gives bit-identical losses, parameters and optimiser state (fixed-order reductions, no float atomics);
  * the optimiser step hidden inside backward gives the same bits as backward() + step();
  * backward is linear: doubling dL/dlogits doubles every gradient exactly (powers of two commute with every rounding);
  * a 3-direction key-merged prediction equals the reference's formulation (three single-axis passes + two pairwise
    merges) bit for bit, equals itself when the slices are split into two shards and max-merged, and is idempotent.
Everything goes through the C ABI; synthetic data as in bench.py / BASELINE.md section 3."""
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # synthetic volumes / batches of the benchmark
from hip_helpers import DEV

pytestmark = pytest.mark.gpu


def _train_run(fuse: bool, steps: int = 3, precision: str = "bf16"):
    from volume_segmantics_amd.data.losses import HipDiceLoss
    from volume_segmantics_amd.engine import VolSegUnet
    model = VolSegUnet(2, device=DEV, precision=precision, seed=0)
    x, lab = bench.synth_batch(32, 256, 2, seed=1234)
    x = x.to(DEV)
    t = torch.nn.functional.one_hot(lab, 2).permute(0, 3, 1, 2).to(DEV, torch.uint8).float()
    opt = model.fused_adamw(lr=1e-4, fuse_step_into_backward=fuse)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=steps + 2, pct_start=0.3)
    crit = HipDiceLoss()
    model.train()
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        loss = crit(model(x), t)
        loss.backward()
        opt.step()
        sched.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    return losses, model._flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), model._bnstate.clone(), model


def test_config2_training_steps_are_deterministic_and_fusion_invariant():
    a = _train_run(fuse=True)
    b = _train_run(fuse=True)
    c = _train_run(fuse=False)
    assert all(np.isfinite(a[0])) and a[0][0] > a[0][-1] - 0.05          # finite, not diverging
    for other, what in ((b, "rerun"), (c, "unfused optimiser")):
        assert a[0] == other[0], (what, a[0], other[0])
        for u, v, nm in zip(a[1:5], other[1:5], ("params", "exp_avg", "exp_avg_sq", "bn running stats")):
            assert torch.equal(u, v), (what, nm)


def test_config2_backward_is_linear_in_the_output_gradient():
    from volume_segmantics_amd.engine import VolSegUnet
    model = VolSegUnet(2, device=DEV, precision="bf16", seed=0)
    x, _ = bench.synth_batch(32, 256, 2, seed=99)
    x = x.to(DEV)
    g = torch.randn(32, 2, 256, 256, device=DEV) * 1e-3
    model.train()
    grads = []
    for scale in (1.0, 2.0):
        model.zero_grad(set_to_none=True)
        model(x).backward(g * scale)
        torch.cuda.synchronize()
        grads.append(model._flat_grad.clone())
    assert torch.isfinite(grads[0]).all() and grads[0].abs().max() > 0
    assert torch.equal(grads[0] * 2.0, grads[1])


def _predictor(classes: int, batch: int, precision: str = "bf16", encoder: str = "resnet34", topology: str = "unet"):
    from volume_segmantics_amd.engine import VolSegUnet
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor
    model = VolSegUnet(classes, device=DEV, precision=precision, seed=1, encoder=encoder, topology=topology)
    with torch.no_grad():   # centre the head bias so that a random-init network uses every class
        model.eval()
        mean_logit = model(torch.randn(4, 1, 256, 256, generator=torch.Generator().manual_seed(7)).to(DEV)).mean(dim=(0, 2, 3))
        dict(model.named_parameters())["segmentation_head.0.bias"].sub_(mean_logit)
    pred = VolSeg2dPredictor.__new__(VolSeg2dPredictor)
    pred.model, pred.num_labels, pred.label_codes = model, classes, {}
    pred.settings = SimpleNamespace(cuda_device=0, prediction_batch_size=batch)
    return pred


def test_config2_three_way_prediction_equals_reference_formulation_and_is_idempotent():
    from volume_segmantics_amd.utilities.base_data_utils import Axis
    pred = _predictor(2, 32)
    vol = bench.synth_volume(256, seed=1234)
    labels, probs = pred._predict_3_ways_max_probs(vol)
    assert labels.shape == vol.shape and labels.dtype == np.uint8 and probs.dtype == np.float16
    assert 0 < (labels == 1).mean() < 1
    # the reference's own sequence (vol_seg_2d_predictor.py:67-88): Z, then Y merged in, then X merged in
    lab = np.empty((2,) + vol.shape, np.uint8)
    prb = np.empty((2,) + vol.shape, np.float16)
    lab[0], prb[0] = pred._predict_single_axis(vol, axis=Axis.Z)
    for ax in (Axis.Y, Axis.X):
        lab[1], prb[1] = pred._predict_single_axis(vol, axis=ax)
        pred._merge_vols_in_mem(prb, lab)
    assert np.array_equal(labels, lab[0]) and np.array_equal(probs.view(np.uint16), prb[0].view(np.uint16))
    # idempotence
    l2, p2 = pred._predict_3_ways_max_probs(vol)
    assert np.array_equal(labels, l2) and np.array_equal(probs.view(np.uint16), p2.view(np.uint16))


def test_twelve_way_prediction_without_the_repeated_directions_is_bit_identical():
    """Four of the reference's twelve directions repeat earlier ones exactly (tests/test_host_logic.py pins the table) and the
    first-wins merge can never take a repeat: the predictor's default (8 forward passes) must return the volume of all twelve
    passes bit for bit - labels and fp16 probabilities - on an odd-shaped volume (pads, crops, non-cubic rotations) and for the
    reference's own pairwise formulation of the twelve directions."""
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import REPEATED_DIRECTIONS, direction_views
    from volume_segmantics_amd.utilities.base_data_utils import Axis
    pred = _predictor(4, 16)
    vol = bench.synth_volume(96, seed=77)[:61, 3:83, :72].copy()
    with torch.no_grad():   # centre the head on THIS volume's slices: a random-init network must use several classes for the test to bite
        xs = torch.from_numpy(((vol[:8, :64, :64].astype(np.float32) / 255) - 0.449) / 0.226).unsqueeze(1).to(DEV)
        dict(pred.model.named_parameters())["segmentation_head.0.bias"].sub_(pred.model(xs).mean(dim=(0, 2, 3)))
    pred.settings.dedup_directions = True
    l8, p8 = pred._predict_12_ways_max_probs(vol)
    assert pred.last_timings["directions_run"] == 8
    pred.settings.dedup_directions = False
    l12, p12 = pred._predict_12_ways_max_probs(vol)
    assert pred.last_timings["directions_run"] == 12
    assert len(np.unique(l12)) >= 2 and 0.02 < (l12 == l12.flat[0]).mean() < 0.98
    assert np.array_equal(l8, l12) and np.array_equal(p8.view(np.uint16), p12.view(np.uint16))
    # the vote form (one-hot variants): a repeat is not run, its earlier twin votes twice
    pred.settings.dedup_directions = True
    v8 = pred._predict_12_ways_one_hot(vol)
    assert pred.last_timings["directions_run"] == 8
    pred.settings.dedup_directions = False
    v12 = pred._predict_12_ways_one_hot(vol)
    assert v12.shape == (4,) + vol.shape and (v12.sum(0) == 12).all() and np.array_equal(v8, v12)
    # and the reason: a repeated direction predicts, voxel for voxel, what its earlier twin predicted
    views = direction_views(np.arange(vol.size).reshape(vol.shape), 12)
    for later, earlier in REPEATED_DIRECTIONS.items():
        assert np.array_equal(views[later], views[earlier][::-1])


def test_config3_twelve_way_shards_merge_to_the_unsharded_volume(monkeypatch):
    """Two ranks' work done one after the other on this GPU: each takes its contiguous half of every direction's slices
    into its own key volume; the elementwise max of the two (what all_reduce(MAX) computes) must equal the one-rank
    result.  512^3, 4 classes, 12 directions."""
    from volume_segmantics_amd import dist as vdist
    from volume_segmantics_amd.model.operations import vol_seg_2d_predictor as P2
    pred = _predictor(4, 64)
    vol = bench.synth_volume(512, seed=5678)
    labels, probs = pred._predict_12_ways_max_probs(vol)
    hist = np.bincount(labels.ravel(), minlength=4)
    assert hist.sum() == vol.size and (hist > 0).sum() >= 2
    keys = []

    class KeepKeys(P2.HipBackend):
        def exchange(self, *args):
            keys.append(self.keys[:self.nvox].clone())

    monkeypatch.setattr(P2.VolSeg2dPredictor, "backend_factory", KeepKeys)
    for rank in (0, 1):
        monkeypatch.setattr(vdist, "world", lambda r=rank: (r, 2))
        pred._predict_12_ways_max_probs(vol)
    monkeypatch.undo()
    merged = torch.maximum(keys[0], keys[1])
    from volume_segmantics_amd import _lib
    lab2 = torch.empty(vol.size, dtype=torch.uint8, device=DEV)
    prb2 = torch.empty(vol.size, dtype=torch.float16, device=DEV)
    _lib.check(_lib.lib.vs_keys_unpack(_lib.ptr(merged), _lib.ptr(lab2), _lib.ptr(prb2), vol.size, _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(labels.ravel(), lab2.cpu().numpy())
    assert np.array_equal(probs.view(np.uint16).ravel(), prb2.cpu().numpy().view(np.uint16))


def test_odd_shaped_volume_twelve_way_equals_reference_formulation():
    """A non-cubic volume whose sides are not multiples of 32 (pad / crop offsets differ per direction, rot90 views have
    swapped dims, the staged key scatter sees ragged pixel tiles and batches): the one-pass packed-key 12-way prediction
    against the reference's own sequence (vol_seg_2d_predictor.py:67-116: 3-way per rotation, rot90 back, pairwise merges),
    built from single-axis passes and vs_merge_maxprob.  Bit-equal labels and probabilities."""
    from volume_segmantics_amd.utilities.base_data_utils import Axis
    pred = _predictor(4, 16)
    vol = np.ascontiguousarray(bench.synth_volume(96, seed=31)[:70, :45, :83])
    with torch.no_grad():   # centre the head bias on this volume's own slices so that several classes occur
        x = torch.zeros(8, 1, 64, 96)
        x[:, 0, :45, :83] = (torch.from_numpy(vol[::9][:8].astype(np.float32)) / 255 - 0.449) / 0.226
        mean_logit = pred.model(x.to(DEV)).mean(dim=(0, 2, 3))
        dict(pred.model.named_parameters())["segmentation_head.0.bias"].sub_(mean_logit)
    labels, probs = pred._predict_12_ways_max_probs(vol)
    assert labels.shape == vol.shape and len(np.unique(labels)) >= 2 and len(np.unique(probs)) > 100

    def three_way(v):
        lab = np.empty((2,) + v.shape, np.uint8)
        prb = np.empty((2,) + v.shape, np.float16)
        lab[0], prb[0] = pred._predict_single_axis(v, axis=Axis.Z)
        for ax in (Axis.Y, Axis.X):
            lab[1], prb[1] = pred._predict_single_axis(v, axis=ax)
            pred._merge_vols_in_mem(prb, lab)
        return lab[0], prb[0]

    lab = np.empty((2,) + vol.shape, np.uint8)
    prb = np.empty((2,) + vol.shape, np.float16)
    lab[0], prb[0] = three_way(vol)
    v = vol
    for k in range(1, 4):
        v = np.rot90(v)
        l, p = three_way(np.ascontiguousarray(v))
        lab[1], prb[1] = np.rot90(l, -k), np.rot90(p, -k)
        pred._merge_vols_in_mem(prb, lab)
    assert np.array_equal(labels, lab[0])
    assert np.array_equal(probs.view(np.uint16), prb[0].view(np.uint16))


def test_config5_sized_volume_1024_cube_three_axis_prediction_stays_resident_and_addresses_correctly():
    """BASELINE configs[4]'s volume (1024^3, 2 classes, 3-axis prediction, everything HBM-resident: 1 GB volume, 4.3 GB of packed
    keys, 3 GB of labels + probabilities) with the network this engine has for it (U-Net / ResNet-34): 2^30 voxels cross every
    32-bit element-offset boundary of the volume-side kernels.  Properties: the result has the volume's shape, single-axis
    slices of the big run equal the same slices predicted on their own (evaluation-mode BatchNorm: a slice's logits do not
    depend on its batch), and the key-merged 3-axis result agrees with the single-axis one wherever that axis won."""
    import time
    from volume_segmantics_amd.utilities.base_data_utils import Axis
    pred = _predictor(2, 16)
    small = bench.synth_volume(256, seed=1234)
    vol = np.tile(small, (4, 4, 4))
    assert vol.shape == (1024, 1024, 1024)
    t0 = time.perf_counter()
    labels, probs = pred._predict_3_ways_max_probs(vol)
    dt = time.perf_counter() - t0
    print(f"1024^3 three-axis prediction (first call, incl. plans): {dt:.2f} s")
    assert labels.shape == vol.shape and labels.dtype == np.uint8 and probs.shape == vol.shape and probs.dtype == np.float16
    assert 0 < (labels[::8, ::8, ::8] == 1).mean() < 1
    lz, pz = pred._predict_single_axis(vol, axis=Axis.Z)
    for k in (0, 517, 1023):                       # slices of the full run vs the same slice as a one-slice volume
        l1, p1 = pred._predict_single_axis(vol[k:k + 1], axis=Axis.Z)
        assert np.array_equal(lz[k], l1[0]) and np.array_equal(pz[k].view(np.uint16), p1[0].view(np.uint16)), k
    # the merge keeps the larger fp16 probability: wherever the merged probability equals the Z pass's, the Z label survived
    won = probs[1000:1008] == pz[1000:1008]
    assert won.mean() > 0.05 and np.array_equal(labels[1000:1008][won], lz[1000:1008][won])
    assert (probs[1000:1008] >= pz[1000:1008]).all()


def test_config5_network_deeplabv3plus_efficientnet_b4_on_1024_square_slices():
    """BASELINE configs[4]'s NETWORK (smp.DeepLabV3Plus over efficientnet-b4, 2 classes) on 1024 x 1024 slices of its volume: a 96-slice
    slab predicted along Z in batches of 8 (the encoder's expansions are 144 channels at 512 x 512: 75 MB per slice and tensor).
    Properties as above: shape / dtypes, both classes present, a slice of the run equals the same slice predicted on its own."""
    import time
    from volume_segmantics_amd.utilities.base_data_utils import Axis
    pred = _predictor(2, 8, encoder="efficientnet-b4", topology="deeplabv3plus")
    small = bench.synth_volume(256, seed=99)
    vol = np.tile(small[:96], (1, 4, 4))
    assert vol.shape == (96, 1024, 1024)
    lz, pz = pred._predict_single_axis(vol[:8], axis=Axis.Z)          # plans, weight copies
    t0 = time.perf_counter()
    lz, pz = pred._predict_single_axis(vol, axis=Axis.Z)
    dt = time.perf_counter() - t0
    print(f"DeepLabV3+ / efficientnet-b4, 96 slices of 1024^2: {dt:.2f} s = {96 / dt:.0f} slices/s")
    assert lz.shape == vol.shape and lz.dtype == np.uint8 and pz.dtype == np.float16
    assert 0 < (lz[::4, ::8, ::8] == 1).mean() < 1
    for k in (0, 50, 95):
        l1, p1 = pred._predict_single_axis(vol[k:k + 1], axis=Axis.Z)
        assert np.array_equal(lz[k], l1[0]) and np.array_equal(pz[k].view(np.uint16), p1[0].view(np.uint16)), k


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_config5_full_volume_with_its_own_network_three_axis_1024_cube(precision):
    """BASELINE configs[4] as it stands: a 1024^3 volume, 2 classes, 3-axis max-probability prediction with smp.DeepLabV3Plus over
    efficientnet-b4 in fp16 - the precision BASELINE names - and in bf16 (batches of 8 slices of 1024 x 1024; volume, packed keys
    and outputs resident in HBM).  Properties: shapes /
    dtypes, both classes present, the merged probability is nowhere below the Z pass's and where it equals it the Z label survived."""
    import time
    from volume_segmantics_amd.utilities.base_data_utils import Axis
    pred = _predictor(2, 8, precision=precision, encoder="efficientnet-b4", topology="deeplabv3plus")
    vol = np.tile(bench.synth_volume(256, seed=4321), (4, 4, 4))
    assert vol.shape == (1024, 1024, 1024)
    pred._predict_single_axis(vol[:8], axis=Axis.Z)                 # plans, weight copies
    t0 = time.perf_counter()
    labels, probs = pred._predict_3_ways_max_probs(vol)
    dt = time.perf_counter() - t0
    print(f"configs[4]: 1024^3, 3 axes, DeepLabV3+ / efficientnet-b4 {precision}: {dt:.2f} s = {3072 / dt:.0f} slices/s")
    assert labels.shape == vol.shape and labels.dtype == np.uint8 and probs.shape == vol.shape and probs.dtype == np.float16
    assert 0 < (labels[::8, ::8, ::8] == 1).mean() < 1
    lz, pz = pred._predict_single_axis(vol[512:520], axis=Axis.Z)
    won = probs[512:520] == pz
    assert (probs[512:520] >= pz).all() and won.mean() > 0.05 and np.array_equal(labels[512:520][won], lz[won])


@pytest.mark.parametrize("topology,encoder", [("unetplusplus", "efficientnet-b3"), ("unet", "timm-resnest101e"), ("manet", "efficientnet-b4")])
def test_bench_sized_training_step_for_pairs_the_parity_tests_see_small(topology, encoder):
    """One bf16 training step at the bench shape (256^2, batch 32) for encoder / decoder pairs whose parity tests run at 64^2, batch 4: the
    tile picks and buffer sizes that depend on the problem size (timm-resnest101e's 128-channel max-pool once overran a 64-channel argmax
    buffer; U-Net++ / efficientnet-b3 splits a data gradient at 32 + 40 channels, which a 64-wide cout tile cannot) - finite loss, finite
    non-zero gradients, and a second step that moves the loss."""
    from volume_segmantics_amd.data.losses import HipDiceLoss
    from volume_segmantics_amd.engine import VolSegUnet
    g = torch.Generator().manual_seed(5)
    x = torch.randn(32, 1, 256, 256, generator=g).to(DEV)
    t = torch.nn.functional.one_hot((torch.rand(32, 256, 256, generator=g) > 0.5).long(), 2).permute(0, 3, 1, 2).contiguous().to(torch.uint8).to(DEV)
    m = VolSegUnet(2, device=DEV, precision="bf16", seed=0, encoder=encoder, topology=topology)
    o = m.fused_adamw(lr=1e-3, fuse_step_into_backward=True)
    m.train()
    losses = []
    for _ in range(2):
        o.zero_grad()
        loss = HipDiceLoss()(m(x), t)
        loss.backward()
        o.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)) and losses[0] != losses[1], losses
    assert torch.isfinite(m._flat_grad).all() and m._flat_grad.abs().sum().item() > 0 and torch.isfinite(m._flat).all()
