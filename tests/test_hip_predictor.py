"""-m gpu: VolSeg2dPredictor / VolSeg2DPredictionManager / VolSeg2dTrainer on the HIP engine against goldens produced by
the reference's own predictor code (oracle/gen_goldens.py G3-G5) and against the CPU oracle."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import fingerprint
from hip_helpers import DEV
from oracle import predictor_numpy as P
from oracle.unet_resnet34_torch import seeded_oracle
from volume_segmantics_amd.utilities.base_data_utils import Axis, ModelType, Quality

pytestmark = pytest.mark.gpu


def _ckpt(tmp_path, classes, seed=0):
    from volume_segmantics_amd.checkpoint_compat import reference_pickle_enum
    net = seeded_oracle(classes, seed)
    path = tmp_path / "model.pytorch"
    torch.save({"model_state_dict": net.state_dict(),
                "model_struc_dict": {"type": reference_pickle_enum(ModelType.U_NET), "encoder_name": "resnet34",
                                     "encoder_weights": "imagenet", "in_channels": 1, "classes": classes},
                "optimizer_state_dict": {}, "loss_val": 0.1, "label_codes": {"fg": 1}}, path)
    return net, path


def _settings(**kw):
    base = dict(quality="low", output_probs=False, clip_data=False, st_dev_factor=2.575, data_hdf5_path="/data",
                cuda_device=0, downsample=False, one_hot=False, prediction_axis="Z", prediction_batch_size=7)
    base.update(kw)
    return SimpleNamespace(**base)


def _ulp(a, b):
    return np.abs(a.view(np.int16).astype(np.int32) - b.view(np.int16).astype(np.int32))


@pytest.fixture(scope="module")
def predictor_and_golden(golden, tmp_path_factory):
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor
    g = golden("g3_predict_29x64x40_c4.npz")
    net, path = _ckpt(tmp_path_factory.mktemp("ck"), 4)
    if not np.array_equal(fingerprint(net), g["fingerprint"]):
        pytest.fail("torch RNG stream differs from the build container: the reference-generated goldens cannot be checked (regenerate them with oracle/gen_goldens.py on this torch)")
    pred = VolSeg2dPredictor(str(path), _settings())
    assert pred.num_labels == 4 and pred.label_codes == {"fg": 1} and pred.model.precision == "fp32"
    return pred, g, net


@pytest.fixture(scope="module")
def decidable(golden):
    """tests/golden/g3_decidable.npz (oracle/gen_goldens.py:gen_decidable): the voxels where no direction involved is within
    rounding of a tie - top-2 logit margin > 1e-3 in every contributing direction and, for merges, a winner that is clear by
    more than 2 fp16 ulps (or the same label among all near-winners).  On those the reference's label must be reproduced
    bit for bit; elsewhere two correct fp32 implementations may legitimately differ."""
    d = golden("g3_decidable.npz")
    shape = tuple(d["shape"])
    n = int(np.prod(shape))
    return {k: np.unpackbits(d[k])[:n].astype(bool).reshape(shape) for k in d.files if d[k].dtype == np.uint8 and k != "shape"}


def _check(labels, probs, ref_l, ref_p, what, mask, min_cover=0.99):
    assert labels.dtype == np.uint8 and labels.shape == ref_l.shape
    bad = labels != ref_l
    print(f"[parity] {what}: decidable {mask.mean():.5f} of voxels; mismatches {int(bad.sum())} total, "
          f"{int((bad & mask).sum())} on decidable voxels")
    assert mask.mean() > min_cover, (what, mask.mean())
    assert not (bad & mask).any(), (what, int((bad & mask).sum()))       # north_star: label volumes bit-exact
    assert bad.mean() <= 2e-4, (what, bad.mean())                          # and the undecidable rest stays a handful
    if probs is not None:
        assert probs.dtype == np.float16
        # on decidable voxels the winning direction is the reference's: its fp16 probability may differ by the last bits
        # (expf vs torch's vectorised exp, 1e-4-level logit differences): <= 2 ulp, and <= 1 ulp on 99.9 %
        u = _ulp(probs, ref_p)[mask]
        assert u.max() <= 2 and (u <= 1).mean() > 0.999, (what, int(u.max()))


def test_single_axis_all_axes_vs_reference_golden(predictor_and_golden, decidable):
    pred, g, _ = predictor_and_golden
    for ax, name in ((Axis.Z, "z"), (Axis.Y, "y"), (Axis.X, "x")):   # Y / X exercise padding and the d=3 crop quirk
        l, p = pred._predict_single_axis(g["vol"], output_probs=True, axis=ax)
        _check(l, p, g[f"single_{name}_labels"], g[f"single_{name}_probs"], name, decidable[f"single_{name}"])
    l, p = pred._predict_single_axis(g["vol"], output_probs=False)
    assert p is None and l.dtype == np.uint8


def test_three_and_twelve_way_merge_vs_reference_golden(predictor_and_golden, decidable):
    pred, g, _ = predictor_and_golden
    l, p = pred._predict_3_ways_max_probs(g["vol"])
    _check(l, p, g["three_labels"], g["three_probs"], "3-way", decidable["three"])
    l, p = pred._predict_12_ways_max_probs(g["vol"])
    # 12 directions of a RANDOM-INIT 4-class network: ~1.7 % of the voxels have some direction within 1e-3 of a class tie
    _check(l, p, g["twelve_labels"], g["twelve_probs"], "12-way", decidable["twelve"], min_cover=0.98)


def test_one_hot_votes_vs_reference_golden(predictor_and_golden, decidable):
    pred, g, _ = predictor_and_golden
    for fn, key, n, mk, cover in ((pred._predict_single_axis_to_one_hot, "onehot_z", 1, "votes_z", 0.99),
                                  (pred._predict_3_ways_one_hot, "onehot_three", 3, "votes_three", 0.99),
                                  (pred._predict_12_ways_one_hot, "onehot_twelve", 12, "votes_twelve", 0.98)):
        oh = fn(g["vol"])
        assert oh.dtype == np.uint8 and oh.shape == g[key].shape and (oh.sum(0) == n).all()
        mask = decidable[mk]
        bad = (oh != g[key]).any(0)
        print(f"[parity] {key}: decidable {mask.mean():.5f}; voxels with a different vote {int(bad.sum())}, on decidable {int((bad & mask).sum())}")
        assert mask.mean() > cover and not (bad & mask).any(), key        # every vote count equal where every direction is decidable
        assert (oh != g[key]).mean() <= 4e-4, key


def test_merge_vols_in_mem_is_the_reference_pairwise_merge(predictor_and_golden, golden):
    pred, _, _ = predictor_and_golden
    g = golden("g4_merge_ties.npz")
    lab = np.stack([g["dlabels"][0], g["dlabels"][1]])
    prb = np.stack([g["dprobs"][0], g["dprobs"][1]])
    pred._merge_vols_in_mem(prb, lab)
    assert np.array_equal(lab[0], g["chain_labels"][0]) and np.array_equal(prb[0].view(np.uint16), g["chain_probs"][0].view(np.uint16))


def test_bit_exact_labels_where_the_oracle_margin_is_clear(predictor_and_golden):
    """north_star: label volumes bit-exact after argmax.  Every voxel whose top-2 logit margin in the CPU oracle
    exceeds the fp32 logit tolerance (1e-3) must carry exactly the oracle's label."""
    pred, g, net = predictor_and_golden
    vol = g["vol"]
    ref_l, _, logits = P.predict_single_axis(net, vol, 0, return_logits=True)
    top2 = np.sort(logits, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-3
    l, _ = pred._predict_single_axis(vol, axis=Axis.Z)
    assert clear.mean() > 0.99 and np.array_equal(l[clear], ref_l[clear])


def test_prediction_does_not_depend_on_the_batch_size(predictor_and_golden, tmp_path):
    """utilities/base_data_utils.py get_batch_size: the reference predicts 4 (or 2) slices per forward, this engine
    HIP_PRED_BATCH = 32 by default.  Eval-mode results are per slice (running BN statistics, no cross-sample op), so the
    labels and fp16 probabilities are the same bits for every batch size - including the reference's 4 and 2."""
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor
    pred, g, _ = predictor_and_golden
    out = {}
    for bs in (2, 4, 7, 32):
        pred.settings.prediction_batch_size = bs
        out[bs] = pred._predict_3_ways_max_probs(g["vol"])
    pred.settings.prediction_batch_size = 7
    for bs in (2, 4, 32):
        assert np.array_equal(out[bs][0], out[7][0]), bs
        assert np.array_equal(out[bs][1].view(np.uint16), out[7][1].view(np.uint16)), bs


def _blurred_volume(shape, seed):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(shape).astype(np.float32)
    for ax in range(3):      # cheap separable smoothing: the slices then look like images, not noise
        v = (np.roll(v, 1, ax) + 2 * v + np.roll(v, -1, ax)) / 4
    return np.clip(128 + 160 * v, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("precision,depth,batches", [("bf16", 131, (1, 3, 32, 128)), ("fp16", 131, (1, 32, 128)), ("fp32", 35, (1, 3, 32))])
def test_prediction_at_512_does_not_depend_on_the_batch_size_across_the_kernel_thresholds(tmp_path, precision, depth, batches):
    """The premise of leaving out the repeated directions (a repeat predicts what its twin predicted) and of every claim that the
    batch size is a free parameter: a slice's label and fp16 probability bits do not depend on how many slices share its forward
    pass.  The kernel chosen for a layer DOES depend on the launch size (csrc/conv_igemm.hip: `conv_direct_min_px`,
    `conv_nw8_min_wgs`, `conv_min_wgs`, `conv_stream_min_tiles`, the persistent kernel's grid), so this runs 512 x 512 slices - where
    batches of 1, 3, 32 and 128 land on different sides of every one of those thresholds - with a ragged last batch each (131 = 4 x 32
    + 3 = 128 + 3), and compares every voxel bit for bit.  Reference: the batch loop of _predict_single_axis,
    vol_seg_2d_predictor.py:38-58."""
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor
    _, path = _ckpt(tmp_path, 4)
    pred = VolSeg2dPredictor(str(path), _settings(precision=precision))
    assert pred.model.precision == precision
    vol = _blurred_volume((depth, 512, 512), 77)
    out = {}
    for bs in batches:
        pred.settings.prediction_batch_size = bs
        out[bs] = pred._predict_single_axis(vol, axis=Axis.Z)
    ref = out[batches[0]]
    assert len(np.unique(ref[0])) >= 2
    for bs in batches[1:]:
        dl = int((out[bs][0] != ref[0]).sum())
        dp = int((out[bs][1].view(np.uint16) != ref[1].view(np.uint16)).sum())
        assert dl == 0 and dp == 0, f"batch {bs} vs batch {batches[0]}: {dl} labels, {dp} probabilities differ"


def test_twelve_way_without_the_repeats_equals_all_twelve_when_twins_fall_into_ragged_batches(tmp_path):
    """dedup_directions at 512-pixel slices with a batch size that does not divide the stacks: a repeated direction visits its twin's
    slices in REVERSED order, so a slice that sat in a full batch in the twin lands in the ragged tail of the repeat (another
    launch size, possibly another kernel).  Labels, fp16 probabilities and the vote volume must still be identical with and without
    the four repeats.  Reference: _predict_12_ways_max_probs / _predict_12_ways_one_hot, vol_seg_2d_predictor.py:100-136."""
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor
    _, path = _ckpt(tmp_path, 4)
    pred = VolSeg2dPredictor(str(path), _settings(precision="bf16", prediction_batch_size=27))
    vol = _blurred_volume((45, 512, 480), 78)          # stacks of 45, 512 and 480 slices; 512 = 18 x 27 + 26, 480 = 17 x 27 + 21
    res = {}
    for dedup in (True, False):
        pred.settings.dedup_directions = dedup
        res[dedup] = (pred._predict_12_ways_max_probs(vol), pred._predict_12_ways_one_hot(vol))
        assert pred.last_timings["directions_run"] == (8 if dedup else 12)
    (l1, p1), v1 = res[True]
    (l0, p0), v0 = res[False]
    assert np.array_equal(l1, l0) and np.array_equal(p1.view(np.uint16), p0.view(np.uint16))
    assert np.array_equal(v1, v0) and (v1.sum(0) == 12).all()


def test_prediction_settings_precision_fp16_overrides_the_checkpoint(predictor_and_golden, tmp_path):
    """`precision: fp16` in the PREDICTION settings (BASELINE configs[4]) loads a checkpoint trained in any precision into the fp16
    inference engine: labels equal the reference golden's wherever the oracle's margin clears fp16's error, the fp16 probabilities
    stay within a few ulps of the fp32 engine's."""
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor
    pred32, g, net = predictor_and_golden
    _, path = _ckpt(tmp_path, 4)
    pred = VolSeg2dPredictor(str(path), _settings(precision="fp16"))
    assert pred.model.precision == "fp16"
    l16, p16 = pred._predict_3_ways_max_probs(g["vol"])
    l32, p32 = pred32._predict_3_ways_max_probs(g["vol"])
    same = l16 == l32
    print(f"[fp16] 3-way labels equal to the fp32 engine's on {same.mean():.5f} of the voxels")
    assert same.mean() > 0.98
    d = np.abs(p16.astype(np.float32) - p32.astype(np.float32))
    assert d[same].max() < 3e-2 and d[same].mean() < 3e-3      # (1.4e-2 / 1.0e-3 measured: a random-init network, softmax of ~1e-3-accurate logits)
    ref_l, _, logits = P.predict_single_axis(net, g["vol"], 0, return_logits=True)
    top2 = np.sort(logits, axis=1)[:, -2:]
    margin = top2[:, 1] - top2[:, 0]
    lz, _ = pred._predict_single_axis(g["vol"], axis=Axis.Z)
    bad = lz != ref_l
    print(f"[fp16] single axis vs the CPU oracle: {int(bad.sum())} of {bad.size} labels differ; largest oracle margin among them {margin[bad].max() if bad.any() else 0:.3e}")
    # (a random-init 4-class network: logits of magnitude ~1 whose fp16 error is ~1e-2 at worst - a label can only flip inside that)
    assert bad.mean() < 5e-3 and (not bad.any() or margin[bad].max() < 0.1) and (margin > 0.1).mean() > 0.5


def test_unclipped_uint16_and_float_volumes_predict_like_the_oracle(predictor_and_golden):
    """clip_data: False hands the predictor the volume in its own dtype; the reference then feeds float32(v) / 255 (integers,
    any range) or v itself (floats) to the network (data/datasets.py:128-134).  Labels must equal the oracle's wherever its
    top-2 logit margin is clear."""
    pred, g, net = predictor_and_golden
    rng = np.random.default_rng(5)
    base = g["vol"][:9].astype(np.float32)
    for vol in ((base * 3 + rng.integers(0, 3, base.shape)).astype(np.uint16),      # values up to ~767: inputs > 1 after / 255
                (base / 255).astype(np.float32)):                                      # a float volume already in [0, 1]
        ref_l, ref_p, logits = P.predict_single_axis(net, vol, 0, return_logits=True)
        top2 = np.sort(logits, axis=1)[:, -2:]
        # uint16 inputs of up to 3 make logits (and their fp32 rounding errors) larger: scale the margin with the logit size
        clear = (top2[:, 1] - top2[:, 0]) > 1e-3 * max(1.0, float(np.abs(logits).max()) / 10)
        l, p = pred._predict_single_axis(vol, axis=Axis.Z)
        assert l.dtype == np.uint8 and p.dtype == np.float16 and clear.mean() > 0.97
        assert np.array_equal(l[clear], ref_l[clear]), vol.dtype


def test_imagenet_encoder_weights_are_never_silently_random(tmp_path, monkeypatch):
    """smp would download the ImageNet encoder; with no network that is an error, not a silent random initialisation."""
    from volume_segmantics_amd.model.model_2d import create_model_on_device
    monkeypatch.delenv("VOLSEG_RESNET34_WEIGHTS", raising=False)
    struct = {"type": "U_Net", "encoder_name": "resnet34", "encoder_weights": "imagenet", "in_channels": 1, "classes": 2}
    with pytest.raises(FileNotFoundError, match="allow_random_encoder"):
        create_model_on_device(0, struct)
    assert create_model_on_device(0, dict(struct, allow_random_encoder=True)).classes == 2
    assert create_model_on_device(0, dict(struct, encoder_weights=None)).classes == 2
    # a local torchvision-shaped state dict is used when supplied
    tv = {k[len("encoder."):]: v for k, v in seeded_oracle(2, 5).state_dict().items() if k.startswith("encoder.")}
    tv["conv1.weight"] = tv["conv1.weight"].repeat(1, 3, 1, 1) / 3        # 3-channel stem, summed by the loader
    torch.save(tv, tmp_path / "r34.pth")
    monkeypatch.setenv("VOLSEG_RESNET34_WEIGHTS", str(tmp_path / "r34.pth"))
    m = create_model_on_device(0, struct)
    assert torch.allclose(m.state_dict()["encoder.layer2.1.conv1.weight"].cpu(), tv["layer2.1.conv1.weight"])


def test_prediction_manager_qualities_and_outputs(tmp_path):
    """Mirrors the reference's tests/test_vol_seg_prediction_manager.py:26-113 (types, shapes, files written)."""
    from volume_segmantics_amd.model.operations.vol_seg_prediction_manager import VolSeg2DPredictionManager
    _, path = _ckpt(tmp_path, 4)
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 255, size=(20, 37, 45)).astype(np.int64)   # reference fixture: random int volume
    for q, one_hot in ((Quality.LOW, False), (Quality.MEDIUM, False), (Quality.HIGH, False), (Quality.MEDIUM, True)):
        mgr = VolSeg2DPredictionManager(str(path), vol, _settings(clip_data=True, one_hot=one_hot, output_probs=True, prediction_axis="Y"))
        assert mgr.data_vol.dtype == np.uint8 and mgr.get_label_codes() == {"fg": 1}
        out = tmp_path / f"out_{q.name}_{one_hot}.npy"
        pred = mgr.predict_volume_to_path(out, q)
        assert pred.dtype == np.uint8 and out.exists()
        assert pred.shape == ((4,) + vol.shape if one_hot else vol.shape)
        assert (tmp_path / f"out_{q.name}_{one_hot}_probs.h5").exists() == (not one_hot)     # the reference's hard-coded name (:94-98)
    assert mgr.predict_volume_to_path(None).shape == (4,) + vol.shape   # quality from settings ("low")... one_hot


def test_prediction_manager_reads_and_writes_hdf5_like_the_reference(tmp_path):
    """The reference's own file format end to end on the box (vol_seg_prediction_manager.py:90-99, base_data_utils.py:179-214,
    351-356): the volume comes from an HDF5 file (internal path `data_hdf5_path`), labels and fp16 probabilities go to
    chunked, gzip-compressed HDF5 files - through h5py where installed, through utilities/hdf5_lite.py (libhdf5) here."""
    from volume_segmantics_amd.model.operations.vol_seg_prediction_manager import VolSeg2DPredictionManager
    from volume_segmantics_amd.utilities import base_data_utils as utils, hdf5_lite
    if utils._h5py() is None and not hdf5_lite.available():
        pytest.skip("neither h5py nor libhdf5 on this machine")
    _, path = _ckpt(tmp_path, 4)
    vol = np.random.default_rng(4).integers(0, 4000, size=(18, 40, 36)).astype(np.uint16)
    utils.save_data_to_hdf5(vol, tmp_path / "scan.h5", internal_path="/entry/data")
    mgr = VolSeg2DPredictionManager(str(path), tmp_path / "scan.h5", _settings(clip_data=True, output_probs=True, data_hdf5_path="/entry/data"))
    assert mgr.data_vol.dtype == np.uint8 and mgr.data_vol.shape == vol.shape
    out = tmp_path / "seg.h5"
    pred = mgr.predict_volume_to_path(out, Quality.MEDIUM)
    lab, chunks = utils.numpy_from_hdf5(out, "/data")
    prb, _ = utils.numpy_from_hdf5(tmp_path / "seg_probs.h5", "/data")
    assert np.array_equal(lab, pred) and lab.dtype == np.uint8 and chunks is not None
    assert prb.dtype == np.float16 and prb.shape == vol.shape and (prb > 0.25).all() and (prb <= 1).all()
    mgr2 = VolSeg2DPredictionManager(str(path), vol, _settings(clip_data=True, output_probs=True))      # the same volume as an array
    assert np.array_equal(mgr2.predict_volume_to_path(None, Quality.MEDIUM), pred)


@pytest.mark.parametrize("mtype,encoder", [("U_Net", "resnet34"), ("U_Net_Plus_Plus", "resnet34"), ("Linknet", "resnet34"), ("FPN", "resnet34"),
                                           ("DeepLabV3", "resnet34"), ("DeepLabV3_Plus", "resnet34"), ("MA_Net", "resnet34"),
                                           ("U_Net", "resnext50_32x4d"), ("U_Net", "efficientnet-b4"), ("DeepLabV3_Plus", "efficientnet-b3"), ("U_Net", "timm-resnest50d")])
def test_trainer_end_to_end_on_synthetic_slices(tmp_path, mtype, encoder):
    """1 frozen + 1 unfrozen epoch through LR finder, one-cycle schedule, early-stopping checkpoint and reload
    (reference flow: scripts/train_2d_model.py:56-71, tests/test_vol_seg_2d_trainer.py:95-116), for every topology the engine
    builds (settings `model: {type: ...}` as in the reference's 2d_model_train_settings.yaml) and a grouped-convolution encoder;
    the checkpoint's model_struc_dict rebuilds the same network in the predictor."""
    from torch.utils.data import DataLoader
    from volume_segmantics_amd.data.datasets import ArraySliceDataset
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor
    from volume_segmantics_amd.model.operations.vol_seg_2d_trainer import VolSeg2dTrainer
    torch.manual_seed(20)       # initial weights and shuffling order do not depend on which tests ran before
    rng = np.random.default_rng(0)
    field = rng.standard_normal((40, 64, 64)).astype(np.float32)
    for ax in (1, 2):
        field = (np.roll(field, 1, ax) + field + np.roll(field, -1, ax)) / 3
    imgs = np.clip(128 + 200 * field, 0, 255).astype(np.uint8)
    masks = (field > 0.1).astype(np.uint8)
    loaders = (DataLoader(ArraySliceDataset(imgs[:32], masks[:32]), batch_size=8, shuffle=True, drop_last=True),
               DataLoader(ArraySliceDataset(imgs[32:], masks[32:]), batch_size=8))
    # (the 4-step LR sweep of this toy run ends at its steepest point, i.e. picks end_lr / 3: 16.7 - four AdamW steps of that size leave
    # finite weights and, through the EfficientNets' swish / sigmoid gates and 100 BatchNorms with barely-moved running statistics, a
    # NaN validation loss; the ReLU ResNets shrug it off.  The EfficientNet cases sweep to a sane end point instead.)
    end_lr = 0.01 if encoder.startswith("efficientnet") else 50
    settings = SimpleNamespace(starting_lr=1e-6, end_lr=end_lr, lr_find_epochs=1, lr_reduce_factor=500, cuda_device=0, patience=3,
                               loss_criterion="DiceLoss", alpha=0.75, beta=0.25, eval_metric="MeanIoU", pct_lr_inc=0.3,
                               plot_lr_graph=False, image_size=64, training_set_proportion=0.8,
                               model={"type": mtype, "encoder_name": encoder, "encoder_weights": None})
    if encoder == "resnet34":      # the opt-in recorded step (`step_mode: graph`: replayed hipGraphs) for the ResNet-34 topologies; call by call otherwise
        settings.step_mode = "graph"
    tr = VolSeg2dTrainer(None, None, {"bg": 0, "fg": 1}, settings, loaders=loaders)
    out = tmp_path / "trained.pytorch"
    tr.train_model(out, 1, 3, create=True, frozen=True)
    assert out.exists() and len(tr.avg_train_losses) == 1 and np.isfinite(tr.avg_valid_losses[0])
    assert tr._count_trainable_parameters() < tr._count_parameters()
    tr.train_model(out, 1, 3, create=False, frozen=False)
    assert len(tr.avg_train_losses) == 2 and tr._count_trainable_parameters() == tr._count_parameters()
    tr.output_loss_fig(out)
    tr.output_prediction_figure(out)
    assert (tmp_path / "trained_train_stats.csv").exists()
    d = torch.load(out, weights_only=False)
    assert d["model_struc_dict"]["type"].name == mtype.upper() and d["label_codes"] == {"bg": 0, "fg": 1}
    pred = VolSeg2dPredictor(str(out), SimpleNamespace(cuda_device=0))
    assert pred.model.topology == {"U_Net": "unet", "U_Net_Plus_Plus": "unetplusplus", "Linknet": "linknet", "FPN": "fpn", "DeepLabV3": "deeplabv3",
                                   "DeepLabV3_Plus": "deeplabv3plus", "MA_Net": "manet"}[mtype]
    assert torch.equal(pred.model._flat, tr.model._flat)
    labels, probs = pred._predict_single_axis(imgs[:8])
    assert labels.shape == (8, 64, 64) and labels.max() <= 1 and probs.dtype == np.float16


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("classes,topology", [(2, "unet"), (4, "unet"), (6, "unet"), (3, "linknet")])
def test_head_scatter_epilogue_equals_forward_plus_logits_to_volume(precision, classes, topology):
    """vs_unet_forward_to_volume (head kernel writes labels / probabilities / keys itself) vs vs_unet_forward +
    vs_logits_to_volume on the same slices: bit-identical volumes.  6 classes exercises the fall-back inside the call, and so
    does Linknet's 1x1 head (the fused epilogue belongs to the 3x3 direct kernel)."""
    import numpy as np
    from volume_segmantics_amd import _lib
    from volume_segmantics_amd.engine import VolSegUnet
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import dirmap_of
    L = _lib
    old = L.lib.vs_get_option(b"conv_direct_min_px")
    L.set_option("conv_direct_min_px", 1)          # small test slices still take the direct head kernel
    try:
        model = VolSegUnet(classes, device=DEV, precision=precision, seed=3, topology=topology)
        model.eval()
        vol = np.zeros((5, 61, 90), np.uint8)                       # padded to 64 x 96: crop offsets are exercised
        for view, direction in ((vol, 0), (np.swapaxes(vol, 0, 1)[:, :5], 4), (np.rot90(vol, 1, (1, 2)), 7)):
            dmap = dirmap_of(vol, view)
            nb = min(3, view.shape[0])
            x = torch.randn(nb, 1, dmap.hp, dmap.wp, generator=torch.Generator().manual_seed(direction)).to(DEV)
            n = vol.size
            for mode in (0, 1):
                ref = dict(labels=torch.zeros(n, dtype=torch.uint8, device=DEV), probs=torch.zeros(n, dtype=torch.float16, device=DEV),
                           keys=torch.full((n,), 5, dtype=torch.int32, device=DEV))
                got = {k: v.clone() for k, v in ref.items()}
                with torch.no_grad():
                    logits = model._forward_impl(x, training=False)
                L.check(L.lib.vs_logits_to_volume(L.ptr(logits), classes, dmap, 1, nb, mode, direction, L.ptr(ref["labels"]),
                                                  L.ptr(ref["probs"]), L.ptr(ref["keys"]), None, n, L.stream_ptr()))
                model._forward_to_volume(x, dmap, 1, mode, direction, got["labels"], got["probs"], got["keys"], None, n)
                torch.cuda.synchronize()
                for k in ref:
                    assert torch.equal(ref[k], got[k]), (precision, classes, direction, mode, k)
                assert (ref["labels"] != 0).any() or mode == 1
    finally:
        L.set_option("conv_direct_min_px", old)


@pytest.mark.parametrize("classes", [2, 4])
def test_staged_key_scatter_for_slices_along_the_contiguous_axis(classes):
    """Directions whose slice index is the volume's contiguous axis: the head stages its keys slice-major and a transposing
    pass merges them into the key volume (vs_unet_forward_to_volume, batch >= 8) - same keys as forward + logits_to_volume,
    with cropping, a ragged last pixel tile, a batch that is not a multiple of anything and pre-existing larger keys."""
    import numpy as np
    from volume_segmantics_amd import _lib as L
    from volume_segmantics_amd.engine import VolSegUnet
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import dirmap_of
    old = L.lib.vs_get_option(b"conv_direct_min_px")
    L.set_option("conv_direct_min_px", 1)
    try:
        model = VolSegUnet(classes, device=DEV, precision="bf16", seed=5)
        model.eval()
        vol = np.zeros((45, 61, 13), np.uint8)
        for view, direction in ((np.swapaxes(vol, 0, 2), 2), (np.swapaxes(np.rot90(vol, 1), 0, 2), 5)):
            dmap = dirmap_of(vol, view)
            assert abs(dmap.ss) == 1 and abs(dmap.sw) != 1
            nb = 11
            x = torch.randn(nb, 1, dmap.hp, dmap.wp, generator=torch.Generator().manual_seed(direction)).to(DEV)
            n = vol.size
            g = torch.Generator().manual_seed(1)
            # a third of the voxels already hold a key no probability can beat
            init = torch.where(torch.rand(n, generator=g) < 0.33, torch.tensor(0x7fff0000), torch.tensor(5)).to(torch.int32).to(DEV)
            ref, got = init.clone(), init.clone()
            with torch.no_grad():
                logits = model._forward_impl(x, training=False)
            L.check(L.lib.vs_logits_to_volume(L.ptr(logits), classes, dmap, 1, nb, 1, direction, None, None, L.ptr(ref), None, n, L.stream_ptr()))
            model._forward_to_volume(x, dmap, 1, 1, direction, None, None, got, None, n)
            torch.cuda.synchronize()
            assert torch.equal(ref, got), (classes, direction)
            assert (got != init).any() and (got == init).any()
    finally:
        L.set_option("conv_direct_min_px", old)


def test_model_train_2d_flow_from_volumes_to_prediction(tmp_path):
    """BASELINE configs[0]'s plumbing (scripts/train_2d_model.py:40-75, predict_2d_model.py) on a synthetic volume pair: slice
    the volumes to PNGs (TrainingDataSlicer), train from the PNG directories (frozen epoch, then unfrozen), write the checkpoint,
    clean up, and predict a volume with the saved model through VolSeg2DPredictionManager."""
    from volume_segmantics_amd.data.slicers import TrainingDataSlicer
    from volume_segmantics_amd.model.operations.vol_seg_2d_trainer import VolSeg2dTrainer
    from volume_segmantics_amd.model.operations.vol_seg_prediction_manager import VolSeg2DPredictionManager
    rng = np.random.default_rng(1)
    field = rng.standard_normal((40, 40, 40)).astype(np.float32)
    for ax in range(3):
        field = (np.roll(field, 1, ax) + field + np.roll(field, -1, ax)) / 3
    data = (field * 1000 + 3000).astype(np.float32)                      # a float volume: clip_data -> uint8 on the device
    labels = (field > 0.05).astype(np.uint8) * 255                        # binary {0, 255} as in the reference's vessels labels
    settings = SimpleNamespace(starting_lr=1e-6, end_lr=50, lr_find_epochs=1, lr_reduce_factor=500, cuda_device=0, patience=3,
                               loss_criterion="BCEDiceLoss", alpha=0.75, beta=0.25, eval_metric="MeanIoU", pct_lr_inc=0.3,
                               plot_lr_graph=False, image_size=64, training_set_proportion=0.8, training_axes="All",
                               st_dev_factor=2.575, downsample=False, clip_data=True, data_hdf5_path="/data", seg_hdf5_path="/data",
                               batch_size=8, num_workers=0, precision="bf16",
                               model={"type": "U_Net", "encoder_name": "resnet34", "encoder_weights": None})
    slicer = TrainingDataSlicer(data, labels, settings)
    slicer.output_data_slices(tmp_path / "data", "data")
    slicer.output_label_slices(tmp_path / "seg", "seg")
    assert len(list((tmp_path / "data").glob("*.png"))) == 120 and slicer.num_seg_classes == 2
    trainer = VolSeg2dTrainer(tmp_path / "data", tmp_path / "seg", slicer.num_seg_classes, settings)
    out = tmp_path / "model.pytorch"
    trainer.train_model(out, 1, 3, create=True, frozen=True)
    trainer.train_model(out, 1, 3, create=False, frozen=False)
    slicer.clean_up_slices()
    assert out.exists() and len(trainer.avg_train_losses) == 2 and np.all(np.isfinite(trainer.avg_valid_losses))
    d = torch.load(out, weights_only=False)
    assert "precision" not in d["model_struc_dict"] and d["engine_settings"] == {"precision": "bf16"}
    psettings = _settings(clip_data=True, quality="medium", output_probs=False)
    mgr = VolSeg2DPredictionManager(str(out), data, psettings)
    pred = mgr.predict_volume_to_path(tmp_path / "pred.npy")
    assert pred.shape == data.shape and pred.dtype == np.uint8 and pred.max() <= 1 and mgr.predictor.model.precision == "bf16"


def test_resident_slice_feed_yields_the_data_loader_batches(tmp_path):
    """data/datasets.py: ResidentSliceLoader (PNG pairs decoded once, kept as uint8 in HBM, batches gathered on the device) against
    the DataLoader over the same dataset and the same ShardedBatchSampler: identical uint8 batches in identical order, over two
    epochs (reshuffled by set_epoch), also the last, partial validation batch."""
    from PIL import Image
    from torch.utils.data import DataLoader
    from volume_segmantics_amd.data.datasets import ResidentSliceLoader, ShardedBatchSampler, VolSeg2dDataset
    rng = np.random.default_rng(2)
    (tmp_path / "d").mkdir(); (tmp_path / "s").mkdir()
    for i in range(21):
        Image.fromarray(rng.integers(0, 255, (64, 48), dtype=np.uint8)).save(tmp_path / "d" / f"data_z_stack_{i}.png")
        Image.fromarray(rng.integers(0, 3, (64, 48), dtype=np.uint8)).save(tmp_path / "s" / f"seg_z_stack_{i}.png")
    ds = VolSeg2dDataset(tmp_path / "d", tmp_path / "s", 64, augment="device")
    for shuffle, drop_last in ((True, True), (False, False)):
        a = ShardedBatchSampler(len(ds), 4, shuffle=shuffle, drop_last=drop_last, seed=5)
        b = ShardedBatchSampler(len(ds), 4, shuffle=shuffle, drop_last=drop_last, seed=5)
        resident, loader = ResidentSliceLoader(ds, a, DEV), DataLoader(ds, batch_sampler=b)
        assert len(resident) == len(loader) == (5 if drop_last else 6)
        for epoch in range(2):
            a.set_epoch(epoch); b.set_epoch(epoch)
            for (xr, mr), (xl, ml) in zip(resident, loader):
                assert xr.is_cuda and xr.dtype == torch.uint8 and xr.shape == xl.shape and torch.equal(xr.cpu(), xl) and torch.equal(mr.cpu(), ml)
