"""CPU tests of the host-side mirror of the reference interface (no GPU compute)."""
import pickle
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import predictor_numpy as P
from volume_segmantics_amd import dist as vdist
from volume_segmantics_amd.data.losses import DiceLoss, MeanIoU
from volume_segmantics_amd.data.settings_data import get_settings_data
from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor, dirmap_of, direction_views
from volume_segmantics_amd.model.operations.vol_seg_2d_trainer import VolSeg2dTrainer
from volume_segmantics_amd.utilities import base_data_utils as utils
from volume_segmantics_amd.utilities.base_data_utils import Axis, ModelType, Quality


def test_settings_yaml_surface():
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent / "volseg-settings"
    tr = get_settings_data(root / "2d_model_train_settings.yaml")
    pr = get_settings_data(root / "2d_model_predict_settings.yaml")
    for k in ("data_im_dirname seg_im_out_dirname model_output_fn clip_data st_dev_factor data_hdf5_path seg_hdf5_path "
              "training_axes image_size downsample training_set_proportion cuda_device num_cyc_frozen num_cyc_unfrozen "
              "patience loss_criterion alpha beta eval_metric pct_lr_inc starting_lr end_lr lr_find_epochs "
              "lr_reduce_factor plot_lr_graph model").split():
        assert hasattr(tr, k), k
    assert tr.model == {"type": "U_Net", "encoder_name": "resnet34", "encoder_weights": "imagenet"}
    assert (tr.image_size, tr.num_cyc_frozen, tr.num_cyc_unfrozen, tr.patience, tr.lr_reduce_factor) == (256, 8, 5, 3, 500)
    for k in "quality output_probs clip_data st_dev_factor data_hdf5_path cuda_device downsample one_hot prediction_axis".split():
        assert hasattr(pr, k), k
    assert pr.quality == "medium" and pr.st_dev_factor == 2.575
    assert get_settings_data(None) == SimpleNamespace() and get_settings_data({"a": 1}).a == 1
    with pytest.raises(SystemExit) as e:   # reference: tests/test_settings_data.py:29-34
        get_settings_data(root / "missing.yaml")
    assert e.value.code == 1


def test_enums_and_bad_names_exit_1():
    assert [q.value for q in Quality] == [1, 3, 12] and Axis.ALL.value == 4 and ModelType.U_NET.value == 1
    assert utils.get_prediction_quality(SimpleNamespace(quality="High")) == Quality.HIGH
    assert utils.get_prediction_axis(SimpleNamespace()) == Axis.Z and utils.get_training_axis(SimpleNamespace()) == Axis.ALL
    with pytest.raises(SystemExit) as e:
        utils.get_prediction_quality(SimpleNamespace(quality="ultra"))
    assert e.value.code == 1
    # checkpoints name the reference's module path
    assert b"volume_segmantics.utilities.base_data_utils" in pickle.dumps(ModelType.U_NET)


def test_pad_crop_offsets_match_oracle_and_golden(golden):
    g = golden("g7_padded_dimension.npz")
    for d, p in zip(g["dims"], g["padded"]):
        assert utils.get_padded_dimension(int(d)) == int(p)
        if d > 0:
            padded, top, crop = utils.pad_crop_offsets(int(d))
            assert (padded, top, crop) == (int(p), P.pad_offsets(int(d))[0], P.crop_offset(int(p), int(d)))


def test_direction_maps_follow_reference_order(golden):
    g = golden("g4_direction_order.npz")
    idx = g["idxvol"]
    views = direction_views(idx, 12)
    flat = idx.ravel()
    for d, v in enumerate(views):
        m = dirmap_of(idx, v)
        s, h, w = np.meshgrid(np.arange(m.depth), np.arange(m.h), np.arange(m.w), indexing="ij")
        assert np.array_equal(flat[m.base + s * m.ss + h * m.sh + w * m.sw], g[f"dir{d:02d}"])
    assert [v.shape for v in direction_views(idx, 3)] == [(5, 6, 7), (6, 5, 7), (7, 6, 5)]


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 512, 513):
        for world in (1, 2, 3, 8):
            parts = [vdist.shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1


def test_clip_to_uint8_and_one_hot():
    rng = np.random.default_rng(0)
    v = rng.normal(100, 20, size=(6, 7, 8))
    v[0, 0, 0] = np.nan
    out = utils.clip_to_uint8(v.copy(), np.nanmean(v), 2.575)
    assert out.dtype == np.uint8 and out.min() == 0 and out.max() == 255
    lab = rng.integers(0, 3, size=(4, 5, 6))
    oh = utils.one_hot_encode_array(lab, 3)
    assert oh.shape == (3, 4, 5, 6) and np.array_equal(oh, P.one_hot_encode_array(lab, 3))


def test_losses_and_lr_finder_match_reference_goldens(golden):
    g = golden("g6_loss_metric_lr.npz")
    logits = torch.tensor(g["logits"], requires_grad=True)
    targets = torch.tensor(g["targets"])
    loss = DiceLoss(normalization="none")(logits, targets.float())
    loss.backward()
    assert np.allclose(loss.item(), g["dice_loss"], atol=1e-7) and np.allclose(logits.grad.numpy(), g["dice_grad"], rtol=1e-6, atol=1e-9)
    probs = torch.softmax(logits.detach(), 1)
    assert np.isclose(MeanIoU()(probs.unsqueeze(2), targets.unsqueeze(2)).item(), g["mean_iou"], atol=1e-7)
    assert MeanIoU()(targets.float(), targets).item() == 1.0   # reference KAT tests/test_pytorch3dunet_metrics.py:57-78
    for i in range(3):
        got = VolSeg2dTrainer._find_lr_from_graph([torch.tensor(c) for c in g[f"lr_curve{i}"]], list(g[f"lr_lrs{i}"]))
        assert np.isclose(got, float(g[f"lr_out{i}"]), rtol=1e-12)
    b = utils.prepare_training_batch([torch.zeros(3, 1, 16, 24), torch.tensor(g["mask"])], "cpu", 4)
    assert np.array_equal(b[1].numpy(), g["targets"]) and b[1].dtype == torch.uint8


def test_trainer_freeze_and_loss_selection():
    from volume_segmantics_amd.engine import VolSegUnet
    tr = VolSeg2dTrainer.__new__(VolSeg2dTrainer)
    tr.model = VolSegUnet(2)
    tr._freeze_model()
    frozen = [n for n, p in tr.model.named_parameters() if not p.requires_grad]
    assert len(frozen) == 33 and not any("downsample" in n or "bn" in n for n in frozen)
    tr._unfreeze_model()
    assert tr._count_trainable_parameters() == tr._count_parameters() == 24_430_242
    base = dict(alpha=0.75, beta=0.25, eval_metric="MeanIoU")
    for name in ("BCEDiceLoss", "DiceLoss", "BCELoss", "CrossEntropyLoss", "GeneralizedDiceLoss"):
        tr.settings = SimpleNamespace(loss_criterion=name, **base)
        assert tr._get_loss_criterion() is not None
    tr.settings = SimpleNamespace(loss_criterion="Nope", **base)
    with pytest.raises(SystemExit) as e:   # reference: tests/test_vol_seg_2d_trainer.py:62-70
        tr._get_loss_criterion()
    assert e.value.code == 1


def test_checkpoint_wire_format_roundtrip(tmp_path):
    """EarlyStopping writes the reference's dict; create_model_from_file-style loading restores it (CPU part)."""
    from volume_segmantics_amd.checkpoint_compat import reference_pickle_enum
    from volume_segmantics_amd.engine import VolSegUnet
    from volume_segmantics_amd.utilities.early_stopping import EarlyStopping
    m = VolSegUnet(3, seed=4)
    struct = {"type": reference_pickle_enum(ModelType.U_NET), "encoder_name": "resnet34", "encoder_weights": "imagenet",
              "in_channels": 1, "classes": 3}
    es = EarlyStopping(patience=2, path=tmp_path / "m.pytorch", model_dict=struct)
    es(0.5, m, torch.optim.AdamW(m.parameters()), {"a": 1})
    es(0.6, m, None, {})
    es(0.7, m, None, {})
    assert es.early_stop and es.counter == 2
    d = torch.load(tmp_path / "m.pytorch", weights_only=False)
    assert set(d) == {"model_state_dict", "model_struc_dict", "optimizer_state_dict", "loss_val", "label_codes"}
    assert d["model_struc_dict"]["type"].name == "U_NET" and d["loss_val"] == 0.5 and d["label_codes"] == {"a": 1}
    m2 = VolSegUnet(3, seed=9)
    m2.load_state_dict(d["model_state_dict"])
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    # an smp / oracle-shaped module loads the same file
    from oracle.unet_resnet34_torch import OracleUnetResnet34
    OracleUnetResnet34(1, 3).load_state_dict(d["model_state_dict"])
    # the optimiser state is in torch.optim.AdamW's own format: the reference's _load_in_weights(optimizer=True) can load it
    from volume_segmantics_amd.engine import FusedAdamW
    fo = FusedAdamW(m, lr=1e-3)
    fo.exp_avg.uniform_(-1, 1); fo.exp_avg_sq.uniform_(0, 1); fo.step_count = 7
    ref_opt = torch.optim.AdamW(OracleUnetResnet34(1, 3).parameters(), lr=1e-3)
    ref_opt.load_state_dict(fo.state_dict())
    w = "decoder.blocks.2.conv1.0.weight"
    idx = [n for n, _ in m.named_parameters()].index(w)
    st = ref_opt.state[ref_opt.param_groups[0]["params"][idx]]
    assert float(st["step"]) == 7 and torch.equal(st["exp_avg"], fo.state_dict()["state"][idx]["exp_avg"])
    assert st["exp_avg"].shape == dict(m.named_parameters())[w].shape
    fo2 = FusedAdamW(VolSegUnet(3, seed=1), lr=1.0)
    fo2.load_state_dict(ref_opt.state_dict())
    assert fo2.step_count == 7 and torch.equal(fo2.exp_avg, fo.exp_avg) and torch.equal(fo2.exp_avg_sq, fo.exp_avg_sq)
    assert fo2.param_groups[0]["lr"] == 1e-3
    # keys only this engine knows stay out of model_struc_dict (the reference does smp.Unet(**model_struc_dict))
    es2 = EarlyStopping(patience=2, path=tmp_path / "m2.pytorch", model_dict=dict(struct, precision="bf16"))
    es2(0.5, m, fo, {})
    d2 = torch.load(tmp_path / "m2.pytorch", weights_only=False)
    assert "precision" not in d2["model_struc_dict"] and d2["engine_settings"] == {"precision": "bf16"}


def test_reference_enum_lookup_with_the_reference_installed_but_not_imported(tmp_path, monkeypatch):
    """Side-by-side install: ``volume_segmantics`` is importable but nobody has imported it yet - reference_pickle_enum must
    import it (not expect it in sys.modules), and fall back to the aliases when its import fails for missing dependencies."""
    import sys
    from volume_segmantics_amd import checkpoint_compat as cc
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "volume_segmantics" or k.startswith("volume_segmantics.")}
    try:
        pkg = tmp_path / "volume_segmantics" / "utilities"
        pkg.mkdir(parents=True)
        (tmp_path / "volume_segmantics" / "__init__.py").write_text("")
        (pkg / "__init__.py").write_text("")
        (pkg / "base_data_utils.py").write_text("from enum import Enum\nclass ModelType(Enum):\n    U_NET = 1\n    FPN = 3\n")
        monkeypatch.syspath_prepend(str(tmp_path))
        import importlib
        importlib.invalidate_caches()
        e = cc.reference_pickle_enum(ModelType.U_NET)
        assert type(e).__module__ == "volume_segmantics.utilities.base_data_utils" and e.name == "U_NET"
        assert "stub" not in repr(sys.modules["volume_segmantics"]) and sys.modules["volume_segmantics"].__file__.startswith(str(tmp_path))
        # the same package, but its module fails to import (a missing third-party dependency): aliases take over
        for k in [k for k in sys.modules if k == "volume_segmantics" or k.startswith("volume_segmantics.")]:
            del sys.modules[k]
        (pkg / "base_data_utils.py").write_text("import a_module_that_is_not_installed\n")
        importlib.invalidate_caches()
        e = cc.reference_pickle_enum(ModelType.FPN)
        assert e.name == "FPN"
    finally:
        for k in [k for k in sys.modules if k == "volume_segmantics" or k.startswith("volume_segmantics.")]:
            del sys.modules[k]
        sys.modules.update(saved)
        from volume_segmantics_amd.utilities import base_data_utils as ours
        for name in ("ModelType", "Quality", "Axis"):   # the alias registration renames the enums' module; keep that consistent
            if saved:
                getattr(ours, name).__module__ = "volume_segmantics.utilities.base_data_utils"


def test_predictor_orchestration_with_cpu_standin_matches_reference_goldens(golden):
    """Single process: the predictor's direction / index-map / key-merge logic, computed by the test-only oracle
    backend, reproduces the goldens generated by the reference's own predictor bit for bit."""
    from conftest import fingerprint
    from cpu_backend import OracleBackend
    from oracle.unet_resnet34_torch import seeded_oracle
    g = golden("g3_predict_29x64x40_c4.npz")
    net = seeded_oracle(4, 0).eval()
    if not np.array_equal(fingerprint(net), g["fingerprint"]):
        pytest.skip("torch RNG stream differs from the build container")
    pred = VolSeg2dPredictor.__new__(VolSeg2dPredictor)
    pred.model, pred.num_labels, pred.settings = net, 4, SimpleNamespace(prediction_batch_size=4, cuda_device=0)
    pred.backend_factory = OracleBackend
    vol = g["vol"]
    for ax, name in ((Axis.Z, "z"), (Axis.Y, "y"), (Axis.X, "x")):
        l, p = pred._predict_single_axis(vol, axis=ax)
        assert np.array_equal(l, g[f"single_{name}_labels"]) and np.array_equal(p.view(np.uint16), g[f"single_{name}_probs"].view(np.uint16))
    l, p = pred._predict_3_ways_max_probs(vol)
    assert np.array_equal(l, g["three_labels"]) and np.array_equal(p, g["three_probs"])
    assert np.array_equal(pred._predict_3_ways_one_hot(vol), g["onehot_three"])
    assert np.array_equal(pred._predict_single_axis_to_one_hot(vol), g["onehot_z"])


def test_training_data_slicer_follows_the_reference_tests(tmp_path):
    """volume_segmantics/data/slicers.py through the reference's own test expectations (tests/test_slicers.py:33-160): label
    classes made sequential from zero, one PNG per slice of every requested axis named <prefix>_<axis>_stack_<i>.png, binary
    labels {0, 255} written as {0, 1}, clean-up removes the directories - plus known answers of skimage.img_as_ubyte (generated
    with scikit-image 0.18.3's own function) for the restated conversion."""
    from PIL import Image
    from volume_segmantics_amd.data.slicers import TrainingDataSlicer, img_as_ubyte
    settings = SimpleNamespace(st_dev_factor=2.575, downsample=False, clip_data=True, data_hdf5_path="/data", seg_hdf5_path="/data",
                               training_axes="All", device_preprocess=False)
    rng = np.random.default_rng(0)
    vol = rng.integers(0, 256, (9, 14, 11))
    labels = rng.integers(1, 5, (9, 14, 11)).astype(np.uint8)           # no zeros: 1..4 -> 0..3
    s = TrainingDataSlicer(vol, labels.copy(), settings)
    assert s.data_vol.dtype == np.uint8 and list(np.unique(s.seg_vol)) == [0, 1, 2, 3] and s.multilabel
    assert s.codes == ["label_val_1", "label_val_2", "label_val_3", "label_val_4"]
    s.output_data_slices(tmp_path / "im", "data")
    s.output_label_slices(tmp_path / "seg", "seg")
    assert len(list((tmp_path / "im").glob("*.png"))) == sum(vol.shape) == len(list((tmp_path / "seg").glob("*.png")))
    assert np.array_equal(np.array(Image.open(tmp_path / "seg" / "seg_y_stack_3.png")), s.seg_vol[:, 3])
    assert np.array_equal(np.array(Image.open(tmp_path / "im" / "data_x_stack_10.png")), s.data_vol[:, :, 10])
    s.clean_up_slices()
    assert not (tmp_path / "im").exists() and not (tmp_path / "seg").exists()
    settings.training_axes = "x"
    binary = (rng.integers(0, 2, (9, 14, 11)) * 255).astype(np.uint8)   # {0, 255}: two classes -> relabelled {0, 1}
    b = TrainingDataSlicer(vol, binary, settings)
    b.output_label_slices(tmp_path / "bin", "seg")
    files = list((tmp_path / "bin").glob("*.png"))
    assert len(files) == 11 and all(np.array_equal(np.unique(np.array(Image.open(f))), [0, 1]) for f in files)
    # skimage.img_as_ubyte known answers (scikit-image 0.18.3): integers are rescaled by bit depth
    src = [0, 1, 2, 3, 100, 255, 1000, 30000]
    assert img_as_ubyte(np.array(src, np.int64)).tolist() == [0] * 8 == img_as_ubyte(np.array(src, np.uint32)).tolist()
    assert img_as_ubyte(np.array(src, np.int16)).tolist() == [0, 0, 0, 0, 0, 1, 7, 234]
    assert img_as_ubyte(np.array(src, np.uint16)).tolist() == [0, 0, 0, 0, 0, 0, 3, 117]
    assert img_as_ubyte(np.array([0, 1, 2, 3, 100, -1, -24, 48], np.int8)).tolist() == [0, 2, 4, 6, 201, 0, 0, 96]
    assert img_as_ubyte(np.array([-1.0, -0.5, 0.0, 0.25, 0.5, 0.999, 1.0])).tolist() == [0, 0, 0, 64, 128, 255, 255]
    assert img_as_ubyte(np.array([True, False])).tolist() == [255, 0]
    with pytest.raises(ValueError, match="between -1 and 1"):
        img_as_ubyte(np.array([1.5]))


def test_fit_to_square_downscales_without_antialiasing():
    """LongestMaxSize + PadIfNeeded of the training dataset (reference data/augmentations.py:12-27 -> albumentations:
    cv2.INTER_LINEAR for the image, INTER_NEAREST for the mask): a 2x down-scale of an 8-pixel checkerboard row samples at
    (dst + 0.5) * 2 - 0.5 - the mean of TWO neighbours (a box-antialiased resize would average more and flatten it to grey
    everywhere the same way), masks stay label-valued, and the short side is reflect-101 padded."""
    import numpy as np
    from volume_segmantics_amd.data.datasets import fit_to_square
    img = np.zeros((4, 8), np.uint8)
    img[:, 0::4] = 200; img[:, 1::4] = 200          # columns 200 200 0 0 200 200 0 0
    mask = np.zeros((4, 8), np.uint8); mask[:, 4:] = 3
    out, m = fit_to_square(img, mask, 4)
    assert out.shape == (4, 4) and m.shape == (4, 4)
    # rows: 2 real rows (4 * 0.5) centred, one reflected row above and below
    assert np.array_equal(out[1], out[2]) and np.array_equal(out[0], out[1]) and np.array_equal(out[3], out[2])
    assert out[1].tolist() == [200, 0, 200, 0]      # x = 0.5, 2.5, 4.5, 6.5: the mean of columns (0,1), (2,3), (4,5), (6,7)
    assert set(np.unique(m).tolist()) <= {0, 3} and m[1].tolist() == [0, 0, 3, 3]
    same, _ = fit_to_square(np.arange(16, dtype=np.uint8).reshape(4, 4), None, 4)
    assert np.array_equal(same, np.arange(16, dtype=np.uint8).reshape(4, 4))


@pytest.mark.parametrize("encoder", ["resnet34", "resnet50", "efficientnet-b3", "timm-resnest50d"])
def test_pretrained_encoder_loader_patches_the_first_convolution(encoder):
    """smp's patch_first_conv (in_channels=1: the pretrained 3-channel kernel summed over its input channels) applies to
    whichever tensor is the encoder's first convolution: conv1.weight (torchvision ResNets), _conv_stem.weight
    (efficientnet-pytorch), conv1.0.weight (timm ResNeSt's deep stem) - model_2d.py:15-16 of the reference via
    smp.Unet(encoder_weights="imagenet", in_channels=1)."""
    from volume_segmantics_amd.engine import VolSegUnet
    from volume_segmantics_amd.model.model_2d import load_pretrained_encoder
    donor = VolSegUnet(2, seed=11, encoder=encoder)
    own = donor.state_dict()
    firsts = [k for k, v in own.items() if k.startswith("encoder.") and v.ndim == 4 and v.shape[1] == 1 and v.shape[2] > 1]
    first = {"resnet34": "encoder.conv1.weight", "resnet50": "encoder.conv1.weight", "efficientnet-b3": "encoder._conv_stem.weight",
             "timm-resnest50d": "encoder.conv1.0.weight"}[encoder]
    assert first in firsts
    g = torch.Generator().manual_seed(3)
    upstream = {k[len("encoder."):]: v.clone() for k, v in own.items() if k.startswith("encoder.")}
    w3 = torch.randn(own[first].shape[0], 3, *own[first].shape[2:], generator=g)
    upstream[first[len("encoder."):]] = w3
    upstream["fc.weight"] = torch.zeros(10, 7)                 # a classifier head: ignored
    model = VolSegUnet(2, seed=5, encoder=encoder)
    load_pretrained_encoder(model, upstream)
    got = model.state_dict()
    assert torch.allclose(got[first], w3.sum(1, keepdim=True))
    for k, v in own.items():
        if k.startswith("encoder.") and k != first:
            assert torch.equal(got[k], v), k
    with pytest.raises(ValueError):
        load_pretrained_encoder(model, {"conv_that_is_not_there.weight": torch.zeros(1)})


def test_fused_adamw_state_dict_has_no_entries_for_frozen_parameters():
    """torch.optim.AdamW (the reference's optimiser) keeps no state for a parameter that never received a gradient: the
    frozen encoder convolutions of the first phase (vol_seg_2d_trainer.py:102-108) must not appear with `step` = N and
    empty moments, or a reference-side resume would bias-correct them as if they had been updated N times."""
    from volume_segmantics_amd.engine import FusedAdamW, VolSegUnet
    from oracle.unet_resnet34_torch import OracleUnetResnet34
    m = VolSegUnet(2, seed=4)
    names = [n for n, _ in m.named_parameters()]
    for n, p in m.named_parameters():
        if "encoder" in n and "conv" in n:
            p.requires_grad = False
    fo = FusedAdamW(m, lr=1e-3)
    assert fo.state_dict()["state"] == {}                     # nothing stepped yet: empty, like a fresh torch optimiser
    fo.exp_avg.uniform_(-1, 1); fo.exp_avg_sq.uniform_(0, 1); fo.step_count = 5
    sd = fo.state_dict()
    frozen = {i for i, n in enumerate(names) if "encoder" in n and "conv" in n}
    assert frozen and not (frozen & set(sd["state"])) and set(sd["state"]) == set(range(len(names))) - frozen
    assert sd["param_groups"][0]["params"] == list(range(len(names)))
    ref = torch.optim.AdamW(OracleUnetResnet34(1, 2).parameters(), lr=1e-3)
    ref.load_state_dict(sd)                                    # the reference's _load_in_weights(optimizer=True) path
    assert len(ref.state) == len(names) - len(frozen)


def test_get_batch_size_follows_the_reference_memory_rule_with_two_documented_changes(monkeypatch):
    """reference utilities/base_data_utils.py:104-122: < 8 GB free -> 2, else 12 (training) / 4 (prediction).  This engine
    keeps the training rule (train-mode BN statistics depend on the batch), lets an explicit settings key win, and
    predicts with HIP_PRED_BATCH slices per forward (eval-mode results are batch independent - see
    tests/test_hip_predictor.py::test_prediction_is_independent_of_the_batch_size); `prediction_batch_size: 4`
    reproduces the reference's number."""
    from volume_segmantics_amd.utilities import config as cfg
    props = SimpleNamespace(total_memory=288 * 1024 ** 3)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda dev: props)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda dev: 0)
    s = SimpleNamespace(cuda_device=0)
    assert utils.get_batch_size(s) == cfg.BIG_CUDA_TRAIN_BATCH == 12
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda dev: 283 * 1024 ** 3)       # 5 GB free
    assert utils.get_batch_size(s) == cfg.SMALL_CUDA_BATCH == 2
    assert utils.get_batch_size(s, prediction=True) == cfg.HIP_PRED_BATCH == 32
    assert utils.get_batch_size(SimpleNamespace(cuda_device=0, prediction_batch_size=cfg.BIG_CUDA_PRED_BATCH),
                                prediction=True) == 4
    assert utils.get_batch_size(SimpleNamespace(cuda_device=0, batch_size=32)) == 32


def test_hdf5_volumes_round_trip_through_the_c_library_without_h5py(tmp_path):
    """utilities/base_data_utils.py:179-214, 351-356 do their HDF5 I/O with h5py; this image has libhdf5 but no h5py for the torch
    interpreter, so `utilities/hdf5_lite.py` binds the two calls to the C library.  Round trips for every dtype the path meets
    (uint8 volumes and labels, uint16 / float32 tomograms, float16 probabilities), nested internal paths, the NeXus candidates,
    chunked + gzip layout like `create_dataset(..., chunks=True, compression="gzip")` - and the files are read back by an
    INDEPENDENT program where the image has one (conda's `h5dump`): datatype, dataspace and values."""
    import shutil
    import subprocess
    from pathlib import Path
    from volume_segmantics_amd.utilities import hdf5_lite
    if utils._h5py() is None and not hdf5_lite.available():
        pytest.skip("neither h5py nor libhdf5 on this machine")
    rng = np.random.default_rng(3)
    for dt in ("uint8", "uint16", "int16", "int32", "float32", "float64", "float16"):
        a = (rng.random((7, 33, 18)) * 200 - 40).astype(dt)
        path = tmp_path / f"vol_{dt}.h5"
        utils.save_data_to_hdf5(a, path, internal_path="/data", chunking=True)
        assert hdf5_lite.is_hdf5(path)
        b, chunks = utils.get_numpy_from_path(path, internal_path="/data")
        assert b.dtype == a.dtype and np.array_equal(a, b) and chunks is not None and len(chunks) == 3
    big = rng.integers(0, 255, (96, 128, 128), dtype=np.uint8)           # 1.5 MiB: several chunks, compressed
    utils.save_data_to_hdf5(big, tmp_path / "big.h5")
    back, chunks = utils.numpy_from_hdf5(tmp_path / "big.h5")
    assert np.array_equal(big, back) and np.prod(chunks) <= 512 * 1024 and (tmp_path / "big.h5").stat().st_size < big.size * 1.1
    nested = np.arange(2 * 3 * 4, dtype=np.uint16).reshape(2, 3, 4)
    if utils._h5py() is None:
        hdf5_lite.write_dataset(tmp_path / "scan.nxs", "/entry/final_result_tomo/data", nested)
        got, _ = utils.get_numpy_from_path(tmp_path / "scan.nxs")                     # NeXus: the reference's two candidate paths
        assert np.array_equal(got, nested)
        assert hdf5_lite.exists(tmp_path / "scan.nxs", "/entry/final_result_tomo") and not hdf5_lite.exists(tmp_path / "scan.nxs", "/processed/result/data")
        with pytest.raises(KeyError):
            hdf5_lite.read_dataset(tmp_path / "scan.nxs", "/data")
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if Path("/opt/conda/bin/h5dump").exists() else None)
    if h5dump:
        head = subprocess.run([h5dump, "-H", str(tmp_path / "vol_float16.h5")], capture_output=True, text=True).stdout
        assert "16-bit little-endian floating-point" in head and "( 7, 33, 18 )" in head, head
        head = subprocess.run([h5dump, "-H", "-p", str(tmp_path / "big.h5")], capture_output=True, text=True).stdout
        assert "H5T_STD_U8LE" in head and "CHUNKED" in head and "DEFLATE" in head, head
        small = np.arange(12, dtype=np.int32).reshape(1, 3, 4)
        utils.save_data_to_hdf5(small, tmp_path / "small.h5", internal_path="/seg/labels")
        body = subprocess.run([h5dump, "-d", "/seg/labels", str(tmp_path / "small.h5")], capture_output=True, text=True).stdout
        assert "(0,2,0): 8, 9, 10, 11" in body, body


def test_normalise_on_load_plan_is_the_one_reader_rule():
    """`vs_unet_nl_plan` (host logic, nothing is launched): with `nl_fwd` on, a bf16 training forward leaves exactly those conv +
    BN + ReLU units without a normalisation sweep whose output has ONE reader, a stride-1 3x3 convolution on the register-staged kernels - for the
    headline network at batch 32 of 256 x 256 the 16 BasicBlock conv1 units and 7 decoder convolutions (the last three decoder
    outputs feed the strip kernels / the head: they keep their sweep), never a unit with a residual input, the stem, a downsample
    1x1 or a unit whose output is a skip connection; none at all in fp32, or with the option off.  By default (`nl_max_c` 64) only
    the six units of up to 64 channels among them."""
    import ctypes as C

    from volume_segmantics_amd import _lib as L

    def plan(dtype, topology_encoder=34, n=32, hw=256):
        h = C.c_void_p()
        L.check(L.lib.vs_unet_create_ex(C.byref(h), dtype, 2, n, hw, hw, topology_encoder))
        try:
            nu = L.lib.vs_unet_num_units(h)
            flags = (C.c_int * nu)()
            assert L.lib.vs_unet_nl_plan(h, n, flags, nu) == nu
            return {name.split(" [")[0]: flags[i] for i, name in enumerate(L.unit_names(h))}
        finally:
            L.lib.vs_unet_destroy(h)

    old = L.lib.vs_get_option(b"nl_fwd"), L.lib.vs_get_option(b"nl_max_c")
    assert old == (1, 64)       # the defaults: on, for producers of up to 64 channels (where it measures faster - DESIGN.md section 5)
    try:
        L.set_option("nl_fwd", 0)
        assert sum(plan(L.VS_BF16).values()) == 0
        L.set_option("nl_fwd", 1)
        small = sorted(k for k, v in plan(L.VS_BF16).items() if v)
        assert small == sorted(["encoder.layer1.0.conv1.weight", "encoder.layer1.1.conv1.weight", "encoder.layer1.2.conv1.weight",
                                "decoder.blocks.2.conv1.0.weight", "decoder.blocks.2.conv2.0.weight", "decoder.blocks.3.conv1.0.weight"]), small
        L.set_option("nl_max_c", 512)
        on = plan(L.VS_BF16)
        assert sum(plan(L.VS_F32).values()) == 0
    finally:
        L.set_option("nl_fwd", old[0])
        L.set_option("nl_max_c", old[1])
    chosen = sorted(k for k, v in on.items() if v)
    assert len(chosen) == 23, chosen
    assert sum(".conv1.weight" in k and k.startswith("encoder.layer") for k in chosen) == 16
    assert all("conv2.weight" not in k and "downsample" not in k and k != "encoder.conv1.weight" for k in chosen if k.startswith("encoder"))
    dec = [k for k in chosen if k.startswith("decoder")]
    assert dec == sorted(["decoder.blocks.0.conv1.0.weight", "decoder.blocks.0.conv2.0.weight", "decoder.blocks.1.conv1.0.weight",
                          "decoder.blocks.1.conv2.0.weight", "decoder.blocks.2.conv1.0.weight", "decoder.blocks.2.conv2.0.weight",
                          "decoder.blocks.3.conv1.0.weight"]), dec


def test_sync_batchnorm_hook_is_accepted_only_where_it_is_built():
    """vs_unet_set_stats_hook (host logic): bf16 plans whose BatchNorms all sit behind bias-free convolutions / the ResNet stem take
    the cross-rank statistics hook (U-Net, U-Net++ / ResNet-50 = BASELINE configs[3], FPN); fp32 plans and networks with BatchNorm
    elsewhere (Linknet's transposed convolutions, EfficientNet's standalone norms) are refused with a message, not silently run
    with per-rank statistics."""
    import ctypes as C

    from volume_segmantics_amd import _lib as L

    cb = L.STATS_HOOK(lambda user, values, count, kind, stream: 0)

    def accepts(dtype, code):
        h = C.c_void_p()
        L.check(L.lib.vs_unet_create_ex(C.byref(h), dtype, 2, 2, 64, 64, code))
        try:
            rc = L.lib.vs_unet_set_stats_hook(h, C.cast(cb, C.c_void_p), None, 2)
            if rc == 0:
                assert L.lib.vs_unet_set_stats_hook(h, None, None, 1) == 0
            return rc == 0, L.last_error()
        finally:
            L.lib.vs_unet_destroy(h)

    assert accepts(L.VS_BF16, 34)[0] and accepts(L.VS_BF16, 1050)[0] and accepts(L.VS_BF16, 3034)[0]
    ok, msg = accepts(L.VS_F32, 34)
    assert not ok and "bf16" in msg
    for code in (2034, 103):          # Linknet / resnet34, U-Net / efficientnet-b3
        ok, msg = accepts(L.VS_BF16, code)
        assert not ok and "SyncBatchNorm is not built" in msg, (code, msg)


def test_repeated_directions_are_exact_repeats_for_every_shape():
    """Four of the reference's twelve prediction directions (vol_seg_2d_predictor.py:100-116: np.rot90 in the (0, 1) plane, then Z / Y / X
    stacks) hold the SAME slices at the SAME voxel addresses as an earlier direction, in reversed order - the table the predictor
    skips by in the max-probability merge (first-wins on ties, :90-98).  Address-valued volumes make the check exact; no other
    pair of directions coincides."""
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import REPEATED_DIRECTIONS
    for shape in ((5, 5, 5), (4, 6, 7), (29, 64, 40), (3, 3, 8)):
        vol = np.arange(int(np.prod(shape))).reshape(shape)
        views = direction_views(vol, 12)
        found = {}
        for a in range(12):
            for b in range(a):
                if views[a].shape == views[b].shape and (np.array_equal(views[a], views[b]) or np.array_equal(views[a], views[b][::-1])):
                    found[a] = b
        assert found == REPEATED_DIRECTIONS, (shape, found)
        for later, earlier in REPEATED_DIRECTIONS.items():
            assert later > earlier and np.array_equal(views[later], views[earlier][::-1])


def test_resident_feed_keeps_the_reference_label_range_error(tmp_path):
    """utilities/base_data_utils.py:150-158 of the reference: F.one_hot raises "Class values must be smaller than num_classes" on the
    first batch whose mask holds a value >= the label count (the classic 0 / 255 PNG with 2 labels).  The resident feed's masks
    already live on the device, where prepare_training_batch's host check does not look: the loader checks its whole subset once
    and raises the same error when iterated; in-range masks pass and the batches equal the DataLoader's."""
    from PIL import Image
    from torch.utils.data import DataLoader
    from volume_segmantics_amd.data.datasets import ResidentSliceLoader, ShardedBatchSampler, VolSeg2dDataset
    rng = np.random.default_rng(3)
    for sub, top in (("bad", 255), ("ok", 1)):
        (tmp_path / sub / "d").mkdir(parents=True); (tmp_path / sub / "s").mkdir()
        for i in range(9):
            Image.fromarray(rng.integers(0, 255, (32, 32), dtype=np.uint8)).save(tmp_path / sub / "d" / f"data_z_stack_{i}.png")
            Image.fromarray((rng.integers(0, 2, (32, 32)) * top).astype(np.uint8)).save(tmp_path / sub / "s" / f"seg_z_stack_{i}.png")
    for sub in ("bad", "ok"):
        ds = VolSeg2dDataset(tmp_path / sub / "d", tmp_path / sub / "s", 32, augment="device")
        loader = ResidentSliceLoader(ds, ShardedBatchSampler(len(ds), 4, shuffle=False, drop_last=False), "cpu")
        assert loader.max_label == (255 if sub == "bad" else 1)
        assert len(list(loader)) == 3          # no label count known yet: nothing to check against
        loader.num_labels = 2                  # what VolSeg2dTrainer.__init__ sets
        if sub == "bad":
            with pytest.raises(RuntimeError, match="Class values must be smaller than num_classes"):
                next(iter(loader))
        else:
            ref = DataLoader(ds, batch_sampler=ShardedBatchSampler(len(ds), 4, shuffle=False, drop_last=False))
            for (xr, mr), (xl, ml) in zip(loader, ref):
                assert torch.equal(xr, xl) and torch.equal(mr, ml)
    assert ResidentSliceLoader.bytes_needed(768, 256) == 768 * 256 * 256 * 2


def test_global_dice_is_a_training_construct_only():
    """`sync_batchnorm` pairs SyncBatchNorm with the Dice of the GLOBAL batch (HipDiceLoss(global_group=...): one all-reduce inside the
    loss).  In the validation loop the ranks run different numbers of batches when the last global batch is partial (a rank whose
    share is empty skips it, data/datasets.py: ShardedBatchSampler), so the criterion must not communicate there: under
    torch.no_grad() - and for logits that carry no gradient - the group is dropped."""
    from types import SimpleNamespace as NS
    from volume_segmantics_amd.data import losses
    seen = []
    orig = losses._FusedDiceFn.apply
    losses._FusedDiceFn.apply = staticmethod(lambda i, t, e, g=None: seen.append(g) or 0.0)
    try:
        crit = losses.HipDiceLoss(global_group="WORLD")
        fake = lambda rg: NS(is_cuda=True, dtype=torch.float32, dim=lambda: 4, shape=(2, 2, 8, 8), requires_grad=rg)
        crit(fake(True), fake(False))
        with torch.no_grad():
            crit(fake(True), fake(False))
        crit(fake(False), fake(False))
    finally:
        losses._FusedDiceFn.apply = orig
    assert seen == ["WORLD", None, None]
