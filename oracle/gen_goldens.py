"""Generate tests/golden/*.npz by running the REFERENCE's own code (build container only).

Run here, never on the GPU box:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_goldens.py

The reference (/root/reference, read-only) is imported with ``sys.modules`` stubs for
its absent non-arithmetic dependencies; every failure that made
the stubs necessary was an ordinary ModuleNotFoundError, no permission was denied.
What is real reference code in these goldens:
  * VolSeg2dPredictor._predict_single_axis / _predict_3_ways_max_probs /
    _predict_12_ways_max_probs / _merge_vols_in_mem / *_one_hot
    (volume_segmantics/model/operations/vol_seg_2d_predictor.py:31-136)
  * VolSeg2dPredictionDataset + get_2d_prediction_dataloader (data/datasets.py:89-181,
    data/dataloaders.py:60-71), get_padded_dimension (data/augmentations.py:30-43)
  * crop_tensor_to_array / rotate_array_to_axis / one_hot_encode_array /
    prepare_training_batch (utilities/base_data_utils.py:125-158)
  * DiceLoss(normalization="none"), MeanIoU (data/pytorch3dunet_{losses,metrics}.py)
  * VolSeg2dTrainer._train_one_batch / _freeze_model / _find_lr_from_graph /
    _lr_exp_stepper (model/operations/vol_seg_2d_trainer.py)
  * BaseDataManager.__init__ / _preprocess_data (data/base_data_manager.py:10-42) with
    clip_to_uint8 (utilities/base_data_utils.py:243-287): g8 (`--only-g8` regenerates just that file)
What is restated (third-party code that is not installed; [3p-memory] in SURVEY.md):
  * the network itself (oracle/unet_resnet34_torch.py, injected as predictor.model the
    way _get_model_from_trainer does, vol_seg_2d_predictor.py:28-29)
  * albumentations.PadIfNeeded (centre, BORDER_REFLECT_101) and ToTensorV2,
    torchvision.transforms.functional.center_crop - pure data movement.
Nothing of the reference (source or bytecode) is written into this repository: the
outputs are data only (inputs + expected outputs).
"""
from __future__ import annotations

import os
import sys
import types
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
OUT = REPO / "tests" / "golden"
sys.dont_write_bytecode = True
sys.path.insert(0, str(REPO))


# ------------------------------------------------------------------------------------------
# stubs for absent third-party modules
# ------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    class Compose:
        def __init__(self, transforms, p=1.0):
            self.transforms = transforms

        def __call__(self, **data):
            for t in self.transforms:
                data = t(**data)
            return data

    class PadIfNeeded:
        """albumentations ^1.1 PadIfNeeded, position=center, border_mode=BORDER_REFLECT_101."""

        def __init__(self, min_height, min_width, p=1.0):
            self.min_height, self.min_width = min_height, min_width

        def __call__(self, **data):
            img = data["image"]
            rows, cols = img.shape[:2]
            if rows < self.min_height:
                top = int((self.min_height - rows) / 2.0)
                bottom = self.min_height - rows - top
            else:
                top = bottom = 0
            if cols < self.min_width:
                left = int((self.min_width - cols) / 2.0)
                right = self.min_width - cols - left
            else:
                left = right = 0
            if top or bottom or left or right:
                img = np.pad(img, ((top, bottom), (left, right)), mode="reflect")
            out = dict(data)
            out["image"] = img
            return out

    class ToTensorV2:
        def __call__(self, **data):
            out = dict(data)
            img = data["image"]
            if img.ndim == 2:
                img = np.expand_dims(img, 2)
            out["image"] = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))
            if "mask" in data and data["mask"] is not None:
                out["mask"] = torch.from_numpy(data["mask"])
            return out

    class _Unused:
        def __init__(self, *a, **k):
            raise RuntimeError("stub: not available in the build container")

    comp = _mod("albumentations.core.composition", Compose=Compose)
    core = _mod("albumentations.core", composition=comp)
    A = _mod("albumentations", Compose=Compose, PadIfNeeded=PadIfNeeded, core=core)
    for n in ("LongestMaxSize", "RandomSizedCrop", "VerticalFlip", "RandomRotate90", "Transpose", "OneOf",
              "ElasticTransform", "GridDistortion", "OpticalDistortion", "CLAHE", "RandomBrightnessContrast",
              "RandomGamma"):
        setattr(A, n, _Unused)
    tr = _mod("albumentations.pytorch.transforms", ToTensorV2=ToTensorV2)
    _mod("albumentations.pytorch", transforms=tr, ToTensorV2=ToTensorV2)
    _mod("cv2", IMREAD_GRAYSCALE=0, BORDER_REFLECT_101=4)
    _mod("h5py")
    _mod("imageio")
    _mod("termplotlib")
    _mod("segmentation_models_pytorch")
    sk = _mod("skimage", img_as_ubyte=None)
    sk.io = _mod("skimage.io")
    sk.measure = _mod("skimage.measure", block_reduce=None)
    sk.metrics = _mod("skimage.metrics", peak_signal_noise_ratio=None, mean_squared_error=None)
    sk.exposure = _mod("skimage.exposure")

    def center_crop(img, output_size):
        """torchvision.transforms.functional.center_crop for tensors, no-padding case."""
        ch, cw = int(output_size[0]), int(output_size[1])
        h, w = img.shape[-2:]
        top = int(round((h - ch) / 2.0))
        left = int(round((w - cw) / 2.0))
        return img[..., top:top + ch, left:left + cw]

    fn = _mod("torchvision.transforms.functional", center_crop=center_crop)
    tf = _mod("torchvision.transforms", functional=fn)
    _mod("torchvision", transforms=tf)


def weight_fingerprint(net) -> np.ndarray:
    sd = net.state_dict()
    tot = sum(float(v.double().sum()) for k, v in sd.items() if v.dtype.is_floating_point)
    sq = sum(float((v.double() ** 2).sum()) for k, v in sd.items() if v.dtype.is_floating_point)
    return np.array([tot, sq, float(sd["segmentation_head.0.weight"].flatten()[3]),
                     float(sd["encoder.layer3.2.conv1.weight"].flatten()[1234])], dtype=np.float64)


def synth_volume(shape, seed):
    """Smooth-ish uint8 test volume (cheap separable box blur of gaussian noise)."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(shape).astype(np.float32)
    for ax in range(3):
        v = (np.roll(v, 1, ax) + v + np.roll(v, -1, ax)) / 3.0
    v = (v - v.mean()) / v.std()
    return np.clip(128 + 40 * v, 0, 255).astype(np.uint8)


def gen_g8():
    """G8: BaseDataManager._preprocess_data + clip_to_uint8 (data/base_data_manager.py:10-42, utilities/base_data_utils.py:243-287)
    run by the reference itself on small volumes of every input type: inputs, the mean it stores, the uint8 volume it keeps."""
    from volume_segmantics.data.base_data_manager import BaseDataManager

    rng = np.random.default_rng(88)
    f32 = (rng.standard_normal((13, 37, 65)) * 900 + 4000).astype(np.float32)
    f32.reshape(-1)[rng.choice(f32.size, 40, replace=False)] = np.nan
    f32[0, 0, :3] = [3.0e4, -2.5e4, 2.0e4]                              # far outliers (infinities make the reference's result undefined)
    f64 = rng.standard_normal((7, 40, 33)) * 3.5e-3 + 0.02
    f64.reshape(-1)[rng.choice(f64.size, 9, replace=False)] = np.nan
    u16 = np.clip(rng.gamma(2.0, 6000.0, (20, 64, 70)), 0, 65535).astype(np.uint16)
    i16 = np.clip(rng.standard_normal((9, 50, 41)) * 5000 - 300, -32768, 32767).astype(np.int16)
    u8 = rng.integers(0, 256, (16, 33, 47)).astype(np.uint8)
    f32b = (rng.random((3, 5, 7)) * 10).astype(np.float32)             # smaller than one summation block
    out = {}
    for name, vol, factor in (("f32", f32, 2.575), ("f64", f64, 2.575), ("u16", u16, 2.0), ("i16", i16, 3.1), ("u8", u8, 1.5),
                              ("f32b", f32b, 0.7)):
        settings = SimpleNamespace(st_dev_factor=factor, downsample=False, clip_data=True, data_hdf5_path="/data")
        dm = BaseDataManager(vol.copy(), settings)
        assert dm.data_vol.dtype == np.uint8 and dm.data_vol.shape == vol.shape
        out[name + "__in"] = vol
        out[name + "__factor"] = np.float64(factor)
        out[name + "__mean"] = np.asarray(dm.data_mean)                # keeps the scalar's dtype
        out[name + "__out"] = dm.data_vol
        print("g8", name, vol.dtype, "mean", repr(dm.data_mean), "hist head", np.bincount(dm.data_vol.reshape(-1), minlength=256)[[0, 1, 127, 254, 255]])
    np.savez_compressed(OUT / "g8_clip_to_uint8.npz", **out)


def gen_g9():
    """G9: the network input the reference builds from prediction volumes that are NOT uint8 (reachable with clip_data: False):
    get_2d_prediction_dataset -> VolSeg2dPredictionDataset.__getitem__ (data/datasets.py:120-142,175-181) run by the reference
    itself: integer types -> float32 / 255 whatever their range, float types without the / 255.  Shapes that need padding."""
    from volume_segmantics.data.datasets import get_2d_prediction_dataset
    rng = np.random.default_rng(99)
    shape = (3, 37, 45)
    vols = {
        "u8": rng.integers(0, 256, shape).astype(np.uint8),
        "u16": rng.integers(0, 65536, shape).astype(np.uint16),
        "i16": rng.integers(-32768, 32768, shape).astype(np.int16),
        "i32": rng.integers(-2 ** 31, 2 ** 31 - 1, shape).astype(np.int32),          # beyond float32's 24-bit integers
        "u32": rng.integers(0, 2 ** 32 - 1, shape).astype(np.uint32),
        "i64": rng.integers(-2 ** 40, 2 ** 40, shape).astype(np.int64),
        "f32": (rng.standard_normal(shape) * 0.3 + 0.5).astype(np.float32),
        "f64": rng.standard_normal(shape) * 0.3 + 0.5,
    }
    out = {}
    for name, vol in vols.items():
        ds = get_2d_prediction_dataset(vol)
        x = np.stack([np.asarray(ds[i]) for i in range(len(ds))])       # (n, 1, hp, wp), the dtype numpy arrived at
        assert x.shape == (3, 1, 64, 64), x.shape
        out[name + "__in"] = vol
        out[name + "__x"] = x
        print("g9", name, vol.dtype, "->", x.dtype, float(x.min()), float(x.max()))
    np.savez_compressed(OUT / "g9_prediction_inputs_typed.npz", **out)


def gen_g10():
    """G10: the non-default training criteria (vol_seg_2d_trainer.py:124-148) evaluated by the reference's own classes
    (data/pytorch3dunet_losses.py: BCEDiceLoss, GeneralizedDiceLoss) and the torch modules it instantiates (BCEWithLogitsLoss,
    CrossEntropyLoss), value + gradient, for 1, 2 and 4 classes; targets as prepare_training_batch builds them."""
    from volume_segmantics.data.pytorch3dunet_losses import BCEDiceLoss, GeneralizedDiceLoss
    out = {}
    for k in (1, 2, 4):
        tg = torch.Generator().manual_seed(40 + k)
        logits = (torch.randn(3, k, 24, 40, generator=tg) * 2.0)
        lab = torch.randint(0, max(k, 2), (3, 24, 40), generator=tg)
        targets = torch.nn.functional.one_hot(lab, max(k, 2)).permute(0, 3, 1, 2)[:, :k].contiguous().to(torch.uint8)
        out[f"k{k}__logits"], out[f"k{k}__targets"] = logits.numpy(), targets.numpy()
        crits = {"BCEDiceLoss": (BCEDiceLoss(0.75, 0.25), lambda t: t.float()),
                 "BCELoss": (torch.nn.BCEWithLogitsLoss(), lambda t: t.float()),
                 "GeneralizedDiceLoss": (GeneralizedDiceLoss(), lambda t: t.float())}
        if k > 1:
            crits["CrossEntropyLoss"] = (torch.nn.CrossEntropyLoss(), lambda t: torch.argmax(t, dim=1))   # vol_seg_2d_trainer.py:425-428
        for name, (crit, conv) in crits.items():
            x = logits.clone().requires_grad_()
            loss = crit(x, conv(targets))
            (loss * 1.3).backward()
            out[f"k{k}__{name}__loss"], out[f"k{k}__{name}__grad"] = loss.detach().numpy(), x.grad.numpy()
            print("g10", k, name, float(loss))
    np.savez_compressed(OUT / "g10_losses.npz", **out)


def gen_decidable():
    """Which voxels of the g3 prediction goldens are DECIDABLE, i.e. where any correct fp32 implementation must reproduce the
    reference's label bit for bit: voxels where no direction involved is within rounding of a tie.  Computed with the oracle
    (oracle/predictor_numpy.py, pinned to g3 bit-exactly by tests/test_oracle_goldens.py) because it needs per-direction
    logits, which the reference's predictor does not return:
      * per direction d: margin_d = top-1 minus top-2 logit at the voxel; the direction's label is decidable if margin_d > 1e-3
        (north_star's logit tolerance), and its fp16 probability is then right to 1 ulp;
      * single-axis outputs: decidable = margin of that direction > 1e-3;
      * merged outputs (first-max of the fp16 probabilities over directions, vol_seg_2d_predictor.py:90-98): decidable if every
        direction's label is decidable AND the winner is clear - the best probability exceeds every other direction's by more
        than 2 fp16 ulps - or all directions within 2 ulps of the best carry the same label;
      * one-hot vote volumes: decidable if every contributing direction's label is decidable.
    Stored bit-packed; tests assert equality on the mask and that the mask covers > 99 % of the voxels."""
    from oracle import predictor_numpy as P
    from oracle.unet_resnet34_torch import seeded_oracle
    g3 = np.load(OUT / "g3_predict_29x64x40_c4.npz")
    vol, classes = g3["vol"], int(g3["classes"])
    net = seeded_oracle(classes=classes, seed=0).eval()
    views, v = [], vol
    for k in range(4):
        if k:
            v = np.rot90(v)
        for ax in (0, 1, 2):
            views.append((k, ax, v))
    margins, labels, probs = [], [], []
    for k, ax, v in views:
        l, p, logits = P.predict_single_axis(net, v, ax, return_logits=True)     # logits: (slices, K, h, w) along the axis
        top2 = np.sort(logits, axis=1)[:, -2:]
        m = P.rotate_array_to_axis(top2[:, 1] - top2[:, 0], ax)                    # back to the rotated volume's axes
        back = (lambda a: np.rot90(a, -k)) if k else (lambda a: a)
        margins.append(np.ascontiguousarray(back(m)))
        labels.append(np.ascontiguousarray(back(l)))
        probs.append(np.ascontiguousarray(back(p)))
    margins, labels, probs = np.stack(margins), np.stack(labels), np.stack(probs)
    assert np.array_equal(labels[0], g3["single_z_labels"]) and np.array_equal(labels[2], g3["single_x_labels"])
    dec_dir = margins > 1e-3

    def merged(dirs):
        pb = probs[dirs].astype(np.float16).view(np.uint16).astype(np.int32)    # fp16 bits order like the values (p >= 0)
        best = pb.max(0)
        near = pb >= best - 2
        lab_near_min = np.where(near, labels[dirs], 255).min(0)
        lab_near_max = np.where(near, labels[dirs], 0).max(0)
        clear = (near.sum(0) == 1) | (lab_near_min == lab_near_max)
        return dec_dir[dirs].all(0) & clear

    out = {"margin_threshold": np.float32(1e-3), "ulp_window": np.int32(2)}
    masks = {"single_z": dec_dir[0], "single_y": dec_dir[1], "single_x": dec_dir[2], "three": merged(list(range(3))),
             "twelve": merged(list(range(12))), "votes_z": dec_dir[0], "votes_three": dec_dir[:3].all(0),
             "votes_twelve": dec_dir.all(0)}
    for k2, m in masks.items():
        out[k2] = np.packbits(m.reshape(-1))
        print("decidable", k2, f"{m.mean():.5f}")
    out["shape"] = np.array(vol.shape)
    np.savez_compressed(OUT / "g3_decidable.npz", **out)


def main():
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    install_stubs()
    sys.path.insert(0, str(REF))
    torch.set_num_threads(8)
    if "--only-decidable" in sys.argv:
        gen_decidable()
        return

    import volume_segmantics.utilities.base_data_utils as utils
    from volume_segmantics.data.augmentations import get_padded_dimension
    from volume_segmantics.data.pytorch3dunet_losses import DiceLoss
    from volume_segmantics.data.pytorch3dunet_metrics import MeanIoU
    from volume_segmantics.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor
    from volume_segmantics.model.operations.vol_seg_2d_trainer import VolSeg2dTrainer
    from volume_segmantics.utilities.base_data_utils import Axis

    from oracle.unet_resnet34_torch import seeded_oracle

    OUT.mkdir(parents=True, exist_ok=True)
    if "--only-g9" in sys.argv:
        gen_g9()
        return
    if "--only-g10" in sys.argv:
        gen_g10()
        return
    gen_g8()
    if "--only-g8" in sys.argv:
        return
    gen_g9()
    gen_g10()
    utils.get_batch_size = lambda settings, prediction=False: 4 if prediction else 12  # needs CUDA in the reference

    # ---- G7: pad / crop tables --------------------------------------------------------------
    dims = np.arange(0, 200)
    padded = np.array([get_padded_dimension(int(d)) for d in dims])
    np.savez_compressed(OUT / "g7_padded_dimension.npz", dims=dims, padded=padded)

    # ---- G3/G4/G5: prediction through the reference's predictor --------------------------------
    classes = 4
    net = seeded_oracle(classes=classes, seed=0)
    pred = VolSeg2dPredictor.__new__(VolSeg2dPredictor)
    pred.model = net
    pred.settings = SimpleNamespace(cuda_device=0)
    pred.model_device_num = "cpu"
    pred.num_labels = classes
    pred.label_codes = {}
    vol = synth_volume((29, 64, 40), seed=1234)  # Z=29 -> padded 32, d=3: exercises the pad/crop quirk
    g = {"vol": vol, "classes": np.array(classes), "seed": np.array(0), "fingerprint": weight_fingerprint(net)}
    for name, ax in (("z", Axis.Z), ("y", Axis.Y), ("x", Axis.X)):
        l, p = pred._predict_single_axis(vol, output_probs=True, axis=ax)
        g[f"single_{name}_labels"], g[f"single_{name}_probs"] = l.copy(), p.copy()
        assert l.dtype == np.uint8 and p.dtype == np.float16 and l.shape == vol.shape
    l, p = pred._predict_3_ways_max_probs(vol)
    g["three_labels"], g["three_probs"] = l.copy(), p.copy()
    l, p = pred._predict_12_ways_max_probs(vol)
    g["twelve_labels"], g["twelve_probs"] = l.copy(), p.copy()
    g["onehot_z"] = pred._predict_single_axis_to_one_hot(vol, axis=Axis.Z)
    g["onehot_three"] = pred._predict_3_ways_one_hot(vol)
    g["onehot_twelve"] = pred._predict_12_ways_one_hot(vol)
    np.savez_compressed(OUT / "g3_predict_29x64x40_c4.npz", **g)
    print("prediction goldens done; label histogram:", np.bincount(g["twelve_labels"].ravel(), minlength=classes))

    # ---- merge alone: tie-heavy synthetic containers ------------------------------------------
    rng = np.random.default_rng(7)
    shape = (6, 10, 12)
    ndir = 12
    probs_choices = np.array([0.25, 0.5, 0.999, 1.0, 0.9995, 0.33325, 6e-5, 0.0], dtype=np.float16)
    dprobs = probs_choices[rng.integers(0, len(probs_choices), size=(ndir, *shape))]
    dlabels = rng.integers(0, 5, size=(ndir, *shape)).astype(np.uint8)
    lab = np.empty((2, *shape), np.uint8)
    prb = np.empty((2, *shape), np.float16)
    lab[0], prb[0] = dlabels[0], dprobs[0]
    chain_l, chain_p = [], []
    for d in range(1, ndir):
        lab[1], prb[1] = dlabels[d], dprobs[d]
        pred._merge_vols_in_mem(prb, lab)
        chain_l.append(lab[0].copy())
        chain_p.append(prb[0].copy())
    np.savez_compressed(OUT / "g4_merge_ties.npz", dlabels=dlabels, dprobs=dprobs,
                        chain_labels=np.stack(chain_l), chain_probs=np.stack(chain_p))

    # direction order / index maps of the 12-way scheme: a fake single-axis predictor that
    # returns the voxel's own flat index as the label lets the goldens record which voxel
    # each (direction, slice, pixel) touches (labels mod 251 to stay uint8-safe is not
    # needed - we record via probabilities instead)
    idxvol = np.arange(5 * 6 * 7, dtype=np.int64).reshape(5, 6, 7)
    calls = []

    def fake_single_axis(data_vol, output_probs=True, axis=Axis.Z):
        v = utils.rotate_array_to_axis(data_vol, axis)
        calls.append(np.ascontiguousarray(v))
        lab_ = utils.rotate_array_to_axis((np.ascontiguousarray(v) % 251).astype(np.uint8), axis)
        prob_ = utils.rotate_array_to_axis(np.full(v.shape, 0.5, np.float16), axis)
        return lab_, prob_

    pred._predict_single_axis = fake_single_axis
    l, _ = pred._predict_12_ways_max_probs(idxvol)
    assert np.array_equal(l, (idxvol % 251).astype(np.uint8))  # every direction maps back onto the voxel it read
    np.savez_compressed(OUT / "g4_direction_order.npz", idxvol=idxvol,
                        **{f"dir{d:02d}": c for d, c in enumerate(calls)})
    del pred._predict_single_axis

    # ---- G6: loss / metric / LR-finder / batch-prep KATs -----------------------------------
    tg = torch.Generator().manual_seed(11)
    logits = torch.randn(3, 4, 16, 24, generator=tg, requires_grad=True)
    mask = torch.randint(0, 4, (3, 16, 24), generator=tg, dtype=torch.uint8)
    img = torch.randn(3, 1, 16, 24, generator=tg)
    inputs, targets = utils.prepare_training_batch([img, mask], "cpu", 4)
    loss = DiceLoss(normalization="none")(logits, targets.float())
    loss.backward()
    probs = torch.softmax(logits.detach(), dim=1)
    miou = MeanIoU()(torch.unsqueeze(probs, 2), torch.unsqueeze(targets, 2))
    perfect = MeanIoU()(torch.unsqueeze(targets.float(), 2), torch.unsqueeze(targets, 2))
    lr_cases = {}
    for i, curve in enumerate((np.array([1.0, 0.9, 0.7, 0.2, 0.25, 0.9, 3.0]),
                               np.array([0.5, 0.6, 0.7, 0.9]),
                               np.linspace(1, 0, 9) ** 2)):
        lrs = list(np.geomspace(1e-6, 50, len(curve)))
        lr_cases[f"lr_curve{i}"] = curve
        lr_cases[f"lr_lrs{i}"] = np.array(lrs)
        lr_cases[f"lr_out{i}"] = np.array(
            VolSeg2dTrainer._find_lr_from_graph([torch.tensor(c) for c in curve], lrs))
    np.savez_compressed(OUT / "g6_loss_metric_lr.npz", logits=logits.detach().numpy(), mask=mask.numpy(),
                        targets=targets.numpy(), dice_loss=loss.detach().numpy(),
                        dice_grad=logits.grad.numpy(), mean_iou=miou.numpy(), mean_iou_perfect=perfect.numpy(),
                        **lr_cases)

    # ---- G2: three training steps through the reference's _train_one_batch ---------------------
    net2 = seeded_oracle(classes=2, seed=3, perturb_bn=False)
    tr = VolSeg2dTrainer.__new__(VolSeg2dTrainer)
    tr.model = net2
    tr.model_device_num = "cpu"
    tr.label_no = 2
    tr.settings = SimpleNamespace(loss_criterion="DiceLoss", pct_lr_inc=0.3)
    tr.loss_criterion = DiceLoss(normalization="none")
    tr.training_loader = [None] * 4  # len() only: steps_per_epoch
    tr._freeze_model()
    frozen = [n for n, p_ in net2.named_parameters() if not p_.requires_grad]
    tr._unfreeze_model()
    assert all(p_.requires_grad for p_ in net2.parameters())
    tr.optimizer = tr._create_optimizer(1e-3)
    sched = tr._create_oc_lr_scheduler(num_epochs=1, lr_to_use=2e-3)
    tg = torch.Generator().manual_seed(5)
    x = torch.randn(4, 1, 64, 64, generator=tg)
    m = (torch.rand(4, 64, 64, generator=tg) > 0.65).to(torch.uint8)
    losses, lrs, betas = [], [], []
    net2.train()
    for step in range(3):
        lrs.append(tr.optimizer.param_groups[0]["lr"])
        betas.append(tr.optimizer.param_groups[0]["betas"][0])
        losses.append(float(tr._train_one_batch(sched, [x, m])))
    sd = net2.state_dict()
    keep = ["segmentation_head.0.weight", "segmentation_head.0.bias", "decoder.blocks.4.conv2.0.weight",
            "decoder.blocks.4.conv2.1.weight", "decoder.blocks.4.conv2.1.running_mean",
            "decoder.blocks.4.conv2.1.running_var", "encoder.conv1.weight", "encoder.bn1.running_mean",
            "encoder.bn1.running_var", "encoder.layer4.2.bn2.weight", "encoder.layer2.0.downsample.0.weight"]
    np.savez_compressed(OUT / "g2_train3_b4_64.npz", x=x.numpy(), mask=m.numpy(), losses=np.array(losses),
                        lrs=np.array(lrs), beta1=np.array(betas), n_frozen=np.array(len(frozen)),
                        frozen_names=np.array(frozen), fingerprint0=weight_fingerprint(seeded_oracle(2, 3, False)),
                        **{"after__" + k: sd[k].numpy() for k in keep})
    print("train goldens done; losses", losses, "lrs", lrs, "beta1", betas, "frozen", len(frozen))
    gen_decidable()


if __name__ == "__main__":
    main()
