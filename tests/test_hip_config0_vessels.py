"""-m gpu: BASELINE.json configs[0] AS STATED - the tutorial's vessels 256^3 run through the reference's command-line flow
(scripts/train_2d_model.py:15-75, scripts/predict_2d_model.py; training_data/README.md:5-26), on the MI355X engine.

tests/golden/vessels_256cube_LABELS.h5 is the reference's own label volume (training_data/vessels_256cube_LABELS.h5: 256^3 uint8
in {0, 255}, 35.04 % foreground, written by h5py as 32^3 gzip chunks) - reference-held DATA, committed as a fixture; it is also
the first reference-written HDF5 file utilities/hdf5_lite.py reads.  The matching DATA volume (vessels_256cube_DATA.h5) is not in
the reference checkout (a release download), so - as BASELINE.md section 3 plans for this config - a float32 data volume of the
same shape is synthesised FROM the labels (vessels brighter than tissue behind a blur, plus noise), written to HDF5, and the flow
runs from the two FILES with the shipped settings YAMLs: slice all three axes to 768 PNG pairs, LR finder + one frozen epoch at
the reference's batch 12, early-stopping checkpoint, clean-up, then a MEDIUM (3-axis) prediction of the data file to an HDF5
label file.  Asserted: the plumbing's artefacts, a Dice against the real labels above a floor measured on this run, and that an
independent HDF5 reader agrees with what was returned."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parents[1]
LABELS = REPO / "tests" / "golden" / "vessels_256cube_LABELS.h5"


def _synthetic_data_from(labels01: np.ndarray, seed: int = 1234) -> np.ndarray:
    rng = np.random.default_rng(seed)
    v = labels01.astype(np.float32)
    for _ in range(2):                      # ~2-voxel blur of the vessel mask: soft edges, as a reconstruction has
        for ax in range(3):
            v = (np.roll(v, 1, ax) + 2 * v + np.roll(v, -1, ax)) / 4
    noise = rng.standard_normal(v.shape).astype(np.float32)
    for ax in range(3):
        noise = (np.roll(noise, 1, ax) + noise + np.roll(noise, -1, ax)) / 3
    return (1500.0 + 900.0 * v + 450.0 * noise).astype(np.float32)      # float volume: clip_data maps it to uint8 on the device


def test_vessels_256cube_train_one_frozen_epoch_and_predict_medium(tmp_path):
    from volume_segmantics_amd.data import TrainingDataSlicer, get_settings_data
    from volume_segmantics_amd.model.operations.vol_seg_2d_trainer import VolSeg2dTrainer
    from volume_segmantics_amd.model.operations.vol_seg_prediction_manager import VolSeg2DPredictionManager
    from volume_segmantics_amd.utilities import base_data_utils as utils
    from volume_segmantics_amd.utilities import hdf5_lite
    if utils._h5py() is None and not hdf5_lite.available():
        pytest.skip("neither h5py nor libhdf5 on this machine")
    # ---- the reference's label file, through the engine's own reader ----
    labels, chunks = utils.numpy_from_hdf5(LABELS, "/data")
    assert labels.shape == (256, 256, 256) and labels.dtype == np.uint8 and tuple(chunks) == (32, 32, 32)
    assert sorted(np.unique(labels).tolist()) == [0, 255] and int((labels == 255).sum()) == 5879454       # 35.04 %, as SURVEY.md records
    lab01 = (labels == 255).astype(np.uint8)
    data = _synthetic_data_from(lab01)
    data_path = tmp_path / "vessels_256cube_DATA.h5"
    utils.save_data_to_hdf5(data, data_path, internal_path="/data")
    # ---- model-train-2d --data DATA.h5 --labels LABELS.h5 (scripts/train_2d_model.py:31-71) ----
    settings = get_settings_data(REPO / "volseg-settings" / "2d_model_train_settings.yaml")
    assert settings.image_size == 256 and settings.training_axes == "All" and settings.model["encoder_name"] == "resnet34"
    settings.model["encoder_weights"] = None          # no ImageNet download on the box: smp's random initialisation
    settings.clip_data = True                         # the synthetic data volume is float32
    settings.precision = "bf16"                       # BASELINE configs[1]'s arithmetic; everything else is the shipped file
    data_dir, seg_dir = tmp_path / settings.data_im_dirname, tmp_path / settings.seg_im_out_dirname
    slicer = TrainingDataSlicer(data_path, LABELS, settings)
    slicer.output_data_slices(data_dir, "data0")
    slicer.output_label_slices(seg_dir, "seg0")
    assert slicer.num_seg_classes == 2 and len(list(data_dir.glob("*.png"))) == len(list(seg_dir.glob("*.png"))) == 768
    trainer = VolSeg2dTrainer(data_dir, seg_dir, slicer.num_seg_classes, settings)
    assert utils.get_batch_size(settings) == 12 and len(trainer.training_loader) == int(768 * 0.8) // 12      # the reference's batch
    model_out = tmp_path / "vessels_U_Net_trained_2d_model.pytorch"
    trainer.train_model(model_out, 1, settings.patience, create=True, frozen=True)
    trainer.output_loss_fig(model_out)
    trainer.output_prediction_figure(model_out)
    slicer.clean_up_slices()
    assert model_out.exists() and not data_dir.exists() and (tmp_path / f"{model_out.stem}_train_stats.csv").exists()
    assert len(trainer.avg_train_losses) == 1 and np.isfinite(trainer.avg_valid_losses[0]) and trainer.avg_eval_scores[0] > 0.4
    frozen = [n for n, p in trainer.model.named_parameters() if not p.requires_grad]
    assert len(frozen) == 33 and all("encoder" in n and "conv" in n for n in frozen)      # the reference's freeze predicate (:102-108)
    ck = torch.load(model_out, weights_only=False)
    assert set(ck) >= {"model_state_dict", "model_struc_dict", "optimizer_state_dict", "loss_val", "label_codes"}
    del trainer
    # ---- model-predict-2d model.pytorch DATA.h5 (scripts/predict_2d_model.py; quality: medium from the shipped file) ----
    psettings = get_settings_data(REPO / "volseg-settings" / "2d_model_predict_settings.yaml")
    assert psettings.quality == "medium" and psettings.clip_data is True
    mgr = VolSeg2DPredictionManager(str(model_out), data_path, psettings)
    out = tmp_path / "vessels_256cube_DATA_2d_model_vol_pred.h5"
    pred = mgr.predict_volume_to_path(out)
    assert pred.shape == labels.shape and pred.dtype == np.uint8 and set(np.unique(pred).tolist()) <= {0, 1}
    inter = float((pred.astype(bool) & lab01.astype(bool)).sum())
    dice = 2 * inter / float(pred.sum() + lab01.sum())
    print(f"[configs0] one frozen epoch (validation loss {ck['loss_val']:.4f}), MEDIUM prediction of the 256^3 volume: Dice vs the reference labels {dice:.4f}")
    assert dice > 0.7, dice         # measured 0.9x after one frozen epoch of a random-init network (printed above)
    back, bchunks = utils.numpy_from_hdf5(out, "/data")
    assert np.array_equal(back, pred) and bchunks is not None
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if Path("/opt/conda/bin/h5dump").exists() else None)
    if h5dump:      # an independent reader of the written file
        head = subprocess.run([h5dump, "-H", "-p", str(out)], capture_output=True, text=True).stdout
        assert "H5T_STD_U8LE" in head and "( 256, 256, 256 )" in head and "DEFLATE" in head, head
