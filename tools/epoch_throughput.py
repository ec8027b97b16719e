"""End-to-end `VolSeg2dTrainer.train_model` epoch throughput on a synthetic 256^3 volume (needs a GPU): the volume and its labels
go through TrainingDataSlicer (PNG slices along all three axes, as `model-train-2d` does), then epochs are timed with (a) the
reference's feed - every PNG decoded again in every epoch by 4 DataLoader workers, batches augmented on the device - and (b) the
resident feed (slices decoded once, kept as uint8 in HBM).  Prints slices/s per epoch next to the bare training step's rate.
    python tools/epoch_throughput.py [cube=256] [epochs=3] [batch=32]"""
import logging
import pathlib
import sys
import tempfile
import time
from types import SimpleNamespace

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch

import bench
from volume_segmantics_amd.data.settings_data import get_settings_data
from volume_segmantics_amd.data.slicers import TrainingDataSlicer
from volume_segmantics_amd.model.operations.vol_seg_2d_trainer import VolSeg2dTrainer


def main():
    cube = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    root = pathlib.Path(tempfile.mkdtemp(prefix="volseg_epoch_"))
    vol = bench.synth_volume(cube, seed=11)
    labels = (vol > np.percentile(vol, 70)).astype(np.uint8)
    settings = get_settings_data(pathlib.Path(__file__).resolve().parents[1] / "volseg-settings" / "2d_model_train_settings.yaml")
    settings.image_size, settings.batch_size, settings.precision = cube, batch, "bf16"
    settings.model = dict(settings.model, encoder_weights=None)
    settings.cuda_device, settings.clip_data, settings.downsample = 0, True, False
    settings.lr_find_epochs, settings.plot_lr_graph = 1, False
    t0 = time.perf_counter()
    slicer = TrainingDataSlicer(vol, labels, settings)
    slicer.output_data_slices(root / "data", "data")
    slicer.output_label_slices(root / "seg", "seg")
    n_slices = len(list((root / "data").glob("*.png")))
    print(f"sliced {n_slices} PNG pairs in {time.perf_counter() - t0:.1f} s")
    results = {}
    for name, resident in (("PNG decode per epoch, 4 workers", False), ("resident uint8 slices in HBM", True)):
        settings.resident_feed = resident
        trainer = VolSeg2dTrainer(root / "data", root / "seg", slicer.num_seg_classes, settings)
        steps = len(trainer.training_loader)
        times = []
        orig = logging.info
        def spy(msg, *a, **k):
            if isinstance(msg, str) and msg.startswith("Time taken for epoch"):
                times.append(float(msg.split(":")[1].split()[0]))
            return orig(msg, *a, **k)
        logging.info = spy
        try:
            trainer.train_model(root / f"model_{int(resident)}.pytorch", epochs, patience=epochs + 1, create=True, frozen=False)
        finally:
            logging.info = orig
        per_epoch = n_slices            # training steps + the validation pass see every slice once per epoch
        results[name] = (steps, times)
        print(f"{name}: {steps} training steps of {batch} + validation per epoch; epoch times {[round(t, 2) for t in times]} s -> "
              f"{per_epoch / min(times):.0f} slices/s in the best epoch (training part alone >= {steps * batch / min(times):.0f})")
    print("bare training step on this box (bench.py): see profiles/r3_bench_default.json (6 900 slices/s)")


if __name__ == "__main__":
    main()
