"""BASELINE configs[4]'s network (DeepLabV3+ / efficientnet-b4) on 1024 x 1024 slices, for rocprofv3 --kernel-trace --stats (needs a GPU):
   python tools/predict_config5_probe.py [slices] [batch] [bf16|fp16]"""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from types import SimpleNamespace
import numpy as np
import torch
import bench
from volume_segmantics_amd.engine import VolSegUnet
from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor
from volume_segmantics_amd.utilities.base_data_utils import Axis

slices = int(sys.argv[1]) if len(sys.argv) > 1 else 96
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
precision = sys.argv[3] if len(sys.argv) > 3 else "bf16"
dev = torch.device("cuda:0")
model = VolSegUnet(2, device=dev, precision=precision, seed=1, encoder="efficientnet-b4", topology="deeplabv3plus")
model.eval()
pred = VolSeg2dPredictor.__new__(VolSeg2dPredictor)
pred.model, pred.num_labels, pred.label_codes = model, 2, {}
pred.settings = SimpleNamespace(cuda_device=0, prediction_batch_size=batch)
vol = np.tile(bench.synth_volume(256, seed=99)[:slices], (1, 4, 4))
pred._predict_single_axis(vol[:batch], axis=Axis.Z)
t0 = time.perf_counter()
pred._predict_single_axis(vol, axis=Axis.Z)
dt = time.perf_counter() - t0
print(f"{slices} slices of 1024^2, batch {batch}: {dt:.3f} s = {slices / dt:.0f} slices/s")
