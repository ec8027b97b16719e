"""A/B of one runtime option on the 512^3 12-direction prediction, interleaved in ONE process (needs a GPU):
    python tools/ab_predict.py conv_stream 0 1 [rounds] [batch]
Box-to-box and run-to-run spread exceeds most kernel-level effects: only same-process alternation separates them."""
import pathlib
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch

import bench
from volume_segmantics_amd import _lib
from volume_segmantics_amd.engine import VolSegUnet
from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor


def main():
    opt, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else 64
    precision = sys.argv[6] if len(sys.argv) > 6 else "bf16"
    dev = torch.device("cuda:0")
    model = VolSegUnet(4, device=dev, precision=precision, seed=1)
    model.eval()
    pred = VolSeg2dPredictor.__new__(VolSeg2dPredictor)
    pred.model, pred.num_labels, pred.label_codes = model, 4, {}
    pred.settings = SimpleNamespace(cuda_device=0, prediction_batch_size=batch)
    vol = bench.synth_volume(512, seed=5678)
    out = {}
    for v in (v0, v1):
        _lib.set_option(opt, v)
        out[v] = pred._predict_12_ways_max_probs(vol)     # warm-up of both settings: plans, workspaces, code objects
    same = np.array_equal(out[v0][0], out[v1][0]) and np.array_equal(out[v0][1].view(np.uint16), out[v1][1].view(np.uint16))
    print(f"labels and probabilities bit-identical between {opt}={v0} and {opt}={v1}: {same}")
    times = {v0: [], v1: []}
    for _ in range(rounds):
        for v in (v0, v1):
            _lib.set_option(opt, v)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pred._predict_12_ways_max_probs(vol)
            torch.cuda.synchronize()
            times[v].append(time.perf_counter() - t0)
    for v in (v0, v1):
        t = sorted(times[v])
        print(f"{opt}={v}: median {t[len(t) // 2]:.4f} s  min {t[0]:.4f} s  ({6144 / t[len(t) // 2]:.0f} slices/s)  all {[round(x, 4) for x in times[v]]}")


if __name__ == "__main__":
    main()
