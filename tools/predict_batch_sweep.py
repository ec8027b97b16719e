"""512^3 12-direction prediction time vs prediction batch size (diagnostics; needs a GPU)."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import bench
dev = torch.device("cuda:0")
for b in (8, 16, 32, 64, 128, 256):
    r = bench.predict_bench(dev, 1, "bf16", 512, 4, 12, b)
    print(b, r["seconds"], r["slices_per_s"], flush=True)
