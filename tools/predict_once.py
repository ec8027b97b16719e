"""One 512^3 12-direction prediction (for rocprofv3 --kernel-trace; needs a GPU)."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import bench
r = bench.predict_bench(torch.device("cuda:0"), 1, "bf16", 512, 4, 12, int(sys.argv[1]) if len(sys.argv) > 1 else 64)
print(r["seconds"], r["slices_per_s"])
