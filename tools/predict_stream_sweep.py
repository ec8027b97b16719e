import sys, pathlib
sys.path.insert(0, "/root/repo")
import torch, bench
from volume_segmantics_amd import _lib
dev = torch.device("cuda:0")
for mt in (4, 2, 1):
    _lib.set_option("conv_stream_min_tiles", mt)
    for b in (64, 128):
        r = bench.predict_bench(dev, 1, "bf16", 512, 4, 12, b)
        print("min_tiles", mt, "batch", b, r["seconds"], r["slices_per_s"], flush=True)
