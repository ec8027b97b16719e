"""Does replaying the forward pass as a HIP graph beat eager launches on the GPU side?  (diagnostics; needs a GPU)"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from volume_segmantics_amd.engine import VolSegUnet
dev = "cuda:0"
model = VolSegUnet(2, device=dev, precision="bf16", seed=0)
x = torch.randn(32, 1, 256, 256, device=dev)
model.train()

def fwd():
    with torch.no_grad():
        return model._forward_impl(x, training=True)

def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): f()
    e1.record(); h = time.perf_counter() - t0
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, h / n * 1e3

print("eager   : gpu %.3f ms/fwd, host %.3f ms/fwd" % timeit(fwd))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): fwd()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fwd()
print("graph   : gpu %.3f ms/fwd, host %.3f ms/fwd" % timeit(g.replay))
