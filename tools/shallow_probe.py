"""Timing of the shallow-layer conv kernel variants (diagnostics; needs a GPU)."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1] / "tests"))
import torch
import hip_helpers as H
L = H.lib()
DEV = "cuda:0"

def run(tag, n, hw, cin, cout, reps=20, up0=0):
    d = H.conv_desc(L, 1, n, hw, hw, cin, cout, 3, 1, 1, up0=up0)
    x = torch.randn(n, hw >> up0, hw >> up0, cin, device=DEV).bfloat16()
    w = (torch.randn(cout, 9, cin, device=DEV) * 0.05).bfloat16()
    y = torch.empty(n, hw, hw, cout, device=DEV, dtype=torch.bfloat16)
    st = L.stream_ptr()
    call = lambda: L.check(L.lib.vs_conv2d_fwd(d, L.ptr(x), None, L.ptr(w), None, None, None, L.ptr(y), None, st))
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    mb = (x.numel() + y.numel()) * 2 / 1e6
    print(f"{tag:44s} {cin}->{cout} @{hw} up{up0}: {us:7.1f} us  ({mb / us:.0f} GB/s of in+out... {mb:.0f} MB)")

for opts in [dict(conv_direct=0), dict(conv_direct=1, conv_direct_rows=32), dict(conv_direct=1, conv_direct_rows=16), dict(conv_direct=1, conv_direct_rows=64)]:
    for k, v in opts.items(): L.set_option(k, v)
    run(str(opts), 32, 256, 16, 16)
    run(str(opts), 32, 256, 32, 16, up0=1)
    run(str(opts), 32, 128, 32, 32)
    run(str(opts), 32, 256, 16, 32)
