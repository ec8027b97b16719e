import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1])); sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1] / "tests"))
import torch, hip_helpers as H
L = H.lib(); DEV = "cuda:0"
def run(name, n, hw, cin, cout, k=3, reps=20):
    d = H.conv_desc(L, 1, n, hw, hw, cin, cout, k, 1, k // 2)
    x = torch.randn(n, hw, hw, cin, device=DEV).bfloat16(); w = (torch.randn(cout, k * k, cin, device=DEV) * 0.05).bfloat16()
    y = torch.empty(n, hw, hw, cout, device=DEV, dtype=torch.bfloat16); st = L.stream_ptr()
    call = lambda: L.check(L.lib.vs_conv2d_fwd(d, L.ptr(x), None, L.ptr(w), None, None, None, L.ptr(y), None, st))
    for _ in range(3): call()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize(); us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name}: {us:.1f} us = {2.0 * n * hw * hw * cout * cin * k * k / us * 1e-6:.0f} TFLOP/s", flush=True)
import os
if os.environ.get("VS_OPT"): k, v = os.environ["VS_OPT"].split("="); L.set_option(k, int(v))
run("layer4 p", 64, 16, 512, 512); run("layer3 p", 64, 32, 256, 256); run("layer2 p", 64, 64, 128, 128); run("layer1 p", 64, 128, 64, 64)
run("layer3 t", 32, 16, 256, 256); run("layer2 t", 32, 32, 128, 128)
