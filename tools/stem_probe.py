"""Timing of the stem kernels (diagnostics; needs a GPU)."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1] / "tests"))
import torch
import hip_helpers as H
L = H.lib()
DEV = "cuda:0"

def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for n, hw in [(32, 256), (16, 512)]:
    x = torch.randn(n, hw, hw, device=DEV)
    w = torch.randn(64, 49, device=DEV) / 7
    y = torch.empty(n, hw // 2, hw // 2, 64, device=DEV, dtype=torch.bfloat16)
    dy = torch.randn(n, hw // 2, hw // 2, 64, device=DEV).bfloat16()
    dw = torch.empty(64, 49, device=DEV)
    wsb = L.lib.vs_stem_wgrad_workspace(n, hw, hw)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    st = L.stream_ptr()
    for opt in (0, 1):
        L.set_option("stem_bf16", opt)
        tf = timeit(lambda: L.check(L.lib.vs_stem_fwd(1, L.ptr(x), L.ptr(w), None, None, 0, L.ptr(y), n, hw, hw, st)))
        tw = timeit(lambda: L.check(L.lib.vs_stem_wgrad(1, L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(ws), wsb, n, hw, hw, st)))
        print(f"n={n} {hw}x{hw} stem_bf16={opt}: fwd {tf:.1f} us ({y.numel() * 2 / tf * 1e-3:.0f} GB/s out), wgrad+reduce {tw:.1f} us")
