"""Time of the device pre-processing passes (csrc/preprocess.hip) on a 512^3 volume vs NumPy on the host (needs a GPU)."""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from volume_segmantics_amd import _lib as L
from volume_segmantics_amd.utilities import base_data_utils as U

DEV = "cuda:0"
rng = np.random.default_rng(0)
for name, vol in (("uint16", np.clip(rng.gamma(2.0, 5000.0, (512, 512, 512)), 0, 65535).astype(np.uint16)),
                  ("float32", (rng.standard_normal((512, 512, 512), dtype=np.float32) * 900 + 4000))):
    n = vol.size
    t0 = time.perf_counter(); dev, vtype = U.volume_to_device(vol, DEV); torch.cuda.synchronize(); t_up = time.perf_counter() - t0
    ws = torch.empty(L.lib.vs_volume_sum_workspace(n), dtype=torch.uint8, device=DEV)
    out2 = torch.empty(2, dtype=torch.float64, device=DEV)
    o = torch.empty(n, dtype=torch.uint8, device=DEV)
    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t_sum = timed(lambda: L.check(L.lib.vs_volume_sum(vtype, L.ptr(dev), n, 0, 0.0, L.ptr(ws), ws.numel(), L.ptr(out2), L.stream_ptr())))
    t_sq = timed(lambda: L.check(L.lib.vs_volume_sum(vtype, L.ptr(dev), n, 1, 4000.0, L.ptr(ws), ws.numel(), L.ptr(out2), L.stream_ptr())))
    t_clip = timed(lambda: L.check(L.lib.vs_clip_to_uint8(vtype, L.ptr(dev), n, 4000.0, 1000.0, 9000.0, L.ptr(o), None, L.stream_ptr())))
    b = vol.itemsize
    print(f"{name} 512^3: upload {t_up * 1e3:.0f} ms | sum {t_sum:.2f} ms ({n * b / t_sum * 1e-6:.0f} GB/s) | sq-dev {t_sq:.2f} ms ({n * b / t_sq * 1e-6:.0f} GB/s)"
          f" | clip {t_clip:.2f} ms ({n * (b + 1) / t_clip * 1e-6:.0f} GB/s)", flush=True)
    t0 = time.perf_counter(); m = np.nanmean(vol); t_m = time.perf_counter() - t0
    t0 = time.perf_counter(); U.clip_to_uint8(vol.copy(), m, 2.575); t_c = time.perf_counter() - t0
    t0 = time.perf_counter(); md = U.nanmean_device(vol, DEV); U.clip_to_uint8_device(vol, md, 2.575, DEV); torch.cuda.synchronize(); t_d = time.perf_counter() - t0
    print(f"   host NumPy: nanmean {t_m:.2f} s + clip_to_uint8 {t_c:.2f} s;  device path end to end (2 uploads, D2H): {t_d:.2f} s", flush=True)
