"""Per-workgroup phase timeline of conv_igemm launches (diagnostics; needs a GPU).

For each geometry: run the conv a few times, then once with vs_debug_probe set, and print where the workgroups spend
their time (100 MHz timestamps): issue of the first loads, arrival of the first chunk in LDS, main loop, epilogue.
"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1] / "tests"))
import numpy as np
import torch
import hip_helpers as H

L = H.lib()
DEV = "cuda:0"


def run(name, n, hw, cin, cout, k=3, stride=1, reps=20):
    d = H.conv_desc(L, 1, n, hw, hw, cin, cout, k, stride, k // 2)
    x = torch.randn(n, hw, hw, cin, device=DEV).bfloat16()
    w = (torch.randn(cout, k * k, cin, device=DEV) * 0.05).bfloat16()
    ho = hw // stride
    y = torch.empty(n, ho, ho, cout, device=DEV, dtype=torch.bfloat16)
    st = L.stream_ptr()
    call = lambda: L.check(L.lib.vs_conv2d_fwd(d, L.ptr(x), None, L.ptr(w), None, None, None, L.ptr(y), None, st))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flops = 2.0 * n * ho * ho * cout * cin * k * k
    cap = 1 << 16
    buf = torch.zeros(cap * 8, dtype=torch.int64, device=DEV)
    L.check(L.lib.vs_debug_probe(L.ptr(buf), cap))
    call(); torch.cuda.synchronize()
    L.check(L.lib.vs_debug_probe(None, 0))
    b = buf.cpu().numpy().reshape(cap, 8)
    b = b[b[:, 0] != 0]
    if len(b) == 0:
        print(f"== {name}: {us:.1f} us/launch (kernel without phase probe: direct / DMA path)")
        return
    t = b[:, :5].astype(np.float64) * 0.01  # us
    t -= t[:, 0].min()
    ph = np.diff(t, axis=1)
    xcc = b[:, 6] & 0xf
    cu = (b[:, 5] >> 8) & 0xf
    se = (b[:, 5] >> 13) & 0x7
    nslots = len(set(zip(xcc.tolist(), se.tolist(), cu.tolist())))
    print(f"== {name}: n={n} {hw}x{hw} {cin}->{cout} k{k} s{stride}: {us:.1f} us/launch back-to-back = {flops / us * 1e-6:.0f} TFLOP/s;"
          f" {len(b)} WGs on {nslots} CUs; in + out = {(x.numel() + y.numel()) * 2 / 1e6:.1f} MB")
    print(f"   span first start -> last end: {t[:, 4].max():.1f} us; WG starts spread {t[:, 0].max():.1f} us; WG lifetime mean {np.mean(t[:, 4] - t[:, 0]):.1f} us")
    for i, nm in enumerate(["addr calc + issue first loads", "first chunk lands in LDS", "main loop", "epilogue + store ack"]):
        print(f"   {nm:32s} mean {ph[:, i].mean():6.2f}  p50 {np.median(ph[:, i]):6.2f}  p95 {np.percentile(ph[:, i], 95):6.2f} us")
    # copy floor for the same bytes
    a = torch.empty(x.numel() + y.numel(), device=DEV, dtype=torch.bfloat16)
    c = torch.empty_like(a)
    for _ in range(3):
        c.copy_(a)
    e0.record()
    for _ in range(reps):
        c.copy_(a)
    e1.record(); torch.cuda.synchronize()
    print(f"   torch copy of in+out bytes (r+w = 2x): {e0.elapsed_time(e1) * 1e3 / reps:.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "predict":       # layer shapes of a batch-64 512x512 prediction forward
        for name, hw, cin, cout in [("layer1", 128, 64, 64), ("layer2", 64, 128, 128), ("layer3", 32, 256, 256), ("layer4", 16, 512, 512),
                                    ("dec0.c1", 32, 768, 256), ("dec0.c2", 32, 256, 256), ("dec1.c1", 64, 384, 128), ("dec1.c2", 64, 128, 128),
                                    ("dec2.c1", 128, 192, 64), ("dec2.c2", 128, 64, 64), ("dec3.c1", 256, 128, 32), ("dec3.c2", 256, 32, 32)]:
            run(name, 64, hw, cin, cout, reps=10)
        sys.exit(0)
    run("layer1", 32, 64, 64, 64)
    run("layer2", 32, 32, 128, 128)
    run("layer3", 32, 16, 256, 256)
    run("layer4", 32, 8, 512, 512)
    run("dec3.c2", 32, 128, 32, 32)
    run("dec4.c2", 32, 256, 16, 16)
    run("dec2.c1-like", 32, 64, 192, 64)
    run("layer1 x4 batch", 128, 64, 64, 64)
