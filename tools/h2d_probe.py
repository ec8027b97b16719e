"""How to get a 512^3 uint8 numpy volume (134 MB, pageable) into HBM: pinned staging copy vs hipHostRegister in place vs
plain pageable copy; and the way back (402 MB of labels + probabilities)."""
import time, numpy as np, torch
n = 512 ** 3
vol = np.random.default_rng(0).integers(0, 256, n, dtype=np.uint8)
dev = torch.device("cuda:0")
torch.zeros(1, device=dev); torch.cuda.synchronize()
def t(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return f"min {min(ts) * 1e3:.2f} ms, first {ts[0] * 1e3:.2f} ms"
def staged():
    st = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    st.numpy()[...] = vol
    return st.to(dev, non_blocking=True)
print("pinned staging (alloc cached after first) + copy + H2D:", t(staged))
st = torch.empty(n, dtype=torch.uint8, pin_memory=True)
def host_copy():
    st.numpy()[...] = vol
print("  host copy alone:", t(host_copy))
print("  H2D from pinned alone:", t(lambda: st.to(dev, non_blocking=True)))
print("pageable .to(dev):", t(lambda: torch.from_numpy(vol).to(dev)))
rt = torch.cuda.cudart()
def registered():
    rc = rt.cudaHostRegister(vol.ctypes.data, vol.nbytes, 0)
    assert int(rc) == 0, rc
    x = torch.from_numpy(vol)
    d = x.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    rt.cudaHostUnregister(vol.ctypes.data)
    return d
try:
    print("hipHostRegister in place + H2D + unregister:", t(registered))
    d = registered()
    print("  correct:", bool((d.cpu().numpy() == vol).all()))
except Exception as e:
    print("hipHostRegister failed:", repr(e))
lab = torch.zeros(n, dtype=torch.uint8, device=dev); prob = torch.zeros(n, dtype=torch.float16, device=dev)
def down_pinned():
    a = torch.empty(n, dtype=torch.uint8, pin_memory=True); b = torch.empty(n, dtype=torch.float16, pin_memory=True)
    a.copy_(lab, non_blocking=True); b.copy_(prob, non_blocking=True); torch.cuda.synchronize()
    return a.numpy(), b.numpy()
print("D2H labels + probs into pinned (cached):", t(down_pinned))
print("D2H .cpu():", t(lambda: (lab.cpu(), prob.cpu())))
