"""Sustained HBM rates of plain streaming kernels on this part (needs a GPU): what a read : write mix can reach, next to the
8 TB/s vendor peak the roofline fractions are quoted against.  torch's own elementwise kernels on 1 GiB tensors."""
import torch

dev = torch.device("cuda:0")
n = 1 << 29                      # 2^29 bf16 = 1 GiB
x = torch.randn(n, device=dev, dtype=torch.bfloat16)
y = torch.empty_like(x)
z = torch.randn(n, device=dev, dtype=torch.bfloat16)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


gib = n * 2
for name, fn, nbytes in (("read only (sum)", lambda: x.sum(), gib),
                         ("1 read : 1 write (copy)", lambda: y.copy_(x), 2 * gib),
                         ("2 reads : 1 write (add)", lambda: torch.add(x, z, out=y), 3 * gib),
                         ("write only (fill)", lambda: y.fill_(1.0), gib)):
    t = timeit(fn)
    print(f"{name:28s} {nbytes / t / 1e12:6.2f} TB/s  ({t * 1e3:.3f} ms for {nbytes / 2**30:.0f} GiB)")
