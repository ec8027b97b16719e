"""Worker of tests/test_hip_dp_rehearsal.py: two ranks on GPU 0 over gloo (a rehearsal of the RCCL run).  Trains the same
data-parallel steps with the optimiser step (a) inside backward, bucket by bucket, and (b) as a separate step() after the
all-reduce and (c) as the recorded step (VolSegUnet.fused_train_step at world size 2: hipGraphs with the bucket all-reduces
between them); the parameters, optimiser state and losses must agree bit for bit, and be identical on both ranks."""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from volume_segmantics_amd.data.losses import HipDiceLoss  # noqa: E402
from volume_segmantics_amd.engine import VolSegUnet  # noqa: E402


TOPOLOGY = sys.argv[1] if len(sys.argv) > 1 else "unet"       # BASELINE configs[3] trains U-Net++ / ResNet-50 data-parallel
ENCODER = sys.argv[2] if len(sys.argv) > 2 else "resnet34"


def run(fuse: bool, frozen: bool, rank: int, graph: bool = False, low: bool = False):
    dev = torch.device("cuda", 0)
    model = VolSegUnet(2, device=dev, precision="bf16", seed=11, encoder=ENCODER, topology=TOPOLOGY)
    dist.broadcast(model._flat, 0)
    dist.broadcast(model._bnstate, 0)
    model.dp_group = dist.group.WORLD
    if low:
        model.dp_grad_dtype = torch.bfloat16        # the gradient buckets travel as bf16
    if frozen:
        for name, p in model.named_parameters():
            if "encoder" in name and "conv" in name:
                p.requires_grad = False
    g = torch.Generator().manual_seed(100 + rank)                      # every rank: its own shard of the global batch
    x = torch.randn(4, 1, 64, 64, generator=g).to(dev)
    t = torch.nn.functional.one_hot((torch.rand(4, 64, 64, generator=g) > 0.5).long(), 2).permute(0, 3, 1, 2).float().to(dev)
    opt = model.fused_adamw(lr=1e-3, fuse_step_into_backward=fuse)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=2e-3, total_steps=8, pct_start=0.3)
    crit = HipDiceLoss()
    model.train()
    losses = []
    for _ in range(5):
        if graph:       # the recorded form: linear hipGraphs per unit range and stream, the bucket all-reduces between them
            assert model.can_fuse_step(opt, x, t)
            loss = model.fused_train_step(x, t, opt, eps=crit.epsilon)
        else:
            opt.zero_grad()
            loss = crit(model(x), t)
            loss.backward()
            opt.step()
        sched.step()
        losses.append(loss.item())
    if graph:
        st = next(iter(model._steps.values()))
        assert st["graphs"][0] is not None and st["graphs"][1] is not None
        n_reduce = sum(1 for op, _ in st["graphs"][0] if op == "reduce")
        assert n_reduce == 4 if TOPOLOGY == "unet" else n_reduce >= 2, n_reduce
    model.eval()
    with torch.no_grad():
        ev = model(x)                                                   # reads the weight copies of the flipped set
    torch.cuda.synchronize()
    return losses, model._flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), ev.clone()


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    for frozen in (False, True):
        a = run(True, frozen, rank)
        b = run(False, frozen, rank)
        c = run(True, frozen, rank, graph=True)
        for o, what in ((b, "step() after the all-reduce"), (c, "recorded step")):
            assert a[0] == o[0], ("losses", what, frozen, a[0], o[0])
            for u, v, nm in zip(a[1:], o[1:], ("params", "exp_avg", "exp_avg_sq", "eval logits")):
                assert torch.equal(u, v), (nm, what, frozen, (u - v).abs().max().item())
        mine = a[1].cpu()
        other = mine.clone()
        dist.broadcast(other, 0)
        assert torch.equal(mine, other), "ranks diverged"
        if TOPOLOGY == "unet" and ENCODER == "resnet34" and not frozen:
            # bf16 transport of the gradient buckets: the step inside backward and the step after the all-reduce still agree bit
            # for bit, the ranks stay identical, and five AdamW steps end within a few learning rates of the fp32-transport run
            d = run(True, frozen, rank, low=True)
            e = run(False, frozen, rank, low=True)
            assert d[0] == e[0], ("losses, bf16 gradient transport", d[0], e[0])
            for u, v, nm in zip(d[1:], e[1:], ("params", "exp_avg", "exp_avg_sq", "eval logits")):
                assert torch.equal(u, v), (nm, "bf16 gradient transport", (u - v).abs().max().item())
            mine = d[1].cpu(); other = mine.clone()
            dist.broadcast(other, 0)
            assert torch.equal(mine, other), "ranks diverged (bf16 gradient transport)"
            assert not torch.equal(d[1], a[1]) and (d[1] - a[1]).abs().max().item() < 2e-2 and (d[1] - a[1]).abs().mean().item() < 2e-3
            assert abs(d[0][-1] - a[0][-1]) < 5e-2
    dist.barrier()
    if rank == 0:
        print("DP_REHEARSAL_OK", TOPOLOGY, ENCODER)


if __name__ == "__main__":
    main()
