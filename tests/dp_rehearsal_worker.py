"""Worker of tests/test_hip_dp_rehearsal.py: two ranks on GPU 0 over gloo (a rehearsal of the RCCL run).  Trains the same
data-parallel steps with the optimiser step (a) inside backward, bucket by bucket, and (b) as a separate step() after the
all-reduce; the parameters, optimiser state and losses must agree bit for bit, and be identical on both ranks."""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from volume_segmantics_amd.data.losses import HipDiceLoss  # noqa: E402
from volume_segmantics_amd.engine import VolSegUnet  # noqa: E402


def run(fuse: bool, frozen: bool, rank: int):
    dev = torch.device("cuda", 0)
    model = VolSegUnet(2, device=dev, precision="bf16", seed=11)
    dist.broadcast(model._flat, 0)
    dist.broadcast(model._bnstate, 0)
    model.dp_group = dist.group.WORLD
    if frozen:
        for name, p in model.named_parameters():
            if "encoder" in name and "conv" in name:
                p.requires_grad = False
    g = torch.Generator().manual_seed(100 + rank)                      # every rank: its own shard of the global batch
    x = torch.randn(4, 1, 64, 64, generator=g).to(dev)
    t = torch.nn.functional.one_hot((torch.rand(4, 64, 64, generator=g) > 0.5).long(), 2).permute(0, 3, 1, 2).float().to(dev)
    opt = model.fused_adamw(lr=1e-3, fuse_step_into_backward=fuse)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=2e-3, total_steps=6, pct_start=0.3)
    crit = HipDiceLoss()
    model.train()
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = crit(model(x), t)
        loss.backward()
        opt.step()
        sched.step()
        losses.append(loss.item())
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
        assert a[0] == b[0], ("losses", frozen, a[0], b[0])
        for u, v, nm in zip(a[1:], b[1:], ("params", "exp_avg", "exp_avg_sq", "eval logits")):
            assert torch.equal(u, v), (nm, frozen, (u - v).abs().max().item())
        mine = a[1].cpu()
        other = mine.clone()
        dist.broadcast(other, 0)
        assert torch.equal(mine, other), "ranks diverged"
    dist.barrier()
    if rank == 0:
        print("DP_REHEARSAL_OK")


if __name__ == "__main__":
    main()
