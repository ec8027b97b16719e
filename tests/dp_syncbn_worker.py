"""Worker of tests/test_hip_dp_rehearsal.py::test_two_rank_sync_batchnorm_equals_one_rank_at_twice_the_batch: two ranks on GPU 0 over
gloo (a rehearsal of the RCCL run).  With ``model.sync_bn = True`` every BatchNorm normalises with the statistics of the GLOBAL
batch: each rank's training logits for its shard must equal - bit for bit, the statistics being integer sums - the corresponding
rows of ONE process running both shards as one batch, the running statistics likewise, and the all-reduced gradients must match
the single process's (sum over the global batch) to bf16 noise."""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from volume_segmantics_amd import _lib as L  # noqa: E402
from volume_segmantics_amd.engine import VolSegUnet  # noqa: E402

TOPOLOGY = sys.argv[1] if len(sys.argv) > 1 else "unet"
ENCODER = sys.argv[2] if len(sys.argv) > 2 else "resnet34"
HW, B = 64, 4


def shard(rank):
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(B, 1, HW, HW, generator=g)
    go = torch.randn(B, 2, HW, HW, generator=g) * 1e-3
    return x, go


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    # one kernel configuration whatever the batch (the tile shapes - and with them the per-tile partial sums that go into the
    # fixed-point statistics - otherwise follow the number of workgroups a launch would have)
    # ... and statistics in fixed-point bins for every unit in the single process too (it would sum the few tiles of these small
    # maps as fp32 partial rows otherwise - the same numbers to ~1e-7, not the same bits)
    for k, v in (("conv_nw8", 0), ("conv_ring", 0), ("conv_min_wgs", 1), ("conv_direct_min_px", 1 << 30), ("conv_stream", 0), ("bn_inline_rows", 0)):
        L.set_option(k, v)

    def fresh(sync, group):
        m = VolSegUnet(2, device=dev, precision="bf16", seed=11, encoder=ENCODER, topology=TOPOLOGY)
        dist.broadcast(m._flat, 0)
        dist.broadcast(m._bnstate, 0)
        m.dp_group = group
        m.sync_bn = sync
        m.dp_buckets = 1
        m.train()
        return m

    x, go = shard(rank)
    x, go = x.to(dev), go.to(dev)
    # (a) two ranks, SyncBatchNorm
    a = fresh(True, dist.group.WORLD)
    la = a(x)
    la.backward(go)
    torch.cuda.synchronize()
    assert a._plans[(HW, HW)]["sync"][2]["error"] is None, a._plans[(HW, HW)]["sync"][2]["error"]
    ga = a._flat_grad.clone() * world                      # the engine averages the gradients over the ranks
    # (b) the same two shards as ONE batch in this process alone
    xs, gs = zip(*[shard(r) for r in range(world)])
    b = fresh(False, None)
    lb = b(torch.cat(xs).to(dev))
    lb.backward(torch.cat(gs).to(dev))
    torch.cuda.synchronize()
    mine = lb[rank * B:(rank + 1) * B]
    if os.environ.get("VS_SYNC_DEBUG") and rank == 0:
        sa, sb = a.state_dict(), b.state_dict()
        for k in sa:
            if "running" in k:
                d = (sa[k].double() - sb[k].double()).abs().max().item()
                print(f"[debug] {k}: max abs diff {d:.3e} (scale {sb[k].double().abs().max().item():.3e})")
        print("[debug] logits scale", lb.abs().max().item(), "diff", (la - mine).abs().max().item())
    assert torch.equal(la, mine), ("training logits differ from the single-process global batch", (la - mine).abs().max().item())
    assert torch.equal(a._bnstate, b._bnstate), ("running statistics", (a._bnstate - b._bnstate).abs().max().item())
    g0, g1 = b._flat_grad.double(), ga.double()
    rel = ((g0 - g1).norm() / g0.norm()).item()
    cos = (torch.dot(g0, g1) / (g0.norm() * g1.norm())).item()
    assert rel < 3e-2 and cos > 0.999, ("gradients", rel, cos)
    # (d) the whole training step with the reference's loss: SyncBatchNorm + the Dice of the GLOBAL batch (HipDiceLoss(global_group))
    # against one process on both shards - the same loss value, the same parameter gradients
    from volume_segmantics_amd.data.losses import HipDiceLoss

    def targets(r):
        gt = torch.Generator().manual_seed(300 + r)
        return torch.nn.functional.one_hot((torch.rand(B, HW, HW, generator=gt) > 0.5).long(), 2).permute(0, 3, 1, 2).float()

    a2 = fresh(True, dist.group.WORLD)
    loss_a = HipDiceLoss(global_group=dist.group.WORLD)(a2(x), targets(rank).to(dev))
    loss_a.backward()
    b2 = fresh(False, None)
    loss_b = HipDiceLoss()(b2(torch.cat(xs).to(dev)), torch.cat([targets(r) for r in range(world)]).to(dev))
    loss_b.backward()
    torch.cuda.synchronize()
    assert abs(loss_a.item() - loss_b.item()) < 1e-6, ("global Dice", loss_a.item(), loss_b.item())
    h0, h1 = b2._flat_grad.double(), a2._flat_grad.double()        # (the factor `world` already rides in the loss gradient)
    rel2 = ((h0 - h1).norm() / h0.norm()).item()
    cos2 = (torch.dot(h0, h1) / (h0.norm() * h1.norm())).item()
    assert rel2 < 3e-2 and cos2 > 0.999, ("gradients of the global Dice step", rel2, cos2)
    # (c) per-rank statistics (the default) are a different computation: the test above is not vacuous
    c = fresh(False, dist.group.WORLD)
    lc = c(x)
    torch.cuda.synchronize()
    assert not torch.equal(lc, mine)
    both = [torch.zeros(2) for _ in range(world)]
    dist.all_gather(both, torch.tensor([rel, cos]))
    if rank == 0:
        print(f"sync_bn: gradients vs the single process: relative L2 {[round(v[0].item(), 5) for v in both]}, cosine {[round(v[1].item(), 6) for v in both]}")
        print(f"sync_bn + global Dice: loss {loss_a.item():.7f} vs {loss_b.item():.7f}, gradients relative L2 {rel2:.5f}, cosine {cos2:.6f}")
        print("DP_SYNCBN_OK")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
