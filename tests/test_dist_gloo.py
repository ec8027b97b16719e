"""world_size-2 gloo tests (CPU) of the N>1 paths: sharded 12-direction prediction meeting in ONE max all-reduce of
the packed keys, one-hot vote sums, and the data-parallel gradient all-reduce of the flat gradient buffer."""
import os
import socket
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      VOLSEG_DIST_TIMEOUT_S="180")
    torch.set_num_threads(2)
    import torch.distributed as dist
    from cpu_backend import OracleBackend
    from oracle.unet_resnet34_torch import seeded_oracle
    from volume_segmantics_amd import dist as vdist
    from volume_segmantics_amd.engine import VolSegUnet
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor
    assert vdist.init_from_env("gloo")[:2] == (rank, world)
    g = np.load(REPO / "tests" / "golden" / "g3_predict_29x64x40_c4.npz")
    net = seeded_oracle(4, 0).eval()
    pred = VolSeg2dPredictor.__new__(VolSeg2dPredictor)
    pred.model, pred.num_labels, pred.settings = net, 4, SimpleNamespace(prediction_batch_size=1, cuda_device=0)
    backends = []
    def factory(*a):
        b = OracleBackend(*a); backends.append(b); return b
    pred.backend_factory = factory
    vol = g["vol"][:, :32, :]   # keep the CPU cost small: (29, 32, 40)
    l12, p12 = pred._predict_12_ways_max_probs(vol)
    oh3 = pred._predict_3_ways_one_hot(vol)
    l1, p1 = pred._predict_single_axis(vol)
    pred.result_ranks = "rank0"            # labels / probabilities gathered on rank 0 only (bench.py's mode)
    r0 = pred._predict_3_ways_max_probs(vol)
    assert (r0[0] is None) == (rank != 0)
    pred.result_ranks = "all"
    l3, p3 = pred._predict_3_ways_max_probs(vol)
    if rank == 0:
        assert np.array_equal(r0[0], l3) and np.array_equal(r0[1].view(np.uint16), p3.view(np.uint16))
    # every rank touched only its own contiguous share of every direction
    calls = backends[0].calls
    per_dir = {}
    for d, s0, nb in calls:
        per_dir.setdefault(d, []).append((s0, s0 + nb))
    shares = {d: (min(a for a, _ in v), max(b for _, b in v)) for d, v in per_dir.items()}
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import REPEATED_DIRECTIONS
    assert sorted(shares) == [d for d in range(12) if d not in REPEATED_DIRECTIONS]     # the four exact repeats are not run
    for d in REPEATED_DIRECTIONS:
        shares[d] = (-1, -1)
    # data-parallel gradient averaging on the flat buffer
    m = VolSegUnet(2)
    m.dp_group = dist.group.WORLD
    m._flat_grad = torch.full_like(m._flat, float(rank + 1))
    m._allreduce_grads()
    gmean = m._flat_grad[:5].numpy().copy()
    # the same averaging with the gradients travelling as bf16 (dp_grad_dtype): bounded by bf16's rounding of the inputs
    # and of the sum, i.e. (world + 1) half-ulps of 2^-8 relative to the largest contribution
    gen = torch.Generator().manual_seed(1000 + rank)
    grads = torch.randn(4096, generator=gen) * torch.logspace(-6, 0, 4096)
    m._flat_grad = m._flat.clone(); m._flat_grad[:4096] = grads
    m._allreduce_grads()
    exact = m._flat_grad[:4096].clone()
    m._flat_grad[:4096] = grads
    m.dp_grad_dtype = torch.bfloat16
    m._allreduce_grads()
    low = m._flat_grad[:4096].clone()
    m.dp_grad_dtype = torch.float32
    absmax = grads.abs().clone()
    dist.all_reduce(absmax, op=dist.ReduceOp.MAX)
    np.savez(Path(out_dir) / f"r{rank}.npz", l12=l12, p12=p12, oh3=oh3, l1=l1, p1=p1,
             shares=np.array([shares[d] for d in sorted(shares)]), gmean=gmean, g_exact=exact.numpy(), g_low=low.numpy(),
             g_absmax=absmax.numpy())
    dist.destroy_process_group()


@pytest.mark.slow
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_prediction_and_grad_allreduce(tmp_path, world):
    """World size 2 and 3: 29 / 32 / 40 slices per direction are not divisible by 3 (uneven shards) and neither is the voxel
    count (the key volume is padded for the reduce-scatter-shaped exchange)."""
    from oracle import predictor_numpy as P
    from oracle.unet_resnet34_torch import seeded_oracle
    for attempt in range(2):   # the port is free when picked, not necessarily when the store binds it: one retry
        try:
            mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
            break
        except Exception as e:   # only the bind race is retried: a failed or hung collective in a worker must fail the test
            if attempt or not any(m in str(e) for m in ("Address already in use", "EADDRINUSE", "address already in use")):
                raise
    rs = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    g = np.load(REPO / "tests" / "golden" / "g3_predict_29x64x40_c4.npz")
    vol = g["vol"][:, :32, :]
    net = seeded_oracle(4, 0).eval()
    # slice-at-a-time with the workers' thread count: the CPU oracle's last float bits depend on batch shape and
    # threading, the sharding logic under test does not
    torch.set_num_threads(2)
    ref_l, ref_p = P.predict_12_ways_max_probs(net, vol, batch_size=1)
    ref_oh = P.predict_3_ways_one_hot(net, vol, 4, batch_size=1)
    ref_l1, ref_p1 = P.predict_single_axis(net, vol, 0, batch_size=1)
    for r in rs:   # every rank holds the full merged result, identical to the single-process reference order
        assert np.array_equal(r["l12"], ref_l) and np.array_equal(r["p12"].view(np.uint16), ref_p.view(np.uint16))
        assert np.array_equal(r["oh3"], ref_oh)
        assert np.array_equal(r["l1"], ref_l1) and np.array_equal(r["p1"].view(np.uint16), ref_p1.view(np.uint16))
        assert np.allclose(r["gmean"], (world + 1) / 2)
        # bf16 transport of the gradient all-reduce: each contribution is rounded to bf16 (relative 2^-8) and so is every partial
        # sum of the ring; the mean stays within world x 2^-8 of the largest contribution (the test allows twice that) -
        # against fp32's 2^-24 - and is identical on every rank
        assert np.all(np.abs(r["g_low"] - r["g_exact"]) <= 2 * world * 2.0 ** -8 * r["g_absmax"] + 1e-30)
        assert np.abs(r["g_low"] - r["g_exact"]).max() <= 2.0 ** -6 * np.abs(r["g_exact"]).max()
        assert np.abs(r["g_low"] - r["g_exact"]).max() > 0 and np.array_equal(r["g_low"], rs[0]["g_low"])
    # shares are disjoint and cover every direction's stack
    depths = [29, 32, 40] * 4
    depths[3:6] = [32, 29, 40]; depths[9:12] = [32, 29, 40]   # rot90 / rot270 volumes have shape (Y, Z, X)
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import REPEATED_DIRECTIONS
    for d in range(12):   # contiguous shares in rank order, from 0 to the direction's depth, sizes differing by at most one
        if d in REPEATED_DIRECTIONS:    # not run: the merged result above equals the oracle's full twelve-direction reference all the same
            assert all(int(r["shares"][d][0]) == -1 for r in rs)
            continue
        edges = [0] + [int(r["shares"][d][1]) for r in rs]
        assert all(int(r["shares"][d][0]) == edges[i] for i, r in enumerate(rs)) and edges[-1] == depths[d]
        sizes = np.diff(edges)
        assert sizes.max() - sizes.min() <= 1


# ---- data-parallel VolSeg2dTrainer: shards, shared decisions, early stop on every rank ---------------------------------------
def _trainer_data(n_train=48, n_valid=21, size=32):
    """Training slices whose label is a threshold of the image; validation slices with the INVERTED labels, so the validation
    loss rises as training fits the training rule - an early stop is certain."""
    rng = np.random.default_rng(0)
    field = rng.standard_normal((n_train + n_valid, size, size)).astype(np.float32)
    for ax in (1, 2):
        field = (np.roll(field, 1, ax) + field + np.roll(field, -1, ax)) / 3
    imgs = np.clip(128 + 200 * field, 0, 255).astype(np.uint8)
    masks = (field > 0.05).astype(np.uint8)
    masks[n_train:] = 1 - masks[n_train:]
    return imgs, masks, n_train


def _trainer_settings():
    return SimpleNamespace(starting_lr=1e-5, end_lr=5.0, lr_find_epochs=1, lr_reduce_factor=500, cuda_device=0, patience=1,
                           loss_criterion="DiceLoss", alpha=0.75, beta=0.25, eval_metric="MeanIoU", pct_lr_inc=0.3,
                           plot_lr_graph=False, image_size=32, training_set_proportion=0.8,
                           model={"type": "U_Net", "encoder_name": "resnet34", "encoder_weights": None})


def _trainer_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      VOLSEG_DIST_TIMEOUT_S="180")
    torch.set_num_threads(1)
    import torch.distributed as dist
    from volume_segmantics_amd import dist as vdist
    from volume_segmantics_amd.data.datasets import ArraySliceDataset, make_training_loaders
    from volume_segmantics_amd.model.operations import vol_seg_2d_trainer as T
    assert vdist.init_from_env("gloo")[:2] == (rank, world)

    def tiny_model(device_num, struct):          # stands in for the GPU engine: a small conv net, seeded per rank (rank 0's
        torch.manual_seed(100 + rank)            # weights must win through the broadcast)
        return torch.nn.Sequential(torch.nn.Conv2d(1, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                                   torch.nn.Conv2d(8, struct["classes"], 3, padding=1))
    T.create_model_on_device = tiny_model
    imgs, masks, cut = _trainer_data()
    seen = []

    class Recording(ArraySliceDataset):
        def __getitem__(self, i):
            seen.append(int(i))
            return super().__getitem__(i)
    loaders = make_training_loaders(Recording(imgs[:cut], masks[:cut]), ArraySliceDataset(imgs[cut:], masks[cut:]), 4, rank, world, seed=7)
    tr = T.VolSeg2dTrainer(None, None, {"bg": 0, "fg": 1}, _trainer_settings(), loaders=loaders)
    assert (tr.rank, tr.world) == (rank, world) and len(tr.training_loader) == 48 // (4 * world)
    out = Path(out_dir) / "dp.pytorch"
    tr.train_model(out, 12, 1, create=True, frozen=False)
    flat = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()] + [b.detach().reshape(-1).float() for b in tr.model.buffers()])
    first_epoch = seen[:len(tr.training_loader) * 4]        # the LR finder's epoch (epoch 0 of the sampler)
    np.savez(Path(out_dir) / f"t{rank}.npz", flat=flat.numpy(), epochs=len(tr.avg_valid_losses), valid=np.array(tr.avg_valid_losses),
             train=np.array(tr.avg_train_losses), first_epoch=np.array(first_epoch), lr=tr.optimizer.param_groups[0]["lr"])
    dist.destroy_process_group()


def _spawn(fn, world, tmp_path):
    for attempt in range(2):   # the port is free when picked, not necessarily when the store binds it: one retry
        try:
            mp.spawn(fn, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
            return
        except Exception as e:   # only the bind race is retried
            if attempt or not any(m in str(e) for m in ("Address already in use", "EADDRINUSE", "address already in use")):
                raise


@pytest.mark.slow
@pytest.mark.parametrize("world", [2, 3])
def test_data_parallel_trainer_shards_decides_and_stops_together(tmp_path, world):
    """train_model at world size 2 and 3 (LR finder -> one-cycle training -> early stop -> reload): the ranks' shards of an
    epoch are disjoint and cover the global batches, every rank sees the same epoch losses, stops in the same epoch (no rank
    is left waiting in a collective) and ends with bit-identical weights."""
    _spawn(_trainer_worker, world, tmp_path)
    rs = [np.load(tmp_path / f"t{r}.npz") for r in range(world)]
    assert (tmp_path / "dp.pytorch").exists()
    for r in rs[1:]:
        assert int(r["epochs"]) == int(rs[0]["epochs"]) and np.array_equal(r["valid"], rs[0]["valid"]) and np.array_equal(r["train"], rs[0]["train"])
        assert np.array_equal(r["flat"], rs[0]["flat"])                  # same weights (reloaded best checkpoint) on every rank
    assert 2 <= int(rs[0]["epochs"]) < 12                                 # the inverted validation labels forced an early stop
    shards = [set(r["first_epoch"].tolist()) for r in rs]
    n_used = (48 // (4 * world)) * 4 * world
    assert all(len(s) == n_used // world for s in shards)
    assert len(set().union(*shards)) == n_used                            # disjoint, and together the epoch's global batches


def test_sharded_batch_sampler_partitions_global_batches():
    from volume_segmantics_amd.data.datasets import ShardedBatchSampler
    for world in (1, 2, 3):
        per_rank = [list(ShardedBatchSampler(50, 4, r, world, shuffle=True, drop_last=True, seed=3)) for r in range(world)]
        assert len({len(p) for p in per_rank}) == 1 and len(per_rank[0]) == 50 // (4 * world)
        ref = list(ShardedBatchSampler(50, 4 * world, 0, 1, shuffle=True, drop_last=True, seed=3))   # the one-process loader
        for step, glob in enumerate(ref):
            assert sum((p[step] for p in per_rank), []) == glob          # the ranks' shards, in rank order, ARE the global batch
        val = [list(ShardedBatchSampler(21, 4, r, world, shuffle=False, drop_last=False)) for r in range(world)]
        assert len({len(v) for v in val}) == 1                            # the same number of iterations on every rank
        assert sorted(sum((sum(v, []) for v in val), [])) == list(range(21))
    s = ShardedBatchSampler(50, 4, 0, 2, seed=3)
    a = list(s); s.set_epoch(1); b = list(s)
    assert a != b and len(a) == len(b)
