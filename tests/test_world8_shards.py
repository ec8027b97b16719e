"""CPU tests of the N = 8 arithmetic - the world size the driver's scaling run uses (one rank per GPU of an 8-GPU node) and
that no box available to the build has: shard ranges of BASELINE configs[2]'s 6 144 slices, the padded key volume of the
reduce-scatter-shaped exchange, ShardedBatchSampler at a global batch of 32 x 8, the gradient buckets' boundaries in the flat
buffer, and an actual 8-process gloo run of the sharded 12-direction prediction (a small stand-in network: the sharding /
exchange logic under test does not depend on what computes the logits).  Reference: the single shuffled loader of
data/dataloaders.py:42-49 and the direction order of vol_seg_2d_predictor.py:67-116; SURVEY.md section 8(e)."""
import ctypes as C
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
WORLD = 8


def test_slice_shards_of_the_512_cube_twelve_directions_at_world_8():
    from volume_segmantics_amd import dist as vdist
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import REPEATED_DIRECTIONS, direction_views
    # configs[2]: 512^3 -> every direction's stack is 512 slices: 64 per rank, 6 144 in all (4 096 with the repeats left out)
    total = 0
    for d in range(12):
        edges = [vdist.shard_range(512, r, WORLD) for r in range(WORLD)]
        assert edges == [(64 * r, 64 * (r + 1)) for r in range(WORLD)]
        total += sum(hi - lo for lo, hi in edges)
    assert total == 6144 and 6144 - 512 * len(REPEATED_DIRECTIONS) == 4096
    # a volume no dimension of which divides by 8: contiguous, disjoint, covering shares, sizes within one of each other
    vol = np.zeros((45, 61, 13), np.uint8)
    for view in direction_views(vol, 12):
        depth = view.shape[0]
        shares = [vdist.shard_range(depth, r, WORLD) for r in range(WORLD)]
        assert shares[0][0] == 0 and shares[-1][1] == depth and all(a[1] == b[0] for a, b in zip(shares, shares[1:]))
        sizes = [hi - lo for lo, hi in shares]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == depth
    # fewer slices than ranks: the surplus ranks get an empty range (and run no batch)
    assert [vdist.shard_range(5, r, WORLD) for r in range(WORLD)] == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 5), (5, 5), (5, 5)]
    # the key volume is padded to a multiple of the world size so that reduce-scatter hands every rank an equal shard
    assert vdist.padded_len(512 ** 3, WORLD) == 512 ** 3 and 512 ** 3 // WORLD * 4 == 67108864          # 64 MiB of keys per rank
    n = 45 * 61 * 13
    assert vdist.padded_len(n, WORLD) % WORLD == 0 and 0 <= vdist.padded_len(n, WORLD) - n < WORLD


def test_sharded_batch_sampler_at_global_batch_32_times_8():
    from volume_segmantics_amd.data.datasets import ShardedBatchSampler
    n, per_rank = 6144, 32
    samplers = [ShardedBatchSampler(n, per_rank, r, WORLD, shuffle=True, drop_last=True, seed=11) for r in range(WORLD)]
    assert all(len(s) == n // (per_rank * WORLD) == 24 for s in samplers)
    for epoch in (0, 1):
        for s in samplers:
            s.set_epoch(epoch)
        order = torch.randperm(n, generator=torch.Generator().manual_seed(11 + epoch)).tolist()      # the ONE shared permutation
        for b, shards in enumerate(zip(*samplers)):
            assert all(len(x) == per_rank for x in shards)
            flat = [i for x in shards for i in x]
            assert flat == order[b * 256:(b + 1) * 256]          # rank r holds items [32 r, 32 r + 32) of global batch b
    seen0 = [i for batch in samplers[0] for i in batch]
    samplers[0].set_epoch(0)
    assert seen0 != [i for batch in samplers[0] for i in batch]      # a new permutation per epoch
    # validation (no shuffle, keep the partial last global batch): 1 000 = 3 x 256 + 232 -> shares of 29; 775 = 3 x 256 + 7 -> one
    # rank's share is empty and the trainer's loop skips it (no collective may sit in that loop: tests/dp_trainer_worker.py)
    for n_valid, last in ((1000, [29] * 8), (775, [1] * 7 + [0])):
        vs = [list(ShardedBatchSampler(n_valid, per_rank, r, WORLD, shuffle=False, drop_last=False)) for r in range(WORLD)]
        assert all(len(v) == 4 for v in vs) and [len(v[-1]) for v in vs] == last
        assert sorted(i for v in vs for batch in v for i in batch) == list(range(n_valid))


def test_gradient_buckets_partition_the_flat_buffer():
    """engine.VolSegUnet._bucket_plan: the four all-reduce buckets (decoder + head, layer4, layer3, stem + layer1 + layer2) in the
    order backward completes them - contiguous slices of the flat gradient buffer from its end to its start, nothing left out,
    nothing twice.  Their sizes are what DESIGN.md section 6 prices the per-step communication with."""
    from volume_segmantics_amd import _lib as L
    from volume_segmantics_amd.engine import VolSegUnet
    m = VolSegUnet(2)                      # host object: the plan builder needs no GPU
    h = C.c_void_p()
    L.check(L.lib.vs_unet_create(C.byref(h), L.VS_BF16, 2, 32, 256, 256))
    try:
        plan = m._bucket_plan(h)
        total = L.lib.vs_unet_param_elems(2)
        assert len(plan) == 4 and plan[0][3] == total and plan[-1][2] == 0 and plan[0][1] == L.lib.vs_unet_num_units(h) and plan[-1][0] == 0
        assert all(a[2] == b[3] and a[0] == b[1] for a, b in zip(plan, plan[1:]))      # contiguous in parameters and in units
        sizes = [b - a for _, _, a, b in plan]
        assert all(s > 0 for s in sizes) and sum(sizes) == total == 24430242
        names = L.unit_names(h)
        assert names[plan[0][0]].startswith("decoder.blocks.0") and names[plan[1][0]].startswith("encoder.layer4.0.conv1")
        assert names[plan[2][0]].startswith("encoder.layer3.0.conv1")
        mb = [round(s * 2 / 1e6, 2) for s in sizes]          # bf16 transport
        print(f"[world8] gradient buckets (bf16 MB, in completion order): {mb}")
        assert mb == [6.3, 26.23, 13.64, 2.68], mb            # decoder + head, layer4, layer3, the rest = 48.86 MB per step
    finally:
        L.lib.vs_unet_destroy(h)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tiny_net():
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(1, 6, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv2d(6, 3, 3, padding=1))
    return net.eval()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      VOLSEG_DIST_TIMEOUT_S="240")
    torch.set_num_threads(1)
    import torch.distributed as dist
    from cpu_backend import OracleBackend
    from volume_segmantics_amd import dist as vdist
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor
    assert vdist.init_from_env("gloo")[:2] == (rank, world)
    pred = VolSeg2dPredictor.__new__(VolSeg2dPredictor)
    pred.model, pred.num_labels, pred.settings = _tiny_net(), 3, SimpleNamespace(prediction_batch_size=2, cuda_device=0)
    backends = []

    def factory(*a):
        b = OracleBackend(*a); backends.append(b); return b
    pred.backend_factory = factory
    vol = np.random.default_rng(9).integers(0, 255, (21, 37, 12)).astype(np.uint8)       # 12 < 16 slices on one axis: some ranks idle there
    l12, p12 = pred._predict_12_ways_max_probs(vol)
    oh12 = pred._predict_12_ways_one_hot(vol)
    touched = sorted({(d, s) for d, s0, nb in backends[0].calls for s in range(s0, s0 + nb)})
    sums = vdist.allreduce_sums([float(rank), 1.0])
    mean = vdist.mean_scalar(float(rank))
    np.savez(Path(out_dir) / f"r{rank}.npz", l12=l12, p12=p12, oh12=oh12, touched=np.array(touched), sums=np.array(sums), mean=mean)
    dist.destroy_process_group()


@pytest.mark.slow
def test_eight_rank_sharded_twelve_direction_prediction_over_gloo(tmp_path):
    """Eight processes (gloo): every rank predicts its contiguous eighth of each direction's stack, the packed keys meet in the one
    MAX exchange (padded key volume, per-rank unpack, all-gather), the votes in one SUM - and every rank returns the volume the
    single-process reference order produces, bit for bit."""
    sys.path.insert(0, str(REPO / "tests"))
    from oracle import predictor_numpy as P
    from volume_segmantics_amd import dist as vdist
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import REPEATED_DIRECTIONS, direction_views
    for attempt in range(2):
        try:
            mp.spawn(_worker, args=(WORLD, _free_port(), str(tmp_path)), nprocs=WORLD, join=True)
            break
        except Exception as e:
            if attempt or not any(m in str(e) for m in ("Address already in use", "EADDRINUSE", "address already in use")):
                raise
    rs = [np.load(tmp_path / f"r{r}.npz") for r in range(WORLD)]
    vol = np.random.default_rng(9).integers(0, 255, (21, 37, 12)).astype(np.uint8)
    torch.set_num_threads(1)
    net = _tiny_net()
    ref_l, ref_p = P.predict_12_ways_max_probs(net, vol, batch_size=1)
    ref_oh = P.predict_12_ways_one_hot(net, vol, 3, batch_size=1)
    depths = [v.shape[0] for v in direction_views(vol, 12)]
    for rank, r in enumerate(rs):
        assert np.array_equal(r["l12"], ref_l) and np.array_equal(r["p12"].view(np.uint16), ref_p.view(np.uint16)), rank
        assert np.array_equal(r["oh12"], ref_oh), rank
        mine = {(d, s) for d in range(12) if d not in REPEATED_DIRECTIONS for s in range(*vdist.shard_range(depths[d], rank, WORLD))}
        assert {tuple(t) for t in r["touched"].reshape(-1, 2).tolist()} == mine, rank
        assert r["sums"].tolist() == [28.0, 8.0] and float(r["mean"]) == 3.5
