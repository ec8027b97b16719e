"""-m gpu: the N > 1 training path on the real kernels - two ranks share GPU 0 and talk over gloo (the driver's multi-GPU run
uses one rank per GPU over RCCL; this pins everything except the transport): bucketed backward + all-reduce with the
optimiser step inside backward vs after it, bit-identical and rank-consistent (tests/dp_rehearsal_worker.py)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def _torchrun(worker, *args):
    import socket
    repo = Path(__file__).resolve().parents[1]
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(repo / "tests" / worker), *args]
    return subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("topology,encoder", [("unet", "resnet34"), ("unetplusplus", "resnet50"), ("fpn", "resnet34"), ("unet", "efficientnet-b3"), ("unet", "timm-resnest50d")])
def test_two_rank_data_parallel_fused_step_matches_plain_step(topology, encoder):
    """U-Net / ResNet-34 (the headline), U-Net++ / ResNet-50 (BASELINE configs[3]'s data-parallel training) and FPN (GroupNorm,
    biased laterals and a Dropout2d whose mask differs per rank)."""
    r = _torchrun("dp_rehearsal_worker.py", topology, encoder)
    assert r.returncode == 0 and "DP_REHEARSAL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("mode", ["per_rank", "sync"])
def test_two_rank_trainer_runs_through_an_early_stop_with_identical_weights(tmp_path, mode):
    """VolSeg2dTrainer.train_model at world size 2 on the HIP engine (tests/dp_trainer_worker.py): disjoint shards of each
    global batch, the LR finder's learning rate and the early stop decided on all-reduced losses, rank 0 writes the
    checkpoint, both ranks reload it and end with bit-identical parameters and running statistics.  `sync`: the same run with
    `sync_batchnorm: true` - SyncBatchNorm + the Dice of the global batch through the trainer's own settings surface."""
    r = _torchrun("dp_trainer_worker.py", str(tmp_path), *(["sync"] if mode == "sync" else []))
    assert r.returncode == 0 and "DP_TRAINER_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("topology,encoder", [("unet", "resnet34"), ("unetplusplus", "resnet50")])
def test_two_rank_sync_batchnorm_equals_one_rank_at_twice_the_batch(topology, encoder):
    """SyncBatchNorm (`model.sync_bn`, vs_unet_set_stats_hook): two ranks with half the batch each reproduce ONE process running the
    whole batch - training logits and running statistics bit for bit (the statistics are integer sums), gradients to bf16 noise
    (tests/dp_syncbn_worker.py).  The headline network and BASELINE configs[3]'s (U-Net++ / ResNet-50 data-parallel training)."""
    r = _torchrun("dp_syncbn_worker.py", topology, encoder)
    assert r.returncode == 0 and "DP_SYNCBN_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
