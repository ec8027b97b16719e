"""Worker of tests/test_hip_dp_rehearsal.py::test_two_rank_trainer...: VolSeg2dTrainer.train_model on the HIP engine with two
ranks sharing GPU 0 over gloo (everything but the transport of the real one-rank-per-GPU RCCL run): LR finder, one-cycle
training, early stop, checkpoint reload - both ranks must take the same decisions and end with bit-identical parameters."""
import os
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["VOLSEG_DP_SINGLE_DEVICE"] = "1"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from test_dist_gloo import _trainer_data, _trainer_settings
    from volume_segmantics_amd.data.datasets import ArraySliceDataset, make_training_loaders
    from volume_segmantics_amd.engine import VolSegUnet
    from volume_segmantics_amd.model.operations.vol_seg_2d_trainer import VolSeg2dTrainer
    sync = len(sys.argv) > 2 and sys.argv[2] == "sync"
    # sync: 17 validation slices = two global batches of 8 + ONE slice - rank 1's share of the last global batch is empty, so the
    # ranks run different numbers of validation batches: nothing in the validation loop may be a collective (the global Dice of
    # `sync_batchnorm` is a training construct; HipDiceLoss drops the group under torch.no_grad())
    imgs, masks, cut = _trainer_data(n_valid=17 if sync else 21, size=64)
    loaders = make_training_loaders(ArraySliceDataset(imgs[:cut], masks[:cut]), ArraySliceDataset(imgs[cut:], masks[cut:]), 4,
                                    rank, world, seed=7)
    settings = _trainer_settings()
    settings.precision = "bf16"
    if sync:      # BatchNorm statistics and the Dice loss of the GLOBAL batch (settings key sync_batchnorm)
        settings.sync_batchnorm = True
    torch.manual_seed(1000 + rank)        # different initial weights per rank: rank 0's must win through the broadcast
    tr = VolSeg2dTrainer(None, None, {"bg": 0, "fg": 1}, settings, loaders=loaders)
    out = Path(sys.argv[1]) / "dp_gpu.pytorch"
    tr.train_model(out, 10, 1, create=True, frozen=True)
    assert isinstance(tr.model, VolSegUnet) and tr.model.device.index == 0 and tr.model.precision == "bf16"
    if sync:
        from volume_segmantics_amd.data.losses import HipDiceLoss
        assert tr.model.sync_bn and isinstance(tr.loss_criterion, HipDiceLoss) and tr.loss_criterion.global_group is not None
        hooked = [p for p in tr.model._plans.values() if p.get("sync")]
        assert hooked and all(p["sync"][2]["error"] is None for p in hooked)
    torch.cuda.synchronize()
    mine = torch.cat([tr.model._flat, tr.model._bnstate]).cpu()
    other = mine.clone()
    dist.broadcast(other, 0)
    assert torch.equal(mine, other), "ranks ended with different parameters"
    ep = torch.tensor([len(tr.avg_valid_losses)])
    ep0 = ep.clone()
    dist.broadcast(ep0, 0)
    assert int(ep) == int(ep0) and 2 <= int(ep) < 10, (int(ep), int(ep0))
    assert np.all(np.isfinite(tr.avg_valid_losses)) and out.exists()
    dist.barrier()
    if rank == 0:
        print(f"DP_TRAINER_OK epochs={int(ep)}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
