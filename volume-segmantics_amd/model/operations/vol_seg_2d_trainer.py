"""VolSeg2dTrainer - LR finder, AdamW + OneCycleLR, frozen / unfrozen phases, early-stopping checkpoints
(reference: volume_segmantics/model/operations/vol_seg_2d_trainer.py:35-535).

The training step is the reference's ``_train_one_batch`` (:419-432); the model behind it is the HIP engine and the
optimiser is its fused AdamW (same update rule, one kernel over the flat parameter buffer).

Data parallel (torchrun, one process per GPU): the reference has ONE shuffled loader and ONE decision maker
(data/dataloaders.py:42-57; vol_seg_2d_trainer.py:189-205,265-274).  Here every rank draws the same permutation and takes
its own disjoint shard of every global batch (ShardedBatchSampler; ``batch_size`` is per rank), the flat gradient is
all-reduced over RCCL inside ``loss.backward()``, and every number a decision hangs on - the LR finder's smoothed loss,
the epoch's validation loss - is the all-reduced value, so all ranks pick the same learning rate and stop in the same
epoch; rank 0 writes the checkpoint, a barrier follows, every rank reloads it.  BatchNorm statistics stay per rank (as
torch's DistributedDataParallel without SyncBatchNorm)."""
from __future__ import annotations

import csv
import logging
import math
import os
import sys
import time
from pathlib import Path
from types import SimpleNamespace
from typing import Union

import numpy as np
import torch
import torch.nn as nn

from ... import dist as vdist
from ...checkpoint_compat import reference_pickle_enum
from ...data.datasets import get_2d_training_dataloaders
from ...data.losses import BCEDiceLoss, DiceCoefficient, DiceLoss, GeneralizedDiceLoss, HipDiceLoss, HipSegLoss, MeanIoU
from ...engine import FusedAdamW, VolSegUnet
from ...utilities import base_data_utils as utils
from ...utilities import config as cfg
from ...utilities.early_stopping import EarlyStopping
from ..model_2d import create_model_on_device


class VolSeg2dTrainer:
    def __init__(self, image_dir_path: Path, label_dir_path: Path, labels: Union[int, dict], settings: SimpleNamespace,
                 loaders=None):
        """``loaders`` = (training_loader, validation_loader) bypasses the PNG directories (synthetic data, tests)."""
        # torchrun environment -> process group (RCCL); a single process, or a group the caller has set up already, is left alone
        self.rank, self.world, local_rank = vdist.init_from_env()
        self.rank, self.world = vdist.world()
        if loaders is None:
            loaders = get_2d_training_dataloaders(image_dir_path, label_dir_path, settings, self.rank, self.world)
        self.training_loader, self.validation_loader = loaders
        self.label_no = labels if isinstance(labels, int) else len(labels)
        for loader in loaders:      # device-resident feeds check their label range themselves (the reference's F.one_hot raises per batch)
            if hasattr(loader, "max_label"):
                loader.num_labels = self.label_no
        self.codes = labels if isinstance(labels, dict) else {}
        self.settings = settings
        self.starting_lr, self.end_lr = float(settings.starting_lr), float(settings.end_lr)
        self.log_lr_ratio = self._calculate_log_lr_ratio()
        self.lr_find_epochs = settings.lr_find_epochs
        self.lr_reduce_factor = settings.lr_reduce_factor
        # one GPU per rank (LOCAL_RANK); VOLSEG_DP_SINGLE_DEVICE=1 keeps every rank on settings.cuda_device (rehearsals on one GPU)
        if self.world > 1 and not os.environ.get("VOLSEG_DP_SINGLE_DEVICE"):
            if "LOCAL_RANK" in os.environ:
                self.model_device_num = local_rank
            elif torch.cuda.is_available():
                # the caller initialised the process group itself (no torchrun environment): its rank's current device,
                # never "every rank on cuda:0"
                self.model_device_num = torch.cuda.current_device()
            else:
                raise RuntimeError("data-parallel training without LOCAL_RANK: select the rank's device (torch.cuda.set_device) "
                                   "before constructing VolSeg2dTrainer, or launch with torchrun")
        else:
            self.model_device_num = int(settings.cuda_device)
        self.patience = settings.patience
        self.loss_criterion = self._get_loss_criterion()
        self.eval_metric = self._get_eval_metric()
        self.model_struc_dict = self._get_model_struc_dict(settings)
        self.avg_train_losses, self.avg_valid_losses, self.avg_eval_scores = [], [], []

    # ---- construction helpers -------------------------------------------------------------------------------
    def _get_model_struc_dict(self, settings):
        d = settings.model     # (the reference fills in the caller's dict too, vol_seg_2d_trainer.py:78-84)
        d["type"] = utils.get_model_type(settings)
        d["in_channels"] = cfg.MODEL_INPUT_CHANNELS
        d["classes"] = self.label_no
        if getattr(settings, "precision", None):
            d["precision"] = settings.precision
        return d

    def _calculate_log_lr_ratio(self):
        return math.log(self.end_lr / self.starting_lr)

    def _get_loss_criterion(self):
        name = self.settings.loss_criterion
        gpu = torch.cuda.is_available()     # fused HIP sweeps on device tensors; the torch restatements otherwise
        if name == "BCEDiceLoss":
            return HipSegLoss(name, self.settings.alpha, self.settings.beta) if gpu else BCEDiceLoss(self.settings.alpha, self.settings.beta)
        if name == "DiceLoss":
            if gpu and self.world > 1 and bool(getattr(self.settings, "sync_batchnorm", False)):
                import torch.distributed as dist
                return HipDiceLoss(global_group=dist.group.WORLD)     # with SyncBatchNorm: the Dice of the global batch as well
            return HipDiceLoss() if gpu else DiceLoss(normalization="none")
        if name == "BCELoss":
            return HipSegLoss(name) if gpu else nn.BCEWithLogitsLoss()
        if name == "CrossEntropyLoss":
            return HipSegLoss(name) if gpu else nn.CrossEntropyLoss()
        if name == "GeneralizedDiceLoss":
            return HipSegLoss(name) if gpu else GeneralizedDiceLoss()
        logging.error("No loss criterion specified, exiting")
        sys.exit(1)

    def _get_eval_metric(self):
        if self.settings.eval_metric == "MeanIoU":
            return MeanIoU()
        if self.settings.eval_metric == "DiceCoefficient":
            return DiceCoefficient()
        logging.error("No evaluation metric specified, exiting")
        sys.exit(1)

    def _create_model_and_optimiser(self, learning_rate, frozen=False):
        logging.info(f"Setting up the model on device {self.settings.cuda_device}.")
        self.model = create_model_on_device(self.model_device_num, self.model_struc_dict)
        if self.world > 1:   # every rank starts from rank 0's weights
            import torch.distributed as dist
            if isinstance(self.model, VolSegUnet):
                dist.broadcast(self.model._flat, 0)
                dist.broadcast(self.model._bnstate, 0)
                self.model.dp_group = dist.group.WORLD
                # `sync_batchnorm: true`: BatchNorm statistics of the GLOBAL batch - N ranks reproduce the single-process batch the
                # reference trains with (engine.VolSegUnet._sync_hook); default: per-rank statistics (torch DDP's default)
                self.model.sync_bn = bool(getattr(self.settings, "sync_batchnorm", False))
                self.model.dropout_seed += 1000003 * self.rank      # Dropout2d (FPN): every rank draws its own masks
            else:
                vdist.broadcast_module(self.model)
        if frozen:
            self._freeze_model()
        logging.info(f"Model has {self._count_trainable_parameters()} trainable parameters, "
                     f"{self._count_parameters()} total parameters.")
        self.optimizer = self._create_optimizer(learning_rate)

    def _freeze_model(self):
        """Reference predicate (:102-108): names containing both "encoder" and "conv"."""
        for name, param in self.model.named_parameters():
            if "encoder" in name and "conv" in name and param.requires_grad:
                param.requires_grad = False

    def _unfreeze_model(self):
        for name, param in self.model.named_parameters():
            if "encoder" in name and "conv" in name and not param.requires_grad:
                param.requires_grad = True

    def _count_trainable_parameters(self) -> int:
        return sum(p.numel() for p in self.model.parameters() if p.requires_grad)

    def _count_parameters(self) -> int:
        return sum(p.numel() for p in self.model.parameters())

    def _create_optimizer(self, learning_rate):
        if isinstance(self.model, VolSegUnet):
            # one step() per backward() with the lr set beforehand (_train_one_batch, LR finder): the update can ride on
            # the backward pass
            return FusedAdamW(self.model, lr=learning_rate, fuse_step_into_backward=True)
        return torch.optim.AdamW(self.model.parameters(), lr=learning_rate)

    def _lr_exp_stepper(self, x):
        return math.exp(x * self.log_lr_ratio / (self.lr_find_epochs * len(self.training_loader)))

    def _create_exponential_lr_scheduler(self):
        return torch.optim.lr_scheduler.LambdaLR(self.optimizer, self._lr_exp_stepper)

    def _create_oc_lr_scheduler(self, num_epochs, lr_to_use):
        return torch.optim.lr_scheduler.OneCycleLR(self.optimizer, max_lr=lr_to_use, steps_per_epoch=len(self.training_loader),
                                                   epochs=num_epochs, pct_start=self.settings.pct_lr_inc)

    def _create_early_stopping(self, output_path, patience, best_score=None):
        struct = dict(self.model_struc_dict)
        struct["type"] = reference_pickle_enum(struct["type"])  # checkpoint readable by the reference
        es = EarlyStopping(patience=patience, verbose=True, path=output_path, model_dict=struct, best_score=best_score)
        es.write = self.rank == 0
        return es

    # ---- the hot loop ---------------------------------------------------------------------------------------
    def _loss(self, output, targets):
        if isinstance(self.loss_criterion, (HipDiceLoss, HipSegLoss)):
            return self.loss_criterion(output, targets)   # reads the uint8 one-hot directly (cross entropy: the position of the 1)
        if self.settings.loss_criterion == "CrossEntropyLoss":
            return self.loss_criterion(output, torch.argmax(targets, dim=1))
        return self.loss_criterion(output, targets.float())

    def _device(self):
        dev = getattr(self.model, "device", None)
        return dev if dev is not None else next(self.model.parameters()).device

    def _train_one_batch(self, lr_scheduler, batch):
        if self._aug_rng is None:     # draws of the device-side augmentation: one stream per rank
            self._aug_rng = np.random.default_rng([int(getattr(self.settings, "augment_seed", 0)), self.rank])
        inputs, targets = utils.prepare_training_batch(batch, self._device(), self.label_no, augment_rng=self._aug_rng)
        # `step_mode: graph` replays the whole step (zero_grad .. optimizer.step) as recorded hipGraphs - same kernels, same order, same
        # bits, 0.4 ms of host time per step instead of ~1.8; the default enqueues it call by call, which every box of rounds 3 and 4
        # ran 2 - 4 % faster (4.65 vs 4.82 ms; the replay's range boundaries cost more than a host that keeps up) - for a slow or busy
        # host the replay is the one to pick (bench.py times both and keeps the faster)
        fuse = getattr(self.model, "can_fuse_step", None) if getattr(self.settings, "step_mode", "eager") == "graph" else None
        if fuse is not None and isinstance(self.loss_criterion, HipDiceLoss) and fuse(self.optimizer, inputs, targets):
            loss = self.model.fused_train_step(inputs, targets, self.optimizer, eps=self.loss_criterion.epsilon)
            lr_scheduler.step()
            return loss
        self.optimizer.zero_grad()
        output = self.model(inputs)
        loss = self._loss(output, targets)
        loss.backward()
        if self.world > 1 and not isinstance(self.model, VolSegUnet):
            vdist.average_gradients(self.model)     # (the engine all-reduces its flat gradient inside backward)
        self.optimizer.step()
        lr_scheduler.step()
        return loss

    def train_model(self, output_path: Path, num_epochs: int, patience: int, create: bool = True, frozen: bool = False) -> None:
        train_losses, valid_losses, eval_scores = [], [], []
        if create:
            self._create_model_and_optimiser(self.starting_lr, frozen=frozen)
            lr_to_use = self._run_lr_finder()
            self._create_model_and_optimiser(lr_to_use, frozen=frozen)
            early_stopping = self._create_early_stopping(output_path, patience)
        else:
            self.starting_lr /= self.lr_reduce_factor
            self.end_lr /= self.lr_reduce_factor
            self.log_lr_ratio = self._calculate_log_lr_ratio()
            self._load_in_model_and_optimizer(self.starting_lr, output_path, frozen=frozen, optimizer=False)
            lr_to_use = self._run_lr_finder()
            min_loss = self._load_in_model_and_optimizer(self.starting_lr, output_path, frozen=frozen, optimizer=False)
            early_stopping = self._create_early_stopping(output_path, patience, best_score=-min_loss)
        lr_scheduler = self._create_oc_lr_scheduler(num_epochs, lr_to_use)
        for epoch in range(1, num_epochs + 1):
            self.model.train()
            tic = time.perf_counter()
            self._set_epoch(self._epochs_run)
            for batch in self.training_loader:
                # (the reference calls loss.item() here, one host-device sync per step; the values are only used as the epoch's
                # average, so they stay on the device until the epoch ends and the host keeps enqueueing ahead of the GPU)
                train_losses.append(self._train_one_batch(lr_scheduler, batch).detach())
            train_losses = torch.stack(train_losses).double().cpu().tolist() if train_losses else []
            self.model.eval()
            with torch.no_grad():
                for batch in self.validation_loader:
                    if batch is None:      # this rank's share of the last, partial global batch is empty
                        continue
                    inputs, targets = utils.prepare_training_batch(batch, self._device(), self.label_no)
                    output = self.model(inputs)
                    valid_losses.append(self._loss(output, targets).item())
                    metric = getattr(self.eval_metric, "from_logits", None)   # MeanIoU: softmax + metric in one HIP sweep
                    eval_scores.append(float(metric(output, targets) if metric else self.eval_metric(torch.softmax(output, dim=1), targets)))
            # epoch averages over ALL ranks' batches: every rank holds the same numbers, so the early-stopping decision below
            # is the same everywhere (the reference has one process: np.average over its own lists)
            sums = vdist.allreduce_sums([sum(train_losses), len(train_losses), sum(valid_losses), len(valid_losses),
                                         sum(eval_scores), len(eval_scores)])
            self.avg_train_losses.append(sums[0] / max(sums[1], 1))
            self.avg_valid_losses.append(sums[2] / max(sums[3], 1))
            self.avg_eval_scores.append(sums[4] / max(sums[5], 1))
            logging.info(f"Epoch {epoch}. Training loss: {self.avg_train_losses[-1]}, Validation Loss: "
                         f"{self.avg_valid_losses[-1]}. {self.settings.eval_metric}: {self.avg_eval_scores[-1]}")
            logging.info(f"Time taken for epoch {epoch}: {time.perf_counter() - tic:0.2f} seconds")
            train_losses, valid_losses, eval_scores = [], [], []
            early_stopping(self.avg_valid_losses[-1], self.model, self.optimizer, self.codes)   # rank 0 writes (EarlyStopping.write)
            vdist.barrier()                     # the file is complete before any rank goes on (and may read it)
            if early_stopping.early_stop:
                logging.info("Early stopping")
                break
        self._load_in_weights(output_path)      # every rank continues from the best checkpoint

    def _load_in_model_and_optimizer(self, learning_rate, output_path, frozen=False, optimizer=False):
        self._create_model_and_optimiser(learning_rate, frozen=frozen)
        return self._load_in_weights(output_path, optimizer=optimizer)

    def _load_in_weights(self, output_path, optimizer=False, gpu=True):
        from ...checkpoint_compat import install_reference_aliases
        install_reference_aliases()
        model_dict = torch.load(output_path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(model_dict["model_state_dict"])
        if optimizer:
            self.optimizer.load_state_dict(model_dict["optimizer_state_dict"])
        return model_dict.get("loss_val", np.inf)

    # ---- LR finder (:298-383) -------------------------------------------------------------------------------
    def _run_lr_finder(self):
        lr_scheduler = self._create_exponential_lr_scheduler()
        lr_find_loss, lr_find_lr = self._lr_finder(lr_scheduler)
        lr_to_use = self._find_lr_from_graph(lr_find_loss, lr_find_lr)
        logging.info(f"LR to use {lr_to_use}")
        return lr_to_use

    def _set_epoch(self, epoch: int) -> None:
        self._epochs_run = epoch + 1
        sampler = getattr(self.training_loader, "batch_sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)           # a new shared permutation (ShardedBatchSampler)

    _epochs_run = 0
    _aug_rng = None

    def _lr_finder(self, lr_scheduler, smoothing=0.05):
        losses, lrs, iters = [], [], 0
        self.model.train()
        for _ in range(self.lr_find_epochs):
            self._set_epoch(self._epochs_run)
            for batch in self.training_loader:
                loss = self._train_one_batch(lr_scheduler, batch).detach()
                if self.world > 1:   # the mean over the ranks' shards: the same curve, break point and learning rate everywhere
                    loss = torch.tensor(vdist.mean_scalar(float(loss)))
                lrs.append(self.optimizer.param_groups[0]["lr"])
                if iters:
                    loss = smoothing * loss + (1 - smoothing) * losses[-1]
                losses.append(loss)
                if loss > 1 and iters > len(self.training_loader) // 1.333:
                    break
                iters += 1
        return losses, lrs

    @staticmethod
    def _find_lr_from_graph(lr_find_loss, lr_find_lr) -> float:
        """lr at the steepest descent of the smoothed loss curve / 3; 0.00075 when the curve never falls."""
        losses = np.array([float(l.detach().cpu()) if isinstance(l, torch.Tensor) else float(l) for l in lr_find_loss])
        try:
            gradients = np.gradient(losses)
            if gradients.min() >= 0:
                logging.info("Minimum gradient was positive, returning default value instead.")
                return cfg.DEFAULT_MIN_LR
            idx = gradients.argmin()
        except Exception as e:
            logging.info(f"Failed to compute gradients, returning default value. {e}")
            return cfg.DEFAULT_MIN_LR
        return lr_find_lr[idx] / cfg.LR_DIVISOR

    # ---- reporting ------------------------------------------------------------------------------------------
    def output_loss_fig(self, model_out_path: Path) -> None:
        """Loss-curve PNG + ``*_train_stats.csv`` next to the model (:434-483)."""
        model_out_path = Path(model_out_path)
        out_dir = model_out_path.parent
        with open(out_dir / f"{model_out_path.stem}_train_stats.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(("Epoch", "Train Loss", "Valid Loss", "Eval Score"))
            for row in zip(range(len(self.avg_train_losses)), self.avg_train_losses, self.avg_valid_losses, self.avg_eval_scores):
                w.writerow(row)
        try:
            import matplotlib
            matplotlib.use("Agg")
            from matplotlib import pyplot as plt
        except ImportError:
            return
        fig = plt.figure(figsize=(10, 8))
        x = range(1, len(self.avg_train_losses) + 1)
        plt.plot(x, self.avg_train_losses, label="Training Loss")
        plt.plot(x, self.avg_valid_losses, label="Validation Loss")
        plt.axvline(self.avg_valid_losses.index(min(self.avg_valid_losses)) + 1, linestyle="--", color="r",
                    label="Early Stopping Checkpoint")
        plt.xlabel("epochs"); plt.ylabel("loss"); plt.grid(True); plt.legend(); plt.tight_layout()
        fig.savefig(out_dir / f"{model_out_path.stem}_loss_plot.png", bbox_inches="tight")
        plt.close(fig)

    def output_prediction_figure(self, model_path: Path) -> None:
        """Data / ground truth / prediction panels for the first validation batch (:485-535)."""
        model_path = Path(model_path)
        self.model.eval()
        # (a rank whose share of a partial validation batch is empty gets None from the loader: the first real batch)
        batch = next((b for b in self.validation_loader if b is not None), None)
        if batch is None:
            return
        with torch.no_grad():
            inputs, targets = utils.prepare_training_batch(batch, self._device(), self.label_no)
            labels = torch.argmax(torch.softmax(self.model(inputs), dim=1), dim=1)
        try:
            import matplotlib
            matplotlib.use("Agg")
            from matplotlib import pyplot as plt
        except ImportError:
            return
        rows = min(4, inputs.shape[0])
        fig = plt.figure(figsize=(12, 4 * rows))
        for j in range(rows):
            for k, (img, title) in enumerate(((inputs[j, 0].cpu(), "Data"), (torch.argmax(targets[j], 0).cpu(), "Ground Truth"),
                                              (labels[j].cpu(), "Prediction"))):
                ax = fig.add_subplot(rows, 3, 3 * j + k + 1)
                ax.imshow(img, cmap="gray")
                if j == 0:
                    ax.set_title(title)
        plt.suptitle(f"Predictions for {model_path.name}", fontsize=16)
        plt.savefig(model_path.parent / f"{model_path.stem}_prediction_image.png", dpi=150)
        plt.close(fig)
