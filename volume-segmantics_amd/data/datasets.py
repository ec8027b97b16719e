"""Slice datasets feeding the trainer (reference: volume_segmantics/data/datasets.py:12-181, dataloaders.py:15-71,
augmentations.py).  PNG slices are read with PIL.  The reference's albumentations pipeline (RandomSizedCrop, flips / rotations /
transposes, elastic / grid / optical distortion, CLAHE, brightness-contrast / gamma) is restated in data/augmentations.py; with
``augment="device"`` (the default of get_2d_training_dataloaders when a GPU is present) the dataset hands out the raw uint8 pairs
and the batch is augmented and normalised in HBM by csrc/augment.hip (prepare_training_batch), with ``augment="host"`` every
sample goes through the NumPy form in the loader workers, as the reference does it."""
from __future__ import annotations

import logging
import re
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, Subset

from ..utilities import base_data_utils as utils
from ..utilities import config as cfg


def natsort_key(item):
    return [int(t) if t.isdigit() else t.lower() for t in re.split(r"(\d+)", str(item))]


def _read_gray(path: Path) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("L"))


def fit_to_square(image: np.ndarray, mask: np.ndarray | None, size: int):
    """LongestMaxSize(size) then centred PadIfNeeded(size, size) with reflect-101 borders (augmentations.py:12-27)."""
    from .augmentations import resize      # cv2.INTER_LINEAR / INTER_NEAREST semantics: no antialiasing on down-scaling
    h, w = image.shape
    scale = size / max(h, w)
    if scale != 1.0:
        nh, nw = max(1, int(round(h * scale))), max(1, int(round(w * scale)))
        image = resize(image, nh, nw)
        if mask is not None:
            mask = resize(mask, nh, nw, nearest=True)
        h, w = nh, nw
    top, left = (size - h) // 2, (size - w) // 2
    pads = ((top, size - h - top), (left, size - w - left))
    if any(p for pair in pads for p in pair):
        mode = "reflect" if min(h, w) > 1 else "edge"
        image = np.pad(image, pads, mode=mode)
        if mask is not None:
            mask = np.pad(mask, pads, mode=mode)
    return image, mask


def normalise(image: np.ndarray) -> np.ndarray:
    """datasets.py:63-69: /255 for integer images, then the single-channel ImageNet mean / std."""
    if np.issubdtype(image.dtype, np.integer):
        image = image.astype(np.float32) / 255
    return (image - cfg.IMAGENET_MEAN) / cfg.IMAGENET_STD


class VolSeg2dDataset(Dataset):
    def __init__(self, images_dir: Path, masks_dir: Path, img_size: int, augment, seed: int = 0):
        """augment: False / None (validation), "host" or True (the NumPy pipeline per sample), "device" (raw uint8 pairs: the
        trainer augments the batch on the GPU)."""
        self.images_fps = sorted(Path(images_dir).glob("*.png"), key=natsort_key)
        self.masks_fps = sorted(Path(masks_dir).glob("*.png"), key=natsort_key)
        if len(self.images_fps) != len(self.masks_fps):
            raise ValueError("image / label slice counts differ")
        self.img_size, self.augment = img_size, augment
        self.seed, self.rng, self._rng_owner = seed, None, None

    def _worker_rng(self):
        """One generator per loader worker and per epoch: DataLoader re-forks its workers every epoch with a fresh
        ``torch.utils.data.get_worker_info().seed``, so copies of ONE generator made at fork time would draw the same
        flips in every worker and every epoch."""
        info = torch.utils.data.get_worker_info()
        owner = (info.id, info.seed) if info is not None else ("main", torch.initial_seed())
        if self.rng is None or owner != self._rng_owner:
            self.rng = np.random.default_rng([self.seed, owner[1] & 0xFFFFFFFF] if info is not None else self.seed)
            self._rng_owner = owner
        return self.rng

    def __len__(self):
        return len(self.images_fps)

    def __getitem__(self, i):
        image, mask = fit_to_square(_read_gray(self.images_fps[i]), _read_gray(self.masks_fps[i]), self.img_size)
        if self.augment == "device":     # uint8 out: augmentation + normalisation happen on the GPU, batch-wise
            return torch.from_numpy(np.ascontiguousarray(image)).unsqueeze(0), torch.from_numpy(np.ascontiguousarray(mask))
        if self.augment:
            from .augmentations import train_augment
            image, mask = train_augment(np.ascontiguousarray(image), np.ascontiguousarray(mask), self.img_size, self._worker_rng())
        image = normalise(np.ascontiguousarray(image)).astype(np.float32)
        return torch.from_numpy(image).unsqueeze(0), torch.from_numpy(np.ascontiguousarray(mask))


class ArraySliceDataset(Dataset):
    """In-memory (image, mask) slices - used by the synthetic benchmarks and tests (no PNG round trip)."""

    def __init__(self, images: np.ndarray, masks: np.ndarray):
        self.images, self.masks = images, masks

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return torch.from_numpy(normalise(self.images[i]).astype(np.float32)).unsqueeze(0), torch.from_numpy(self.masks[i])


class ShardedBatchSampler(torch.utils.data.Sampler):
    """The reference's ONE loader (shuffle, drop_last; data/dataloaders.py:42-57) seen from ``world`` data-parallel ranks:
    every rank draws the SAME permutation (generator seeded with ``seed + epoch``), cuts it into global batches of
    ``batch_size * world`` samples and takes its own contiguous ``batch_size`` of each - disjoint shards that together are the
    global batch.  ``drop_last`` acts on the GLOBAL batch, so every rank runs the same number of steps (a rank that ran one
    step more would wait in an all-reduce for ever).  Without ``drop_last`` (validation) the last, partial global batch is
    dealt out as evenly as possible and a rank may receive an empty list for it.  ``set_epoch`` reshuffles."""

    def __init__(self, n: int, batch_size: int, rank: int = 0, world: int = 1, shuffle: bool = True, drop_last: bool = True,
                 seed: int = 0):
        self.n, self.batch_size, self.rank, self.world = int(n), int(batch_size), int(rank), int(world)
        self.shuffle, self.drop_last, self.seed, self.epoch = shuffle, drop_last, int(seed), 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __len__(self) -> int:
        g = self.batch_size * self.world
        return self.n // g if self.drop_last else (self.n + g - 1) // g

    def __iter__(self):
        if self.shuffle:
            order = torch.randperm(self.n, generator=torch.Generator().manual_seed(self.seed + self.epoch)).tolist()
        else:
            order = list(range(self.n))
        g = self.batch_size * self.world
        for b in range(len(self)):
            chunk = order[b * g:(b + 1) * g]
            if len(chunk) == g:
                yield chunk[self.rank * self.batch_size:(self.rank + 1) * self.batch_size]
            else:   # partial last global batch (validation): shares differ by at most one sample
                base, rem = divmod(len(chunk), self.world)
                lo = self.rank * base + min(self.rank, rem)
                yield chunk[lo:lo + base + (1 if self.rank < rem else 0)]


def _collate_maybe_empty(batch):
    if not batch:    # this rank's share of a partial validation batch is empty: the trainer skips it (weight 0)
        return None
    return torch.utils.data.default_collate(batch)


def make_training_loaders(train_ds, valid_ds, batch_size: int, rank: int = 0, world: int = 1, seed: int = 0, **loader_kw):
    """(training loader, validation loader) over the two datasets for rank ``rank`` of ``world`` (see ShardedBatchSampler;
    world == 1 is exactly the reference's shuffle + drop_last loader pair)."""
    ts = ShardedBatchSampler(len(train_ds), batch_size, rank, world, shuffle=True, drop_last=True, seed=seed)
    vs = ShardedBatchSampler(len(valid_ds), batch_size, rank, world, shuffle=False, drop_last=False)
    return (DataLoader(train_ds, batch_sampler=ts, **loader_kw),
            DataLoader(valid_ds, batch_sampler=vs, collate_fn=_collate_maybe_empty, **loader_kw))


class ResidentSliceLoader:
    """The training feed without the per-step PNG decode: every slice pair of the subset is decoded ONCE (the reference decodes
    each PNG again in every epoch, in 4 loader workers - data/datasets.py:42-60, utilities/config.py:33: ~1 ms per 256 x 256 pair
    and worker, i.e. at most ~4 000 slices/s against a training step that consumes 7 000), kept as uint8 in HBM, and a batch
    is an index gather on the device.  Iterates like the DataLoader it replaces: the same ShardedBatchSampler decides the
    batches (`batch_sampler`, `set_epoch`), it yields (images (b, 1, s, s) uint8, masks (b, s, s) uint8) on the device - what
    `prepare_training_batch` takes for device-side augmentation / normalisation - or None for an empty validation share."""

    def __init__(self, dataset, batch_sampler, device, decode_threads: int = 8):
        from concurrent.futures import ThreadPoolExecutor
        self.batch_sampler, self.device = batch_sampler, torch.device(device)
        self.num_labels = None      # set by the trainer: the label-range check of the reference's F.one_hot (see __iter__)
        n = len(dataset)
        if not n:
            self.images = torch.empty((0, 1, 0, 0), dtype=torch.uint8, device=self.device)
            self.masks = torch.empty((0, 0, 0), dtype=torch.uint8, device=self.device)
            self.max_label = -1
            return
        first = dataset[0]
        if first[0].dtype != torch.uint8:
            raise ValueError("ResidentSliceLoader needs raw uint8 pairs (VolSeg2dDataset(augment='device'))")
        # decoded straight into two preallocated (pinned, when a GPU is there) host buffers: no list of n tensors plus a stacked copy
        pin = self.device.type == "cuda"
        images = torch.empty((n,) + tuple(first[0].shape), dtype=torch.uint8, pin_memory=pin)
        masks = torch.empty((n,) + tuple(first[1].shape), dtype=torch.uint8, pin_memory=pin)

        def decode(i):
            img, msk = dataset[i] if i else first
            images[i].copy_(img)
            masks[i].copy_(msk)
        with ThreadPoolExecutor(max_workers=max(1, decode_threads)) as pool:      # PIL releases the GIL while it decodes
            list(pool.map(decode, range(n)))
        self.images, self.masks = images.to(self.device), masks.to(self.device)
        self.max_label = int(self.masks.max())      # one device reduction per loader, nothing per step

    @staticmethod
    def bytes_needed(n: int, image_size: int) -> int:
        """HBM the resident copy of n slice pairs takes (uint8 image + uint8 mask)."""
        return int(n) * int(image_size) * int(image_size) * 2

    def __len__(self):
        return len(self.batch_sampler)

    def __iter__(self):
        if self.num_labels is not None and self.max_label >= self.num_labels:
            # torch.nn.functional.one_hot in the reference's prepare_training_batch (utilities/base_data_utils.py:150-158) raises on
            # the first such batch (e.g. 0 / 255 PNG masks with 2 labels); the device one-hot kernel would silently give such
            # pixels an all-zero target.  The masks live on the device here, so the check is made once, on the whole subset
            raise RuntimeError("Class values must be smaller than num_classes.")
        for idx in self.batch_sampler:
            if not idx:
                yield None
                continue
            i = torch.as_tensor(idx, dtype=torch.int64, device=self.device)
            yield self.images.index_select(0, i), self.masks.index_select(0, i)


def shared_seed(rank: int, world: int) -> int:
    """A random seed that is the same on every rank: rank 0 draws it (the reference's split is unseeded), the others receive it."""
    seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64)
    if world > 1:
        import torch.distributed as dist
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        seed = seed.to(dev)
        dist.broadcast(seed, 0)
    return int(seed.item())


def get_2d_training_dataloaders(image_dir: Path, label_dir: Path, settings, rank: int = 0, world: int = 1):
    """80/20 random split, shuffled drop_last training loader (dataloaders.py:15-57).  With several ranks: the same split on
    every rank (one shared seed), disjoint shards of every global batch (batch_size is per rank)."""
    batch_size = utils.get_batch_size(settings)
    mode = getattr(settings, "augment", None) or ("device" if torch.cuda.is_available() and settings.image_size % 8 == 0 else "host")
    train_full = VolSeg2dDataset(image_dir, label_dir, settings.image_size, augment=mode)
    valid_full = VolSeg2dDataset(image_dir, label_dir, settings.image_size, augment=False)
    n = len(train_full)
    seed = shared_seed(rank, world)
    indices = torch.randperm(n, generator=torch.Generator().manual_seed(seed)).tolist()
    cut = int(n * settings.training_set_proportion)
    workers = int(getattr(settings, "num_workers", cfg.NUM_WORKERS))
    resident = getattr(settings, "resident_feed", None)
    if resident is None:      # default: on whenever the batches are augmented on the device anyway
        resident = mode == "device" and torch.cuda.is_available()
    if resident and mode == "device" and torch.cuda.is_available() and getattr(settings, "resident_feed", None) is None:
        # default-on only while the decoded subsets fit comfortably next to the training workspace: a quarter of the free HBM at most;
        # larger training sets stream through the DataLoader as in the reference
        free, _total = torch.cuda.mem_get_info()
        need = ResidentSliceLoader.bytes_needed(n, settings.image_size)
        if need > free // 4:
            logging.info(f"Training set of {n} slice pairs ({need / 2**30:.1f} GiB decoded) does not fit the resident feed's share of "
                         f"free device memory ({free / 2**30:.1f} GiB free): streaming through the DataLoader instead.")
            resident = False
    if resident:
        if mode != "device":
            raise ValueError("resident_feed needs device-side augmentation (a GPU and an image_size that is a multiple of 8)")
        dev = torch.device("cuda", torch.cuda.current_device())
        valid_raw = VolSeg2dDataset(image_dir, label_dir, settings.image_size, augment="device")      # raw pairs: normalised on the device
        ts = ShardedBatchSampler(cut, batch_size, rank, world, shuffle=True, drop_last=True, seed=seed + 1)
        vs = ShardedBatchSampler(n - cut, batch_size, rank, world, shuffle=False, drop_last=False)
        return (ResidentSliceLoader(Subset(train_full, indices[:cut]), ts, dev), ResidentSliceLoader(Subset(valid_raw, indices[cut:]), vs, dev))
    return make_training_loaders(Subset(train_full, indices[:cut]), Subset(valid_full, indices[cut:]), batch_size, rank, world,
                                 seed=seed + 1, num_workers=workers, pin_memory=cfg.PIN_CUDA_MEMORY and torch.cuda.is_available())
