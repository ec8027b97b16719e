"""VolSeg2dPredictor - slice-wise prediction of a 3-D volume along 1, 3 or 12 (axis x rotation) directions and
max-probability merge (reference: volume_segmantics/model/operations/vol_seg_2d_predictor.py:16-136).

Same methods, same return types (labels uint8, probabilities float16, one-hot votes uint8) - but the volume is
uploaded ONCE, every direction reads its slices straight from the resident uint8 volume through an index map
(np.rot90 / swapaxes are pure index maps, :34,:108), the network runs in libvolseg_hip, and softmax -> argmax ->
max-prob -> crop -> scatter writes each voxel's packed key (prob<<16 | (15-dir)<<8 | label) back at the voxel's own
address.  The running maximum of that key over directions equals the reference's chain of first-wins pairwise merges
(:90-98), is order independent, and is what the ranks exchange (one max all-reduce) when the slices of every
direction are sharded over several GPUs.
"""
from __future__ import annotations

import logging
import time
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

from ... import _lib, dist as vdist
from ..._lib import check, lib, ptr
from ...utilities import base_data_utils as utils
from ...utilities.base_data_utils import Axis


def direction_views(vol: np.ndarray, n_dirs: int):
    """The slice stacks of the 1 / 3 / 12 directions in the reference's call order, as numpy views of ``vol``
    (Z, Y, X per rotation; rotations are cumulative np.rot90 in the (0, 1) plane, :73-87 and :105-115)."""
    views, v = [], vol
    for k in range(4 if n_dirs == 12 else 1):
        if k:
            v = np.rot90(v)
        for axis in ((Axis.Z, Axis.Y, Axis.X) if n_dirs >= 3 else (Axis.Z,)):
            views.append(utils.rotate_array_to_axis(v, axis))
    return views


# Four of the reference's twelve directions are EXACT repeats of earlier ones: a rotation by 90 degrees in the (0, 1) plane turns the
# Z stack into the Y stack and back, so (rotation 1, Z) holds the slices of (rotation 0, Y), (rotation 2, Z) those of (rotation 1, Y),
# (rotation 3, Z) those of (rotation 2, Y) and (rotation 3, Y) those of (rotation 0, Z) - the same images at the same voxel addresses,
# in reversed slice order (np.array_equal(views[later], views[earlier][::-1]) for every shape; tests/test_host_logic.py).  A later
# twin predicts what its earlier twin predicted, and the reference's merge keeps the earlier of two equal probabilities
# (vol_seg_2d_predictor.py:90-98: np.argmax over (running, new) returns index 0 on a tie), so in the max-probability merge the
# four later twins can never change a voxel: 8 forward passes give the reference's 12-direction result, bit for bit.
REPEATED_DIRECTIONS = {3: 1, 6: 4, 9: 7, 10: 0}      # later twin -> the earlier direction it repeats (indices in call order)


def dirmap_of(vol: np.ndarray, view: np.ndarray) -> _lib.DirMap:
    """vs_dirmap of a view: element strides and base offset are read off numpy's own view object."""
    item = vol.itemsize
    base = (view.__array_interface__["data"][0] - vol.__array_interface__["data"][0]) // item
    d, h, w = view.shape
    hp, top, ctop = utils.pad_crop_offsets(h)
    wp, left, cleft = utils.pad_crop_offsets(w)
    return _lib.DirMap(base=base, ss=view.strides[0] // item, sh=view.strides[1] // item, sw=view.strides[2] // item,
                       depth=d, h=h, w=w, hp=hp, wp=wp, pad_top=top, pad_left=left, crop_top=ctop, crop_left=cleft)


class HipBackend:
    """Device side of one prediction pass: resident volume + output volumes + the per-batch kernels."""

    def __init__(self, model, vol: np.ndarray, classes: int, mode: int, want_probs: bool):
        self.model, self.classes, self.mode = model, classes, mode
        dev = model.device
        self.vtype = _lib.VS_VOL[vol.dtype.name]        # the kernels read the volume in its own type (vs_slices_gather_typed)
        # plain copy from the caller's (pageable) array: 2.4 ms for 512^3 bytes on this platform - the same as from pinned
        # memory, and without the 3.7 ms staging copy into it (tools/h2d_probe.py)
        self.vol = torch.from_numpy(vol.reshape(-1).view(np.uint8)).to(dev)
        n = vol.size
        self.nvox = n
        self.npad = vdist.padded_len(n, vdist.world()[1])      # the key volume splits into equal shards for the exchange
        self.labels = torch.zeros(n, dtype=torch.uint8, device=dev) if mode == 0 else None
        self.probs = torch.zeros(n, dtype=torch.float16, device=dev) if (mode == 0 and want_probs) else None
        self.keys = torch.zeros(self.npad, dtype=torch.int32, device=dev) if mode == 1 else None
        self.votes = torch.zeros((classes, n), dtype=torch.uint8, device=dev) if mode == 2 else None
        self._x = None
        self._merged = None

    def run_batch(self, dmap: _lib.DirMap, direction: int, s0: int, nb: int, votes: int = 1) -> None:
        """votes (vote mode only): 2 = the direction also stands for its exact repeat (REPEATED_DIRECTIONS), which is not run."""
        need = nb * dmap.hp * dmap.wp
        if self._x is None or self._x.numel() < need:
            self._x = torch.empty(need, dtype=torch.float32, device=self.model.device)
        x = self._x[:need].view(nb, 1, dmap.hp, dmap.wp)
        st = _lib.stream_ptr()
        check(lib.vs_slices_gather_typed(self.vtype, ptr(self.vol), dmap, s0, nb, ptr(x), st))
        # forward + softmax / arg-max / crop / scatter in one call: with <= 4 classes the head kernel writes labels /
        # probabilities / keys itself and no logits exist (identical results to vs_unet_forward + vs_logits_to_volume)
        mode = 3 if (self.mode == 2 and votes == 2) else self.mode
        self.model._forward_to_volume(x, dmap, s0, mode, direction, self.labels, self.probs, self.keys, self.votes, self.nvox)

    def _unpack(self, keys_i32: torch.Tensor, want_probs: bool = True):
        n = keys_i32.numel()
        labels = torch.empty(n, dtype=torch.uint8, device=keys_i32.device)
        probs = torch.empty(n, dtype=torch.float16, device=keys_i32.device) if want_probs else None
        check(lib.vs_keys_unpack(ptr(keys_i32), ptr(labels), ptr(probs), n, _lib.stream_ptr()))
        return labels, probs

    def exchange(self, want_probs: bool = True, all_ranks: bool = True) -> None:
        """Keys: reduce-scatter(MAX) -> every rank unpacks its shard -> labels / probabilities gathered (rank 0, or all ranks);
        votes: one SUM all-reduce."""
        if self.mode == 1:
            self._merged = vdist.exchange_keys_sharded(self.keys, lambda k: self._unpack(k, want_probs), want_probs, all_ranks)
        elif self.mode == 2:
            vdist.allreduce_sum_votes(self.votes)

    def results(self, shape, want_probs: bool):
        if self.mode == 2:
            return self.votes.cpu().numpy().reshape((self.classes,) + tuple(shape)), None
        if self.mode == 1:
            if self._merged is None:
                self._merged = self._unpack(self.keys, want_probs)
            labels, probs = self._merged
            labels = labels[:self.nvox]
            probs = probs[:self.nvox] if probs is not None else None
        else:
            labels, probs = self.labels, self.probs
        def download(t):
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)    # (the pinned blocks are cached across calls)
            host.copy_(t, non_blocking=True)
            return host
        hl = download(labels)
        hp = download(probs) if (want_probs and probs is not None) else None
        torch.cuda.current_stream().synchronize()
        return hl.numpy().reshape(shape), (hp.numpy().reshape(shape) if hp is not None else None)


class VolSeg2dPredictor:
    """Class that performs U-Net prediction operations. Does not interact with disk."""

    backend_factory = HipBackend  # tests of the sharding / merge logic substitute a CPU stand-in here
    result_ranks = "all"          # "all": every rank returns the merged volume; "rank0": only rank 0 downloads it (others get None)
    last_timings: dict = {}

    def __init__(self, model_file_path: str, settings: SimpleNamespace) -> None:
        from ..model_2d import create_model_from_file
        self.model_file_path = Path(model_file_path)
        self.settings = settings
        self.model_device_num = int(settings.cuda_device)
        self.model, self.num_labels, self.label_codes = create_model_from_file(
            self.model_file_path, device_num=self.model_device_num, precision=getattr(settings, "precision", None))

    def _get_model_from_trainer(self, trainer):
        self.model = trainer.model

    # ---- engine ---------------------------------------------------------------------------------------------
    @staticmethod
    def _device_volume(data_vol: np.ndarray) -> np.ndarray:
        """The volume as the device reads it: C-contiguous, in its OWN dtype.  The reference normalises every integer dtype
        as float32(v) / 255 and passes float volumes through without the / 255 (data/datasets.py:128-134) - reachable with
        ``clip_data: False`` - and vs_slices_gather_typed does the same per type; uint8 (clip_data's output,
        data/base_data_manager.py:38-40) is the common case."""
        if data_vol.dtype.name not in _lib.VS_VOL:
            raise TypeError(f"prediction volumes of dtype {data_vol.dtype} are not supported "
                            f"(supported: {', '.join(sorted(_lib.VS_VOL))})")
        vol = np.ascontiguousarray(data_vol)
        return vol if vol.dtype.isnative else vol.astype(vol.dtype.newbyteorder("="))

    def _run(self, data_vol: np.ndarray, n_dirs: int, mode: int, want_probs: bool, first_axis: Axis = Axis.Z):
        vol = self._device_volume(data_vol)
        if vol.ndim != 3:
            raise ValueError(f"expected a 3-D volume, got shape {vol.shape}")
        rank, world = vdist.world()
        if world > 1 and mode == 0:
            mode = 1  # shards meet through the key volume
        t0 = time.perf_counter()
        backend = self.backend_factory(self.model, vol, self.num_labels, mode, want_probs)
        batch = utils.get_batch_size(self.settings, prediction=True)
        if hasattr(self.model, "eval"):
            self.model.eval()
        views = direction_views(vol, n_dirs) if n_dirs > 1 else [utils.rotate_array_to_axis(vol, first_axis)]
        # 12 directions: the four exact repeats are not run (`dedup_directions: false` runs them anyway - same result, 12 / 8 of the
        # time).  Max-probability merge: a repeat cannot change it.  Votes: its earlier twin votes twice.
        skip = REPEATED_DIRECTIONS if (n_dirs == 12 and mode in (1, 2) and bool(getattr(self.settings, "dedup_directions", True))) else {}
        twice = set(skip.values()) if mode == 2 else set()
        with torch.no_grad():
            for d, view in enumerate(views):
                if d in skip:
                    logging.info(f"Direction {d + 1}/{len(views)} repeats direction {skip[d] + 1}: not run.")
                    continue
                dmap = dirmap_of(vol, view)
                lo, hi = vdist.shard_range(dmap.depth, rank, world)
                logging.info(f"Predicting direction {d + 1}/{len(views)}: slices [{lo}, {hi}) of stack {view.shape}.")
                for s0 in range(lo, hi, batch):
                    if d in twice:
                        backend.run_batch(dmap, d, s0, min(batch, hi - s0), votes=2)
                    else:
                        backend.run_batch(dmap, d, s0, min(batch, hi - s0))
        profile = bool(getattr(self.settings, "profile_phases", False)) and torch.cuda.is_available()
        if profile:
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        backend.exchange(want_probs, self.result_ranks != "rank0")
        if profile:
            torch.cuda.synchronize()
        t2 = time.perf_counter()
        if self.result_ranks == "rank0" and rank != 0:
            out = (None, None)
        else:
            out = backend.results(vol.shape, want_probs)
        t3 = time.perf_counter()
        self.last_timings = {"upload_and_directions_s": t1 - t0, "exchange_s": t2 - t1, "unpack_download_s": t3 - t2,
                             "directions_run": len(views) - len(skip)}
        return out

    # ---- the reference's methods --------------------------------------------------------------------------------
    def _predict_single_axis(self, data_vol, output_probs=True, axis=Axis.Z):
        return self._run(data_vol, 1, 0, output_probs, axis)

    def _predict_3_ways_max_probs(self, data_vol):
        return self._run(data_vol, 3, 1, True)

    def _predict_12_ways_max_probs(self, data_vol):
        return self._run(data_vol, 12, 1, True)

    def _merge_vols_in_mem(self, prob_container, label_container):
        """The reference's pairwise merge on host containers (2, Z, Y, X), executed by vs_merge_maxprob."""
        dev = self.model.device
        l0, p0 = torch.from_numpy(label_container[0]).to(dev), torch.from_numpy(prob_container[0]).to(dev)
        l1, p1 = torch.from_numpy(label_container[1]).to(dev), torch.from_numpy(prob_container[1]).to(dev)
        check(lib.vs_merge_maxprob(ptr(l0), ptr(p0), ptr(l1), ptr(p1), l0.numel(), _lib.stream_ptr()))
        label_container[0], prob_container[0] = l0.cpu().numpy(), p0.cpu().numpy()

    def _predict_single_axis_to_one_hot(self, data_vol, axis=Axis.Z):
        return self._run(data_vol, 1, 2, False, axis)[0]

    def _predict_3_ways_one_hot(self, data_vol):
        return self._run(data_vol, 3, 2, False)[0]

    def _predict_12_ways_one_hot(self, data_vol):
        return self._run(data_vol, 12, 2, False)[0]
