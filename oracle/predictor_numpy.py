"""ORACLE (test infrastructure, never the product path).

numpy / torch-CPU restatement of the reference's slice-wise prediction and
multi-direction merge, each function citing the reference lines it follows.
Pinned against the reference's own code by oracle/gen_goldens.py ->
tests/golden/*.npz (tests/test_oracle_goldens.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

import math

import numpy as np
import torch

IM_SIZE_DIVISOR = 32  # volume_segmantics/utilities/config.py:34
IMAGENET_MEAN = 0.449  # config.py:41
IMAGENET_STD = 0.226  # config.py:42
PRED_BATCH = 4  # config.py:31


def get_padded_dimension(dimension: int) -> int:
    """volume_segmantics/data/augmentations.py:30-43 (KAT: 32->32, 33->64, 13->32, 0->0)."""
    if dimension % IM_SIZE_DIVISOR == 0:
        return dimension
    return (math.floor(dimension / IM_SIZE_DIVISOR) + 1) * IM_SIZE_DIVISOR


def pad_offsets(dim: int) -> tuple[int, int]:
    """albumentations ^1.1 PadIfNeeded (centre): pad_top = int(d/2), pad_bottom = d - pad_top."""
    d = get_padded_dimension(dim) - dim
    top = int(d / 2.0)
    return top, d - top


def crop_offset(padded: int, dim: int) -> int:
    """torchvision center_crop: int(round((padded - dim) / 2.0)) (Python banker's rounding).
    Differs from the pad offset by one when (padded-dim) == 3 (mod 4): SURVEY.md row P."""
    return int(round((padded - dim) / 2.0))


def preprocess_slice(image: np.ndarray) -> np.ndarray:
    """VolSeg2dPredictionDataset.__getitem__ (volume_segmantics/data/datasets.py:120-142):
    PadIfNeeded(reflect-101, centred) -> /255 for integer dtypes -> -mean -> /std."""
    (t, b), (l, r) = pad_offsets(image.shape[0]), pad_offsets(image.shape[1])
    if t or b or l or r:
        image = np.pad(image, ((t, b), (l, r)), mode="reflect")
    if np.issubdtype(image.dtype, np.integer):
        image = image.astype(np.float32)
        image = image / 255
    image = image - IMAGENET_MEAN
    image = image / IMAGENET_STD
    return image


def rotate_array_to_axis(array: np.ndarray, axis: int) -> np.ndarray:
    """volume_segmantics/utilities/base_data_utils.py:132-138. axis: 0=Z, 1=Y, 2=X."""
    if axis == 0:
        return array
    return array.swapaxes(0, axis)


def predict_single_axis(model: torch.nn.Module, data_vol: np.ndarray, axis: int = 0,
                        batch_size: int = PRED_BATCH, return_logits: bool = False):
    """VolSeg2dPredictor._predict_single_axis (vol_seg_2d_predictor.py:31-65)."""
    data_vol = rotate_array_to_axis(data_vol, axis)
    h, w = data_vol.shape[1:]
    hp, wp = get_padded_dimension(h), get_padded_dimension(w)
    top, left = crop_offset(hp, h), crop_offset(wp, w)
    labels_out, probs_out, logits_out = [], [], []
    model.eval()
    with torch.no_grad():
        for s0 in range(0, data_vol.shape[0], batch_size):
            batch = np.stack([preprocess_slice(data_vol[i]) for i in range(s0, min(s0 + batch_size, data_vol.shape[0]))])
            x = torch.from_numpy(np.ascontiguousarray(batch)).unsqueeze(1)
            if x.dtype == torch.float64:  # default collate keeps float64 for float volumes; the model is fp32
                x = x.float()
            output = model(x)
            probs = torch.softmax(output, dim=1)
            labels = torch.argmax(probs, dim=1)
            labels_out.append(labels[:, top:top + h, left:left + w].numpy().astype(np.uint8))
            idx = torch.argmax(probs, dim=1, keepdim=True)
            mp = torch.gather(probs, 1, idx).squeeze(1)
            probs_out.append(mp[:, top:top + h, left:left + w].numpy().astype(np.float16))
            if return_logits:
                logits_out.append(output[:, :, top:top + h, left:left + w].numpy())
    labels = rotate_array_to_axis(np.concatenate(labels_out), axis)
    probs = rotate_array_to_axis(np.concatenate(probs_out), axis)
    if return_logits:
        return labels, probs, np.concatenate(logits_out)
    return labels, probs


def merge_vols_in_mem(prob_container: np.ndarray, label_container: np.ndarray) -> None:
    """VolSeg2dPredictor._merge_vols_in_mem (vol_seg_2d_predictor.py:90-98): first max wins."""
    idx = np.argmax(prob_container, axis=0)[np.newaxis]
    prob_container[0] = np.squeeze(np.take_along_axis(prob_container, idx, axis=0))
    label_container[0] = np.squeeze(np.take_along_axis(label_container, idx, axis=0))


def predict_3_ways_max_probs(model, data_vol, batch_size=PRED_BATCH):
    """vol_seg_2d_predictor.py:67-88."""
    lab = np.empty((2, *data_vol.shape), dtype=np.uint8)
    prob = np.empty((2, *data_vol.shape), dtype=np.float16)
    lab[0], prob[0] = predict_single_axis(model, data_vol, 0, batch_size)
    lab[1], prob[1] = predict_single_axis(model, data_vol, 1, batch_size)
    merge_vols_in_mem(prob, lab)
    lab[1], prob[1] = predict_single_axis(model, data_vol, 2, batch_size)
    merge_vols_in_mem(prob, lab)
    return lab[0], prob[0]


def predict_12_ways_max_probs(model, data_vol, batch_size=PRED_BATCH):
    """vol_seg_2d_predictor.py:100-116."""
    lab = np.empty((2, *data_vol.shape), dtype=np.uint8)
    prob = np.empty((2, *data_vol.shape), dtype=np.float16)
    lab[0], prob[0] = predict_3_ways_max_probs(model, data_vol, batch_size)
    for k in range(1, 4):
        data_vol = np.rot90(data_vol)
        l, p = predict_3_ways_max_probs(model, data_vol, batch_size)
        lab[1] = np.rot90(l, -k)
        prob[1] = np.rot90(p, -k)
        merge_vols_in_mem(prob, lab)
    return lab[0], prob[0]


def one_hot_encode_array(input_array: np.ndarray, num_labels: int) -> np.ndarray:
    """volume_segmantics/utilities/base_data_utils.py:141-147."""
    out = np.zeros((num_labels, input_array.size), dtype=np.uint8)
    out[input_array.ravel(), np.arange(input_array.size)] = 1
    out.shape = (num_labels,) + input_array.shape
    return out


def predict_3_ways_one_hot(model, data_vol, num_labels, batch_size=PRED_BATCH):
    """vol_seg_2d_predictor.py:118-126."""
    out = one_hot_encode_array(predict_single_axis(model, data_vol, 0, batch_size)[0], num_labels)
    out += one_hot_encode_array(predict_single_axis(model, data_vol, 1, batch_size)[0], num_labels)
    out += one_hot_encode_array(predict_single_axis(model, data_vol, 2, batch_size)[0], num_labels)
    return out


def predict_12_ways_one_hot(model, data_vol, num_labels, batch_size=PRED_BATCH):
    """vol_seg_2d_predictor.py:128-136."""
    out = predict_3_ways_one_hot(model, data_vol, num_labels, batch_size)
    for k in range(1, 4):
        data_vol = np.rot90(data_vol)
        out += np.rot90(predict_3_ways_one_hot(model, data_vol, num_labels, batch_size), -k, axes=(-3, -2))
    return out


# ---- packed-key formulation -------------------------------

def pack_key(prob_f16: np.ndarray, label_u8: np.ndarray, direction: int) -> np.ndarray:
    """uint32 key = fp16 bits << 16 | (15 - direction) << 8 | label.  For non-negative
    probabilities the fp16 bit pattern orders like the value, so the elementwise max of
    keys over directions == the reference's chain of first-wins pairwise merges."""
    bits = prob_f16.view(np.uint16).astype(np.uint32)
    return (bits << 16) | (np.uint32(15 - direction) << 8) | label_u8.astype(np.uint32)


def unpack_key(key: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    return (key & 0xFF).astype(np.uint8), (key >> 16).astype(np.uint16).view(np.float16)


def direction_views(vol: np.ndarray, n_dirs: int):
    """The n_dirs (1|3|12) slice stacks in the reference's call order
    (vol_seg_2d_predictor.py:73-87,105-115), as numpy *views* of ``vol``: element
    [s,h,w] of view d is the voxel that slice s / pixel (h,w) of direction d reads and
    writes.  rot90/swapaxes are pure index maps, so the views carry (offset, strides)."""
    views = []
    v = vol
    for k in range(4 if n_dirs == 12 else 1):
        if k:
            v = np.rot90(v)
        for axis in ((0, 1, 2) if n_dirs >= 3 else (0,)):
            views.append(rotate_array_to_axis(v, axis))
    return views


# ---- training-side helpers ----------------------------------------------------------------------

def dice_loss_none(output: torch.Tensor, target: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """DiceLoss(normalization='none') (volume_segmantics/data/pytorch3dunet_losses.py:15-41,89-135)."""
    c = output.size(1)
    x = output.transpose(0, 1).reshape(c, -1)
    t = target.transpose(0, 1).reshape(c, -1).float()
    inter = (x * t).sum(-1)
    denom = (x * x).sum(-1) + (t * t).sum(-1)
    return 1.0 - torch.mean(2 * (inter / denom.clamp(min=eps)))


def prepare_training_batch(img: torch.Tensor, mask: torch.Tensor, num_labels: int):
    """volume_segmantics/utilities/base_data_utils.py:150-158."""
    t = torch.nn.functional.one_hot(mask.to(torch.int64), num_classes=num_labels)
    return img, t.permute(0, 3, 1, 2).to(torch.uint8)


def find_lr_from_graph(losses, lrs, default_min_lr=0.00075, lr_divisor=3):
    """VolSeg2dTrainer._find_lr_from_graph (vol_seg_2d_trainer.py:347-383)."""
    losses = np.array([float(l) for l in losses])
    try:
        gradients = np.gradient(losses)
        if gradients.min() >= 0:
            return default_min_lr
        idx = gradients.argmin()
    except Exception:
        return default_min_lr
    return lrs[idx] / lr_divisor


def mean_iou(probs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """MeanIoU.__call__ for (N,C,H,W) probs and one-hot targets
    (volume_segmantics/data/pytorch3dunet_metrics.py:47-106; the reference unsqueezes both
    to 5-D at vol_seg_2d_trainer.py:241-243, which does not change the result)."""
    n_classes = probs.size(1)
    per_batch = []
    for p, t in zip(probs, targets):
        if n_classes == 1:
            pred = (p > 0.5).to(torch.uint8)
        else:
            pred = torch.zeros_like(p, dtype=torch.uint8).scatter_(0, torch.argmax(p, dim=0, keepdim=True), 1)
        t = t.to(torch.uint8)
        ious = [torch.sum(pred[c] & t[c]).float() / torch.clamp(torch.sum(pred[c] | t[c]).float(), min=1e-8)
                for c in range(n_classes)]
        per_batch.append(torch.mean(torch.tensor(ious)))
    return torch.mean(torch.tensor(per_batch))
