"""Losses and metrics the trainer can select (reference: volume_segmantics/data/pytorch3dunet_losses.py:15-184 and
pytorch3dunet_metrics.py:16-106, themselves vendored from pytorch-3dunet).  Restated on (N, C, H, W) tensors; torch
ops on the logits the engine returns - SURVEY.md section 8f row N3 lists their fusion as a next step."""
from __future__ import annotations

import torch
from torch import nn


def _per_channel(t: torch.Tensor) -> torch.Tensor:
    """(N, C, ...) -> (C, N * ...)"""
    return t.transpose(0, 1).reshape(t.size(1), -1)


def compute_per_channel_dice(input, target, epsilon=1e-6, weight=None):
    """2 * sum(x t) / clamp(sum(x^2) + sum(t^2)) per channel (V-Net form, pytorch3dunet_losses.py:15-41)."""
    if input.size() != target.size():
        raise ValueError("'input' and 'target' must have the same shape")
    x, t = _per_channel(input), _per_channel(target).float()
    inter = (x * t).sum(-1)
    if weight is not None:
        inter = weight * inter
    denom = (x * x).sum(-1) + (t * t).sum(-1)
    return 2 * (inter / denom.clamp(min=epsilon))


class DiceLoss(nn.Module):
    """1 - mean per-channel Dice; ``normalization`` in {'sigmoid', 'softmax', 'none'} (the trainer uses 'none',
    vol_seg_2d_trainer.py:133-135, i.e. the loss acts on raw logits)."""

    def __init__(self, weight=None, normalization="sigmoid"):
        super().__init__()
        if normalization not in ("sigmoid", "softmax", "none"):
            raise ValueError(normalization)
        self.register_buffer("weight", weight)
        self.normalization = {"sigmoid": torch.sigmoid, "softmax": lambda x: torch.softmax(x, 1), "none": lambda x: x}[normalization]

    def forward(self, input, target):
        return 1.0 - torch.mean(compute_per_channel_dice(self.normalization(input), target, weight=self.weight))


class _FusedDiceFn(torch.autograd.Function):
    """DiceLoss(normalization="none") forward + gradient as two HIP sweeps (vs_dice_loss_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, logits, targets, eps, group=None):
        from .. import _lib
        n, k = logits.shape[:2]
        hw = logits[0, 0].numel()
        logits = logits.contiguous()
        targets = targets.contiguous()
        is_f32 = targets.dtype == torch.float32
        if not is_f32 and targets.dtype != torch.uint8:
            targets, is_f32 = targets.float(), True
        ws = torch.empty(_lib.lib.vs_dice_workspace(k) // 4, dtype=torch.float32, device=logits.device)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        _lib.check(_lib.lib.vs_dice_loss_fwd(_lib.ptr(logits), _lib.ptr(targets), int(is_f32), n, k, hw, eps, _lib.ptr(loss),
                                            _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()))
        world = 1
        if group is not None:
            import torch.distributed as dist
            world = dist.get_world_size(group)
        if world > 1:
            # the Dice of the GLOBAL batch (the reference's single loader sees all of it): the per-class sums I_c, D_c - the last 2 k
            # floats of the workspace, which the gradient sweep reads - summed over the ranks; the loss from the global sums
            stats = ws[ws.numel() - 2 * k:]
            dist.all_reduce(stats, group=group)
            inter, denom = stats.view(k, 2)[:, 0].double(), stats.view(k, 2)[:, 1].double()
            loss = (1.0 - (2.0 * inter / denom.clamp(min=eps)).mean()).float()
        ctx.save_for_backward(logits, targets, ws)
        ctx.meta = (n, k, hw, eps, is_f32, world)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        from .. import _lib
        logits, targets, ws = ctx.saved_tensors
        n, k, hw, eps, is_f32, world = ctx.meta
        dx = torch.empty_like(logits)
        # (global Dice: every rank holds d loss / d its own logits; the gradient all-reduce AVERAGES the ranks' parameter gradients,
        # the global loss wants their SUM)
        g = (grad_out if world == 1 else grad_out * float(world)).contiguous().float()    # (one rank: no multiply launch)
        _lib.check(_lib.lib.vs_dice_loss_bwd(_lib.ptr(logits), _lib.ptr(targets), int(is_f32), _lib.ptr(g), n, k, hw, eps,
                                            _lib.ptr(ws), _lib.ptr(dx), _lib.stream_ptr()))
        return dx, None, None, None


class HipDiceLoss(nn.Module):
    """Drop-in for DiceLoss(normalization="none") on GPU tensors (falls through to the torch ops for anything else,
    e.g. class weights or CPU tensors in the tests of the host logic).  ``global_group``: a torch.distributed group - the loss is
    then the Dice of the GLOBAL batch (per-class sums all-reduced; with SyncBatchNorm, N ranks train exactly the step one process
    would run on the whole batch); default: every rank's own batch, as under a DistributedDataParallel wrap of the reference."""

    def __init__(self, epsilon: float = 1e-6, global_group=None):
        super().__init__()
        self.epsilon = epsilon
        self.global_group = global_group
        self._torch = DiceLoss(normalization="none")

    def forward(self, input, target):
        if input.is_cuda and input.dtype == torch.float32 and input.dim() >= 3 and input.shape == target.shape:
            # the cross-rank Dice is a TRAINING construct (it pairs with SyncBatchNorm's global batch): under torch.no_grad() - the
            # validation loop - every rank evaluates its own batches, whose count can differ between ranks when the last global
            # batch is partial (a rank with an empty share skips it), so a collective there would be mismatched across ranks
            group = self.global_group if (torch.is_grad_enabled() and input.requires_grad) else None
            return _FusedDiceFn.apply(input, target, self.epsilon, group)
        return self._torch(input, target)


class GeneralizedDiceLoss(nn.Module):
    """pytorch3dunet_losses.py:138-169: label contributions weighted by the inverse squared label volume."""

    def __init__(self, normalization="sigmoid", epsilon=1e-6):
        super().__init__()
        self.epsilon = epsilon
        self.normalization = {"sigmoid": torch.sigmoid, "softmax": lambda x: torch.softmax(x, 1), "none": lambda x: x}[normalization]

    def forward(self, input, target):
        x, t = _per_channel(self.normalization(input)), _per_channel(target).float()
        if x.size(0) == 1:  # put foreground and background in separate channels
            x, t = torch.cat((x, 1 - x), 0), torch.cat((t, 1 - t), 0)
        w = t.sum(-1)
        w = (1 / (w * w).clamp(min=self.epsilon)).detach()
        inter = ((x * t).sum(-1) * w).sum()
        denom = ((x + t).sum(-1) * w).clamp(min=self.epsilon).sum()
        return 1.0 - 2 * (inter / denom)


class BCEDiceLoss(nn.Module):
    def __init__(self, alpha, beta):
        super().__init__()
        self.alpha, self.beta = alpha, beta
        self.bce, self.dice = nn.BCEWithLogitsLoss(), DiceLoss()

    def forward(self, input, target):
        return self.alpha * self.bce(input, target) + self.beta * self.dice(input, target)


class _FusedSegLossFn(torch.autograd.Function):
    """BCE-Dice / BCE / cross entropy / generalised Dice as one HIP reduction sweep + one gradient sweep (vs_seg_loss_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, logits, targets, kind, alpha, beta, eps):
        from .. import _lib
        n, k = logits.shape[:2]
        hw = logits[0, 0].numel()
        logits = logits.contiguous()
        targets = targets.contiguous()
        is_f32 = targets.dtype == torch.float32
        if not is_f32 and targets.dtype != torch.uint8:
            targets, is_f32 = targets.float(), True
        nbytes = _lib.lib.vs_seg_loss_workspace(k)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=logits.device)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        _lib.check(_lib.lib.vs_seg_loss_fwd(kind, _lib.ptr(logits), _lib.ptr(targets), int(is_f32), n, k, hw, float(alpha), float(beta),
                                           float(eps), _lib.ptr(loss), _lib.ptr(ws), nbytes, _lib.stream_ptr()))
        ctx.save_for_backward(logits, targets, ws)
        ctx.meta = (kind, n, k, hw, is_f32)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        from .. import _lib
        logits, targets, ws = ctx.saved_tensors
        kind, n, k, hw, is_f32 = ctx.meta
        dx = torch.empty_like(logits)
        g = grad_out.contiguous().float()
        _lib.check(_lib.lib.vs_seg_loss_bwd(kind, _lib.ptr(logits), _lib.ptr(targets), int(is_f32), _lib.ptr(g), n, k, hw, _lib.ptr(ws),
                                           _lib.ptr(dx), _lib.stream_ptr()))
        return dx, None, None, None, None, None


class HipSegLoss(nn.Module):
    """The trainer's non-default criteria on GPU tensors (vol_seg_2d_trainer.py:124-148): ``kind`` in {"BCEDiceLoss", "BCELoss",
    "CrossEntropyLoss", "GeneralizedDiceLoss"}; one-hot targets (uint8 or float, the shape of the logits) - for the cross
    entropy the class index is the position of the 1 (the trainer's ``torch.argmax(targets, dim=1)``).  Anything the fused
    kernels do not take (CPU tensors, > 16 classes) goes through the torch restatement ``fallback``."""

    KINDS = {"BCEDiceLoss": 1, "BCELoss": 2, "CrossEntropyLoss": 3, "GeneralizedDiceLoss": 4}

    def __init__(self, kind: str, alpha: float = 1.0, beta: float = 1.0, epsilon: float = 1e-6):
        super().__init__()
        self.kind, self.alpha, self.beta, self.epsilon = kind, alpha, beta, epsilon
        self.fallback = {"BCEDiceLoss": lambda: BCEDiceLoss(alpha, beta), "BCELoss": nn.BCEWithLogitsLoss,
                         "CrossEntropyLoss": nn.CrossEntropyLoss, "GeneralizedDiceLoss": lambda: GeneralizedDiceLoss(epsilon=epsilon)}[kind]()

    def forward(self, input, target):
        if (input.is_cuda and input.dtype == torch.float32 and input.dim() >= 3 and input.shape == target.shape
                and input.size(1) <= 16):
            return _FusedSegLossFn.apply(input, target, self.KINDS[self.kind], self.alpha, self.beta, self.epsilon)
        if self.kind == "CrossEntropyLoss":
            return self.fallback(input, torch.argmax(target, dim=1))
        return self.fallback(input, target.float())


class DiceCoefficient:
    def __init__(self, epsilon=1e-6, **kwargs):
        self.epsilon = epsilon

    def __call__(self, input, target):
        return torch.mean(compute_per_channel_dice(input.flatten(2), target.flatten(2), epsilon=self.epsilon))


class MeanIoU:
    """Per-sample argmax one-hot, per-class Jaccard, mean over classes then samples (pytorch3dunet_metrics.py:34-106).
    Accepts the (N, C, 1, H, W) tensors the reference's trainer builds or plain (N, C, H, W).  GPU tensors go through
    one HIP sweep (vs_mean_iou); the torch ops below serve CPU tensors and the skip_channels / ignore_index options."""

    def from_logits(self, logits, target):
        """metric(softmax(logits, dim=1), target) - the trainer's validation call - without materialising the softmax."""
        if (logits.is_cuda and logits.dtype == torch.float32 and logits.shape == target.shape and 1 < logits.size(1) <= 16
                and not self.skip_channels and self.ignore_index is None and target.dtype in (torch.uint8, torch.float32)):
            return self._hip(logits, target, from_logits=1)
        return self(torch.softmax(logits, dim=1), target)

    @staticmethod
    def _hip(input, target, from_logits=0):
        from .._lib import check, lib, ptr, stream_ptr
        n, c = input.size(0), input.size(1)
        x, t = input.contiguous(), target.contiguous()
        hw = x.numel() // (n * c)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        nbytes = lib.vs_mean_iou_workspace(n, c)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        check(lib.vs_mean_iou(ptr(x), ptr(t), 1 if t.dtype == torch.float32 else 0, from_logits, n, c, hw, ptr(out), ptr(ws),
                              nbytes, stream_ptr()))
        return out

    def __init__(self, skip_channels=(), ignore_index=None, **kwargs):
        self.skip_channels, self.ignore_index = skip_channels, ignore_index

    def __call__(self, input, target):
        if (input.is_cuda and input.dtype == torch.float32 and input.shape == target.shape and input.size(1) <= 16
                and not self.skip_channels and self.ignore_index is None and target.dtype in (torch.uint8, torch.float32)):
            return self._hip(input, target)
        n_classes = input.size(1)
        scores = []
        for p, t in zip(input, target):
            if n_classes == 1:
                pred = (p > 0.5).to(torch.uint8)
            else:
                pred = torch.zeros_like(p, dtype=torch.uint8).scatter_(0, torch.argmax(p, dim=0, keepdim=True), 1)
            t = t.to(torch.uint8)
            if self.ignore_index is not None:
                keep = t != self.ignore_index
                pred, t = pred * keep, t * keep
            ious = [torch.sum(pred[c] & t[c]).float() / torch.clamp(torch.sum(pred[c] | t[c]).float(), min=1e-8)
                    for c in range(n_classes) if c not in self.skip_channels]
            scores.append(torch.mean(torch.stack(ious)))
        return torch.mean(torch.stack(scores))
