"""ORACLE (test infrastructure, never the product path): the encoder of ``smp.Unet(encoder_name="efficientnet-b3" | "efficientnet-b4",
in_channels=1)`` (the reference's encoder list: README.md:57-76, tests/test_model_2d.py:36-44; BASELINE configs[4] names b4).

Restated from the published sources - neither package is installed here (no network):
  * segmentation-models-pytorch 0.2.1, encoders/efficientnet.py: ``EfficientNetEncoder(EfficientNet, EncoderMixin)`` - stages
    [identity, stem + bn0 + swish, blocks[:s0], blocks[s0:s1], blocks[s1:s2], blocks[s2:]], the drop-connect rate of block i is
    0.2 * i / len(blocks), ``_fc`` deleted, ``_conv_head`` / ``_bn1`` kept in the state dict but never run; stage_idxs / out_channels:
    b3 (5, 8, 18, 26) / (3, 40, 32, 48, 136, 384), b4 (6, 10, 22, 32) / (3, 48, 32, 56, 160, 448);
  * efficientnet-pytorch 0.6.3 (smp 0.2.1's pin), model.py / utils.py: MBConvBlock (expand 1x1 -> BN -> swish, depthwise k x k -> BN ->
    swish, squeeze-excitation with max(1, int(input_filters * 0.25)) hidden channels and swish, project 1x1 -> BN, drop_connect + skip
    when stride 1 and filters match), BatchNorm2d(momentum=0.01, eps=1e-3), ``Conv2dStaticSamePadding`` with the GLOBAL nominal image
    size (300 for b3, 380 for b4: even, so every stride-2 convolution pads (0, 1) for k = 3 and (1, 2) for k = 5, every stride-1
    convolution symmetrically), round_filters / round_repeats with width / depth coefficients (1.2, 1.4) / (1.4, 1.8), divisor 8.
Structural pins (tests/test_oracle_topology.py): the published parameter counts of efficientnet-pytorch - b3 12,233,232 and b4
19,341,616 with the 3-channel stem and the 1000-way ``_fc`` added back - and the encoder widths above."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# (repeats, kernel, stride, expand, input filters, output filters) of EfficientNet-B0; se_ratio 0.25 everywhere
BLOCKS_B0 = ((1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112),
             (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320))
COEFFS = {"efficientnet-b3": (1.2, 1.4, 300), "efficientnet-b4": (1.4, 1.8, 380)}     # width, depth, nominal image size
STAGE_IDXS = {"efficientnet-b3": (5, 8, 18, 26), "efficientnet-b4": (6, 10, 22, 32)}
OUT_CHANNELS = {"efficientnet-b3": (1, 40, 32, 48, 136, 384), "efficientnet-b4": (1, 48, 32, 56, 160, 448)}
PUBLISHED_PARAMS = {"efficientnet-b3": 12_233_232, "efficientnet-b4": 19_341_616}
BN_MOMENTUM, BN_EPS, DROP_CONNECT = 0.01, 1e-3, 0.2


def round_filters(filters: int, width: float, divisor: int = 8) -> int:
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def round_repeats(repeats: int, depth: float) -> int:
    return int(math.ceil(depth * repeats))


def block_plan(name: str):
    """[(kernel, stride, expand, input filters, output filters)] per MBConv block, in order."""
    width, depth, _ = COEFFS[name]
    plan = []
    for r, k, s, e, i, o in BLOCKS_B0:
        i, o = round_filters(i, width), round_filters(o, width)
        for j in range(round_repeats(r, depth)):
            plan.append((k, s if j == 0 else 1, e, i if j == 0 else o, o))
    return plan


class SamePadConv2d(nn.Conv2d):
    """efficientnet-pytorch's Conv2dStaticSamePadding: the padding is fixed at construction from the NOMINAL image size."""

    def __init__(self, cin, cout, kernel_size, stride=1, groups=1, bias=True, image_size=380):
        super().__init__(cin, cout, kernel_size, stride, 0, 1, groups, bias)
        k, s = kernel_size, stride
        out = math.ceil(image_size / s)
        pad = max((out - 1) * s + (k - 1) + 1 - image_size, 0)
        self.static_pad = (pad // 2, pad - pad // 2, pad // 2, pad - pad // 2)

    def forward(self, x):      # utils.Conv2dStaticSamePadding.forward: the static padding, then the convolution with its OWN attributes
        if any(self.static_pad):
            x = F.pad(x, self.static_pad)
        return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


def swish(x):
    return x * torch.sigmoid(x)


class MBConvBlock(nn.Module):
    def __init__(self, k, s, e, inp, out, image_size):
        super().__init__()
        self.stride, self.inp, self.out, self.expand = s, inp, out, e
        oup = inp * e
        bn = lambda c: nn.BatchNorm2d(c, momentum=BN_MOMENTUM, eps=BN_EPS)   # noqa: E731
        if e != 1:
            self._expand_conv = SamePadConv2d(inp, oup, 1, bias=False, image_size=image_size)
            self._bn0 = bn(oup)
        self._depthwise_conv = SamePadConv2d(oup, oup, k, s, groups=oup, bias=False, image_size=image_size)
        self._bn1 = bn(oup)
        sq = max(1, int(inp * 0.25))
        self._se_reduce = SamePadConv2d(oup, sq, 1, image_size=image_size)
        self._se_expand = SamePadConv2d(sq, oup, 1, image_size=image_size)
        self._project_conv = SamePadConv2d(oup, out, 1, bias=False, image_size=image_size)
        self._bn2 = bn(out)

    def forward(self, inputs, drop_connect_rate=None, mask=None):
        x = inputs
        if self.expand != 1:
            x = swish(self._bn0(self._expand_conv(inputs)))
        x = swish(self._bn1(self._depthwise_conv(x)))
        sq = F.adaptive_avg_pool2d(x, 1)
        sq = self._se_expand(swish(self._se_reduce(sq)))
        x = torch.sigmoid(sq) * x
        x = self._bn2(self._project_conv(x))
        if self.stride == 1 and self.inp == self.out:
            if drop_connect_rate and self.training:      # utils.drop_connect: one draw per sample, survivors scaled by 1 / keep
                keep = 1 - drop_connect_rate
                if mask is None:
                    mask = torch.floor(keep + torch.rand(x.shape[0], dtype=x.dtype, device=x.device)) / keep
                x = x * mask.view(-1, 1, 1, 1)
            x = x + inputs
        return x


class EfficientNetEncoder(nn.Module):
    def __init__(self, name: str, in_channels: int = 1):
        super().__init__()
        width, _, image_size = COEFFS[name]
        self.name = name
        self._conv_stem = SamePadConv2d(in_channels, round_filters(32, width), 3, 2, bias=False, image_size=image_size)
        self._bn0 = nn.BatchNorm2d(round_filters(32, width), momentum=BN_MOMENTUM, eps=BN_EPS)
        self._blocks = nn.ModuleList(MBConvBlock(k, s, e, i, o, image_size) for k, s, e, i, o in block_plan(name))
        last = block_plan(name)[-1][4]
        self._conv_head = SamePadConv2d(last, round_filters(1280, width), 1, bias=False, image_size=image_size)     # kept, never run
        self._bn1 = nn.BatchNorm2d(round_filters(1280, width), momentum=BN_MOMENTUM, eps=BN_EPS)
        self.drop_masks = None       # {block index: [n] tensor of 0 or 1 / keep}: the engine's draws replayed (tests)

    def forward(self, x):
        feats = [x]
        x = swish(self._bn0(self._conv_stem(x)))
        feats.append(x)
        s0, s1, s2, s3 = STAGE_IDXS[self.name]
        for i, blk in enumerate(self._blocks):
            rate = DROP_CONNECT * i / len(self._blocks)
            x = blk(x, rate, None if self.drop_masks is None else self.drop_masks.get(i))
            if i + 1 in (s0, s1, s2, s3):
                feats.append(x)
        return feats
