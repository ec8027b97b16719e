"""ORACLE (test infrastructure, never the product path): the encoders ``timm-resnest50d`` / ``timm-resnest101e`` of the reference's list
(README.md:57-76, tests/test_model_2d.py:36-44) as segmentation-models-pytorch 0.2.1 builds them (encoders/timm_resnest.py:
``ResNestEncoder(ResNet)`` over timm 0.4.12 - features [identity, conv1 + bn1 + act1, maxpool + layer1, layer2, layer3, layer4]).

Restated from the published sources - timm is not installed here (no network):
  * timm 0.4.12 models/resnet.py: ``ResNet(block=ResNestBottleneck, layers, stem_type='deep', stem_width, avg_down=True, base_width=64,
    cardinality=1, block_args=dict(radix=2, avd=True, avd_first=False))`` - deep stem conv1 = [3x3 / 2 (in -> sw), BN, ReLU, 3x3 (sw -> sw),
    BN, ReLU, 3x3 (sw -> 2 sw)], bn1, ReLU, MaxPool2d(3, 2, 1); ``downsample_avg`` = [AvgPool2d(2, stride, ceil_mode=True,
    count_include_pad=False) (identity at stride 1), Conv2d 1x1, BN]; resnest50d: layers (3, 4, 6, 3), stem_width 32; resnest101e:
    (3, 4, 23, 3), stem_width 64;
  * timm 0.4.12 models/resnest.py ``ResNestBottleneck``: conv1 1x1 -> bn1 -> ReLU -> conv2 = SplitAttnConv2d(3x3, stride 1 when avd
    carries the stride) -> avd_last = AvgPool2d(3, stride, padding=1) where stride > 1 (``is_first`` is never passed by make_blocks:
    False) -> conv3 1x1 -> bn3 -> + shortcut -> ReLU;
  * timm 0.4.12 models/layers/split_attn.py ``SplitAttnConv2d``: conv 3x3 (C -> radix C, groups = radix) -> bn0 -> ReLU -> sum of the radix
    splits -> global average pool -> fc1 (C -> max(C radix / 4, 32), bias) -> bn1 -> ReLU -> fc2 (-> radix C, bias) -> RadixSoftmax
    (softmax over the radix dimension) -> the attention-weighted sum of the splits.
Structural pins (tests/test_oracle_topology.py): timm's published parameter counts - resnest50d 27,483,240 and resnest101e 48,275,016
with the 3-channel stem and the 1000-way fc added back - and smp's out_channels (3, 64 | 128, 256, 512, 1024, 2048).  NOT pinnable by
counts: where the parameter-free pools sit (avd_last only at stride > 1; the 2x2 pool of the shortcut) - restated from the source text."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

LAYERS = {"timm-resnest50d": (3, 4, 6, 3), "timm-resnest101e": (3, 4, 23, 3)}
STEM_WIDTH = {"timm-resnest50d": 32, "timm-resnest101e": 64}
OUT_CHANNELS = {"timm-resnest50d": (1, 64, 256, 512, 1024, 2048), "timm-resnest101e": (1, 128, 256, 512, 1024, 2048)}
PUBLISHED_PARAMS = {"timm-resnest50d": 27_483_240, "timm-resnest101e": 48_275_016}


class SplitAttnConv2d(nn.Module):
    def __init__(self, channels: int, radix: int = 2, reduction_factor: int = 4):
        super().__init__()
        self.radix = radix
        mid, attn = channels * radix, max(channels * radix // reduction_factor, 32)
        self.conv = nn.Conv2d(channels, mid, 3, 1, 1, groups=radix, bias=False)
        self.bn0 = nn.BatchNorm2d(mid)
        self.fc1 = nn.Conv2d(channels, attn, 1)
        self.bn1 = nn.BatchNorm2d(attn)
        self.fc2 = nn.Conv2d(attn, mid, 1)

    def forward(self, x):
        x = F.relu(self.bn0(self.conv(x)))
        b, rc, h, w = x.shape
        xs = x.reshape(b, self.radix, rc // self.radix, h, w)
        gap = F.adaptive_avg_pool2d(xs.sum(1), 1)
        att = self.fc2(F.relu(self.bn1(self.fc1(gap))))
        att = torch.softmax(att.view(b, 1, self.radix, -1).transpose(1, 2), dim=1).reshape(b, -1)      # RadixSoftmax(radix, cardinality 1)
        return (xs * att.reshape(b, self.radix, rc // self.radix, 1, 1)).sum(1)


class ResNestBottleneck(nn.Module):
    def __init__(self, inplanes: int, planes: int, stride: int, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = SplitAttnConv2d(planes)
        self.avd_last = nn.AvgPool2d(3, stride, padding=1) if stride > 1 else None
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.conv2(out)
        if self.avd_last is not None:
            out = self.avd_last(out)
        out = self.bn3(self.conv3(out))
        return F.relu(out + (x if self.downsample is None else self.downsample(x)))


class ResNestEncoder(nn.Module):
    def __init__(self, name: str, in_channels: int = 1):
        super().__init__()
        sw = STEM_WIDTH[name]
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, sw, 3, 2, 1, bias=False), nn.BatchNorm2d(sw), nn.ReLU(inplace=True),
                                   nn.Conv2d(sw, sw, 3, 1, 1, bias=False), nn.BatchNorm2d(sw), nn.ReLU(inplace=True),
                                   nn.Conv2d(sw, 2 * sw, 3, 1, 1, bias=False))
        self.bn1 = nn.BatchNorm2d(2 * sw)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 2 * sw
        for i, (planes, blocks) in enumerate(zip((64, 128, 256, 512), LAYERS[name])):
            stride = 1 if i == 0 else 2
            down = None
            if stride != 1 or inplanes != planes * 4:      # downsample_avg
                pool = nn.Identity() if stride == 1 else nn.AvgPool2d(2, stride, ceil_mode=True, count_include_pad=False)
                down = nn.Sequential(pool, nn.Conv2d(inplanes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4))
            layer = [ResNestBottleneck(inplanes, planes, stride, down)]
            inplanes = planes * 4
            layer += [ResNestBottleneck(inplanes, planes, 1, None) for _ in range(1, blocks)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*layer))
        for m in self.modules():       # timm ResNet.init_weights
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        feats = [x]
        x = F.relu(self.bn1(self.conv1(x)))
        feats.append(x)
        x = self.layer1(self.maxpool(x))
        feats.append(x)
        for layer in (self.layer2, self.layer3, self.layer4):
            x = layer(x)
            feats.append(x)
        return feats
