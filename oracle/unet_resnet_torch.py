"""ORACLE (test infrastructure, never the product path): ``smp.Unet(encoder_name=E, in_channels=1, classes=K)`` for the other
ResNet encoders of the reference's list (README.md:57-76, tests/test_model_2d.py:36-44) that the engine builds: resnet18,
resnet50 and resnext50_32x4d next to resnet34 (oracle/unet_resnet34_torch.py, whose blocks are reused here).

Restated from the published architectures - segmentation-models-pytorch ^0.2.1 and torchvision are not installed here
(oracle/unet_resnet34_torch.py explains the pinning situation; the same applies):
  * torchvision ResNet-18: BasicBlock x (2, 2, 2, 2); ResNet-50: Bottleneck x (3, 4, 6, 3), expansion 4, the stride on the 3x3
    convolution (the "v1.5" form torchvision ships), 1x1 projection shortcut where shape changes;
  * smp encoder out_channels: resnet18 (3, 64, 64, 128, 256, 512), resnet50 (3, 64, 256, 512, 1024, 2048);
  * the U-Net decoder / head as for resnet34; the U-Net++ decoder (UnetPlusPlusDecoder below) from smp 0.2.1's published source.
Structural pins (tests/test_oracle_topology.py): state-dict keys and shapes, and torchvision's published parameter counts -
resnet18 11,689,512, resnet34 21,797,672, resnet50 25,557,032, resnext50_32x4d 25,028,904 with the 3-channel stem and the 1000-way fc layer added back."""
from __future__ import annotations

import torch
import torch.nn as nn

from .efficientnet_torch import OUT_CHANNELS as EFFICIENTNET_OUT_CHANNELS, EfficientNetEncoder
from .resnest_torch import OUT_CHANNELS as RESNEST_OUT_CHANNELS, ResNestEncoder
from .unet_resnet34_torch import DECODER_CHANNELS, BasicBlock, DecoderBlock

LAYERS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3), "resnet50": (3, 4, 6, 3), "resnext50_32x4d": (3, 4, 6, 3)}
OUT_CHANNELS = {"resnet18": (1, 64, 64, 128, 256, 512), "resnet34": (1, 64, 64, 128, 256, 512), "resnet50": (1, 64, 256, 512, 1024, 2048),
                "resnext50_32x4d": (1, 64, 256, 512, 1024, 2048)}
FC_PARAMS = {"resnet18": 512 * 1000 + 1000, "resnet34": 512 * 1000 + 1000, "resnet50": 2048 * 1000 + 1000,
             "resnext50_32x4d": 2048 * 1000 + 1000}
TORCHVISION_PARAMS = {"resnet18": 11_689_512, "resnet34": 21_797_672, "resnet50": 25_557_032, "resnext50_32x4d": 25_028_904}
BOTTLENECK = ("resnet50", "resnext50_32x4d")
GROUPS = {"resnext50_32x4d": (32, 4)}      # torchvision resnext50_32x4d: groups = 32, width_per_group = 4


class Bottleneck(nn.Module):
    """torchvision Bottleneck: relu(bn1(conv1x1)) -> relu(bn2(conv3x3, stride)) -> bn3(conv1x1 x4) -> + identity -> relu.
    ResNeXt: the 3x3 convolution is grouped and the inner width is int(planes * base_width / 64) * groups."""
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, groups: int = 1, base_width: int = 64):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class ResNetEncoder(nn.Module):
    def __init__(self, name: str, in_channels: int = 1):
        super().__init__()
        exp = 4 if name in BOTTLENECK else 1
        if name in BOTTLENECK:
            groups, base_width = GROUPS.get(name, (1, 64))
            block = lambda i, p, s=1: Bottleneck(i, p, s, groups, base_width)   # noqa: E731
        else:
            block = BasicBlock
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for i, (planes, blocks) in enumerate(zip((64, 128, 256, 512), LAYERS[name])):
            layer = [block(inplanes, planes, 1 if i == 0 else 2)]
            inplanes = planes * exp
            layer += [block(inplanes, planes) for _ in range(1, blocks)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*layer))
        for m in self.modules():   # torchvision ResNet init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        feats = [x]
        x = self.relu(self.bn1(self.conv1(x)))
        feats.append(x)
        x = self.layer1(self.maxpool(x))
        feats.append(x)
        for l in (self.layer2, self.layer3, self.layer4):
            x = l(x)
            feats.append(x)
        return feats


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels):
        super().__init__()
        enc = encoder_channels[1:][::-1]
        cins = [enc[0]] + list(DECODER_CHANNELS[:-1])
        cskips = list(enc[1:]) + [0]
        self.blocks = nn.ModuleList(DecoderBlock(i, s, o) for i, s, o in zip(cins, cskips, DECODER_CHANNELS))

    def forward(self, feats):
        feats = feats[1:][::-1]
        x, skips = feats[0], feats[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class UnetPlusPlusDecoder(nn.Module):
    """smp.UnetPlusPlus's decoder (segmentation-models-pytorch 0.2.1, decoders/unetplusplus/decoder.py), restated: dense nested
    skips.  With the encoder features reversed (deepest first) f[0..4], in_channels = [C(f0)] + decoder_channels[:-1],
    skip_channels = [C(f1), C(f2), C(f3), C(f4), 0]:
      blocks["x_0_l"] = DecoderBlock(in_channels[l], skip_channels[l] * (l + 1), decoder_channels[l])
      blocks["x_d_l"] = DecoderBlock(skip_channels[l - 1], skip_channels[l] * (l + 1 - d), skip_channels[l])     (0 < d <= l)
      blocks["x_0_4"] = DecoderBlock(in_channels[4], 0, decoder_channels[4])
    registered in that (l outer, d inner) order; forward: first x_d_d(f[d], f[d+1]) for d = 0..3, then by increasing l - d
    x_d_l(x_d_(l-1), cat(x_(d+1)_l, .., x_l_l, f[l+1])), finally x_0_4(x_0_3)."""

    def __init__(self, encoder_channels):
        super().__init__()
        enc = list(encoder_channels[1:][::-1])
        self.in_channels = [enc[0]] + list(DECODER_CHANNELS[:-1])
        self.skip_channels = enc[1:] + [0]
        self.out_channels = DECODER_CHANNELS
        blocks = {}
        for l in range(len(self.in_channels) - 1):
            for d in range(l + 1):
                if d == 0:
                    in_ch, skip_ch, out_ch = self.in_channels[l], self.skip_channels[l] * (l + 1), self.out_channels[l]
                else:
                    out_ch, skip_ch, in_ch = self.skip_channels[l], self.skip_channels[l] * (l + 1 - d), self.skip_channels[l - 1]
                blocks[f"x_{d}_{l}"] = DecoderBlock(in_ch, skip_ch, out_ch)
        blocks[f"x_0_{len(self.in_channels) - 1}"] = DecoderBlock(self.in_channels[-1], 0, self.out_channels[-1])
        self.blocks = nn.ModuleDict(blocks)
        self.depth = len(self.in_channels) - 1

    def forward(self, feats):
        f = feats[1:][::-1]
        x = {}
        for layer in range(len(self.in_channels) - 1):
            for d in range(self.depth - layer):
                if layer == 0:
                    x[f"x_{d}_{d}"] = self.blocks[f"x_{d}_{d}"](f[d], f[d + 1])
                else:
                    l = d + layer
                    cat = torch.cat([x[f"x_{i}_{l}"] for i in range(d + 1, l + 1)] + [f[l + 1]], dim=1)
                    x[f"x_{d}_{l}"] = self.blocks[f"x_{d}_{l}"](x[f"x_{d}_{l - 1}"], cat)
        return self.blocks[f"x_0_{self.depth}"](x[f"x_0_{self.depth - 1}"])


class LinknetDecoder(nn.Module):
    """smp.Linknet's decoder (segmentation-models-pytorch 0.2.1, decoders/linknet/decoder.py), restated.  channels = encoder
    features without the input, deepest first, + [prefinal_channels = 32]; block i =
      Sequential(Conv2dReLU(in, in // 4, 1), TransposeX2(in // 4, in // 4), Conv2dReLU(in // 4, out, 1))
    with Conv2dReLU = Sequential(Conv2d(bias=False), BatchNorm2d, ReLU) and TransposeX2 = Sequential(ConvTranspose2d(kernel 4,
    stride 2, padding 1 - with its bias), BatchNorm2d, ReLU); forward: x = block(x) (+ skip: the next-shallower feature) for
    the first four blocks.  Keys: decoder.blocks.{i}.block.{0,1,2}.{0,1}.*."""

    def __init__(self, encoder_channels, prefinal_channels: int = 32):
        super().__init__()
        ch = list(encoder_channels[1:][::-1]) + [prefinal_channels]

        def conv_relu(i, o):
            return nn.Sequential(nn.Conv2d(i, o, 1, bias=False), nn.BatchNorm2d(o), nn.ReLU(inplace=True))

        class Block(nn.Module):
            def __init__(self, i, o):
                super().__init__()
                self.block = nn.Sequential(
                    conv_relu(i, i // 4),
                    nn.Sequential(nn.ConvTranspose2d(i // 4, i // 4, kernel_size=4, stride=2, padding=1), nn.BatchNorm2d(i // 4), nn.ReLU(inplace=True)),
                    conv_relu(i // 4, o))

            def forward(self, x, skip=None):
                x = self.block(x)
                return x if skip is None else x + skip

        self.blocks = nn.ModuleList(Block(ch[i], ch[i + 1]) for i in range(5))

    def forward(self, feats):
        f = feats[1:][::-1]
        x, skips = f[0], f[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class FPNDecoder(nn.Module):
    """smp.FPN's decoder (segmentation-models-pytorch 0.2.1, decoders/fpn/decoder.py), restated: pyramid_channels 256,
    segmentation_channels 128, merge_policy "add", dropout 0.2.
      p5 = Conv2d(C5, 256, 1); p4 / p3 / p2 = FPNBlock: F.interpolate(x, scale_factor=2, mode="nearest") + skip_conv(skip)
      (skip_conv = Conv2d(C_k, 256, 1), biased);
      seg_blocks[i] (on p5, p4, p3, p2; n_upsamples 3, 2, 1, 0) = SegmentationBlock: Conv3x3GNReLU(256, 128, upsample = n > 0)
      followed by n - 1 more Conv3x3GNReLU(128, 128, upsample=True); Conv3x3GNReLU = Conv2d(3x3, bias=False) +
      GroupNorm(32, out) + ReLU (+ F.interpolate(scale_factor=2, mode="bilinear", align_corners=True));
      x = sum of the four outputs; Dropout2d(p=0.2).
    ``mask`` (tests): a fixed (N, 128) Dropout2d mask (0 or 1/(1-p)) instead of torch's own draw, so that the engine's draw can
    be replayed here."""

    class _GN(nn.Module):
        def __init__(self, i, o, upsample):
            super().__init__()
            self.upsample = upsample
            self.block = nn.Sequential(nn.Conv2d(i, o, 3, 1, 1, bias=False), nn.GroupNorm(32, o), nn.ReLU(inplace=True))

        def forward(self, x):
            x = self.block(x)
            return torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True) if self.upsample else x

    class _Seg(nn.Module):
        def __init__(self, i, o, n_upsamples):
            super().__init__()
            blocks = [FPNDecoder._GN(i, o, bool(n_upsamples))]
            blocks += [FPNDecoder._GN(o, o, True) for _ in range(1, n_upsamples)]
            self.block = nn.Sequential(*blocks)

        def forward(self, x):
            return self.block(x)

    class _Lat(nn.Module):
        def __init__(self, pyramid, skip):
            super().__init__()
            self.skip_conv = nn.Conv2d(skip, pyramid, 1)

        def forward(self, x, skip):
            return torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest") + self.skip_conv(skip)

    def __init__(self, encoder_channels, pyramid_channels: int = 256, segmentation_channels: int = 128, dropout: float = 0.2):
        super().__init__()
        ch = list(encoder_channels[::-1])            # deepest first
        self.p5 = nn.Conv2d(ch[0], pyramid_channels, 1)
        self.p4 = FPNDecoder._Lat(pyramid_channels, ch[1])
        self.p3 = FPNDecoder._Lat(pyramid_channels, ch[2])
        self.p2 = FPNDecoder._Lat(pyramid_channels, ch[3])
        self.seg_blocks = nn.ModuleList(FPNDecoder._Seg(pyramid_channels, segmentation_channels, n) for n in (3, 2, 1, 0))
        self.dropout = nn.Dropout2d(p=dropout, inplace=True)
        self.mask = None

    def forward(self, feats):
        c2, c3, c4, c5 = feats[-4:]
        p5 = self.p5(c5)
        p4 = self.p4(p5, c4)
        p3 = self.p3(p4, c3)
        p2 = self.p2(p3, c2)
        x = sum(blk(p) for blk, p in zip(self.seg_blocks, (p5, p4, p3, p2)))
        if self.mask is not None and self.training:
            return x * self.mask[:, :, None, None]
        return self.dropout(x)


def replace_strides_with_dilation(module: nn.Module, dilation_rate: int) -> None:
    """smp.encoders._utils.replace_strides_with_dilation (restated): every Conv2d of the stage gets stride 1, dilation =
    padding-scale = dilation_rate ((k // 2) * rate of padding)."""
    for mod in module.modules():
        if isinstance(mod, nn.Conv2d):
            mod.stride = (1, 1)
            mod.dilation = (dilation_rate, dilation_rate)
            kh, _ = mod.kernel_size
            mod.padding = ((kh // 2) * dilation_rate, (kh // 2) * dilation_rate)
            if hasattr(mod, "static_pad"):      # smp: "Kostyl for EfficientNet" - mod.static_padding = nn.Identity()
                mod.static_pad = (0, 0, 0, 0)


class SeparableConv2d(nn.Sequential):
    """smp.base.modules.SeparableConv2d: depthwise Conv2d(groups = in) then pointwise 1x1 Conv2d, no norm in between."""

    def __init__(self, i, o, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__(nn.Conv2d(i, i, kernel_size, stride=stride, padding=padding, dilation=dilation, groups=i, bias=False),
                         nn.Conv2d(i, o, 1, bias=bias))


class DeepLabV3PlusDecoder(nn.Module):
    """smp.DeepLabV3Plus's decoder (segmentation-models-pytorch 0.2.1, decoders/deeplabv3/decoder.py), restated: out_channels 256,
    atrous_rates (12, 24, 36), output_stride 16.
      aspp = Sequential(ASPP(C5, 256, rates, separable=True), SeparableConv2d(256, 256, 3, padding=1, bias=False), BatchNorm2d, ReLU)
      ASPP.convs = [Conv1x1 + BN + ReLU, 3 x (SeparableConv2d(C5, 256, 3, padding=r, dilation=r, bias=False) + BN + ReLU),
                    ASPPPooling = Sequential(AdaptiveAvgPool2d(1), Conv1x1, BN, ReLU) + F.interpolate(size, "bilinear",
                    align_corners=False)]; ASPP.project = Conv1x1(5 * 256 -> 256) + BN + ReLU + Dropout(0.5)
      up = UpsamplingBilinear2d(scale_factor=4); block1 = Conv1x1(C2 -> 48) + BN + ReLU; block2 = SeparableConv2d(304, 256, 3,
      padding=1, bias=False) + BN + ReLU; forward: block2(cat(up(aspp(c5)), block1(c2))).
    ``drop`` (tests): a callable replacing the Dropout(0.5) draw (the engine's mask, replayed)."""

    class _Pool(nn.Sequential):
        def __init__(self, i, o):
            super().__init__(nn.AdaptiveAvgPool2d(1), nn.Conv2d(i, o, 1, bias=False), nn.BatchNorm2d(o), nn.ReLU())

        def forward(self, x):
            size = x.shape[-2:]
            for mod in self:
                x = mod(x)
            return torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=False)

    class _ASPP(nn.Module):
        def __init__(self, i, o, rates, separable=True):
            super().__init__()
            mods = [nn.Sequential(nn.Conv2d(i, o, 1, bias=False), nn.BatchNorm2d(o), nn.ReLU())]
            if separable:      # ASPPSeparableConv
                mods += [nn.Sequential(SeparableConv2d(i, o, 3, padding=r, dilation=r, bias=False), nn.BatchNorm2d(o), nn.ReLU()) for r in rates]
            else:              # ASPPConv: a dense dilated 3x3
                mods += [nn.Sequential(nn.Conv2d(i, o, 3, padding=r, dilation=r, bias=False), nn.BatchNorm2d(o), nn.ReLU()) for r in rates]
            mods.append(DeepLabV3PlusDecoder._Pool(i, o))
            self.convs = nn.ModuleList(mods)
            self.project = nn.Sequential(nn.Conv2d(5 * o, o, 1, bias=False), nn.BatchNorm2d(o), nn.ReLU(), nn.Dropout(0.5))
            self.drop = None

        def forward(self, x):
            res = torch.cat([conv(x) for conv in self.convs], dim=1)
            if self.drop is not None and self.training:
                return self.drop(self.project[2](self.project[1](self.project[0](res))))
            return self.project(res)

    def __init__(self, encoder_channels, out_channels: int = 256, atrous_rates=(12, 24, 36)):
        super().__init__()
        self.aspp = nn.Sequential(DeepLabV3PlusDecoder._ASPP(encoder_channels[-1], out_channels, atrous_rates),
                                  SeparableConv2d(out_channels, out_channels, 3, padding=1, bias=False), nn.BatchNorm2d(out_channels), nn.ReLU())
        self.up = nn.UpsamplingBilinear2d(scale_factor=4)
        self.block1 = nn.Sequential(nn.Conv2d(encoder_channels[-4], 48, 1, bias=False), nn.BatchNorm2d(48), nn.ReLU())
        self.block2 = nn.Sequential(SeparableConv2d(48 + out_channels, out_channels, 3, padding=1, bias=False), nn.BatchNorm2d(out_channels), nn.ReLU())

    def forward(self, feats):
        a = self.up(self.aspp(feats[-1]))
        h = self.block1(feats[-4])
        return self.block2(torch.cat([a, h], dim=1))


class DeepLabV3Decoder(nn.Sequential):
    """smp.DeepLabV3's decoder (decoders/deeplabv3/decoder.py, restated): Sequential(ASPP(C5, 256, (12, 24, 36)) with dense dilated 3x3
    branches, Conv2d(256, 256, 3, padding=1, bias=False), BatchNorm2d, ReLU) on the last feature (encoder at output stride 8)."""

    def __init__(self, encoder_channels, out_channels: int = 256, atrous_rates=(12, 24, 36)):
        super().__init__(DeepLabV3PlusDecoder._ASPP(encoder_channels[-1], out_channels, atrous_rates, separable=False),
                         nn.Conv2d(out_channels, out_channels, 3, padding=1, bias=False), nn.BatchNorm2d(out_channels), nn.ReLU())

    def forward(self, feats):
        return super().forward(feats[-1])


class MAnetDecoder(nn.Module):
    """smp.MAnet's decoder (segmentation-models-pytorch 0.2.1, decoders/manet/decoder.py), restated: decoder_channels (256, 128, 64,
    32, 16), reduction 16, pab_channels 64.  center = PAB(C5); blocks[i] = MFAB(in, skip, out) where a skip exists, the U-Net
    DecoderBlock for the last level.  PAB and MFAB follow smp line by line - including PAB's `reshape(b, C, h, w)` of the
    (b, hw, C) attention product without a transpose, and the softmax over ALL hw * hw entries of a sample."""

    class PAB(nn.Module):
        def __init__(self, in_channels, pab_channels=64):
            super().__init__()
            self.in_channels = in_channels
            self.top_conv = nn.Conv2d(in_channels, pab_channels, kernel_size=1)
            self.center_conv = nn.Conv2d(in_channels, pab_channels, kernel_size=1)
            self.bottom_conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, padding=1)
            self.map_softmax = nn.Softmax(dim=1)
            self.out_conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, padding=1)

        def forward(self, x):
            bsize, _, h, w = x.size()
            x_top = self.top_conv(x).flatten(2)
            x_center = self.center_conv(x).flatten(2).transpose(1, 2)
            x_bottom = self.bottom_conv(x).flatten(2).transpose(1, 2)
            sp_map = torch.matmul(x_center, x_top)
            sp_map = self.map_softmax(sp_map.view(bsize, -1)).view(bsize, h * w, h * w)
            sp_map = torch.matmul(sp_map, x_bottom)
            sp_map = sp_map.reshape(bsize, self.in_channels, h, w)
            return self.out_conv(x + sp_map)

    class MFAB(nn.Module):
        def __init__(self, in_channels, skip_channels, out_channels, reduction=16):
            super().__init__()

            def cbr(i, o, k):
                return nn.Sequential(nn.Conv2d(i, o, k, padding=k // 2, bias=False), nn.BatchNorm2d(o), nn.ReLU(inplace=True))

            def se(c):
                r = max(1, c // reduction)
                return nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(c, r, 1), nn.ReLU(inplace=True), nn.Conv2d(r, c, 1), nn.Sigmoid())

            self.hl_conv = nn.Sequential(cbr(in_channels, in_channels, 3), cbr(in_channels, skip_channels, 1))
            self.SE_ll = se(skip_channels)
            self.SE_hl = se(skip_channels)
            self.conv1 = cbr(skip_channels + skip_channels, out_channels, 3)
            self.conv2 = cbr(out_channels, out_channels, 3)

        def forward(self, x, skip=None):
            x = self.hl_conv(x)
            x = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")
            attention_hl = self.SE_hl(x)
            if skip is not None:
                attention_hl = attention_hl + self.SE_ll(skip)
                x = x * attention_hl
                x = torch.cat([x, skip], dim=1)
            return self.conv2(self.conv1(x))

    def __init__(self, encoder_channels):
        super().__init__()
        enc = list(encoder_channels[1:][::-1])
        ins = [enc[0]] + list(DECODER_CHANNELS[:-1])
        skips = enc[1:] + [0]
        self.center = MAnetDecoder.PAB(enc[0])
        self.blocks = nn.ModuleList(MAnetDecoder.MFAB(i, s_, o) if s_ > 0 else DecoderBlock(i, s_, o) for i, s_, o in zip(ins, skips, DECODER_CHANNELS))

    def forward(self, feats):
        f = feats[1:][::-1]
        x, skips = self.center(f[0]), f[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class PANDecoder(nn.Module):
    """smp.PAN's decoder (segmentation-models-pytorch 0.2.1, decoders/pan/decoder.py), restated: decoder_channels 32, upscale_mode
    "bilinear" (align_corners=True).  ConvBnRelu = Conv2d (WITH bias) + BatchNorm2d (+ ReLU); FPABlock on the last feature, GAUBlock
    on the three before it."""

    class ConvBnRelu(nn.Module):
        def __init__(self, i, o, k, padding=0, add_relu=True):
            super().__init__()
            self.conv = nn.Conv2d(i, o, k, padding=padding)
            self.bn = nn.BatchNorm2d(o)
            self.add_relu = add_relu

        def forward(self, x):
            x = self.bn(self.conv(x))
            return torch.relu(x) if self.add_relu else x

    class FPABlock(nn.Module):
        def __init__(self, i, o):
            super().__init__()
            C = PANDecoder.ConvBnRelu
            self.branch1 = nn.Sequential(nn.AdaptiveAvgPool2d(1), C(i, o, 1))
            self.mid = nn.Sequential(C(i, o, 1))
            self.down1 = nn.Sequential(nn.MaxPool2d(2, 2), C(i, 1, 7, padding=3))
            self.down2 = nn.Sequential(nn.MaxPool2d(2, 2), C(1, 1, 5, padding=2))
            self.down3 = nn.Sequential(nn.MaxPool2d(2, 2), C(1, 1, 3, padding=1), C(1, 1, 3, padding=1))
            self.conv2 = C(1, 1, 5, padding=2)
            self.conv1 = C(1, 1, 7, padding=3)

        def forward(self, x):
            h, w = x.size(2), x.size(3)
            up = dict(mode="bilinear", align_corners=True)
            F_ = torch.nn.functional
            b1 = F_.interpolate(self.branch1(x), size=(h, w), **up)
            mid = self.mid(x)
            x1 = self.down1(x); x2 = self.down2(x1); x3 = self.down3(x2)
            x3 = F_.interpolate(x3, size=(h // 4, w // 4), **up)
            x = self.conv2(x2) + x3
            x = F_.interpolate(x, size=(h // 2, w // 2), **up)
            x = x + self.conv1(x1)
            x = F_.interpolate(x, size=(h, w), **up)
            return torch.mul(x, mid) + b1

    class GAUBlock(nn.Module):
        def __init__(self, i, o):
            super().__init__()
            C = PANDecoder.ConvBnRelu
            self.conv1 = nn.Sequential(nn.AdaptiveAvgPool2d(1), C(o, o, 1, add_relu=False), nn.Sigmoid())
            self.conv2 = C(i, o, 3, padding=1)

        def forward(self, x, y):
            y_up = torch.nn.functional.interpolate(y, size=(x.size(2), x.size(3)), mode="bilinear", align_corners=True)
            return y_up + torch.mul(self.conv2(x), self.conv1(y))

    def __init__(self, encoder_channels, decoder_channels: int = 32):
        super().__init__()
        self.fpa = PANDecoder.FPABlock(encoder_channels[-1], decoder_channels)
        self.gau3 = PANDecoder.GAUBlock(encoder_channels[-2], decoder_channels)
        self.gau2 = PANDecoder.GAUBlock(encoder_channels[-3], decoder_channels)
        self.gau1 = PANDecoder.GAUBlock(encoder_channels[-4], decoder_channels)

    def forward(self, feats):
        x5 = self.fpa(feats[-1])
        x4 = self.gau3(feats[-2], x5)
        x3 = self.gau2(feats[-3], x4)
        return self.gau1(feats[-4], x3)


class OracleUnet(nn.Module):
    def __init__(self, encoder_name: str, in_channels: int = 1, classes: int = 2, topology: str = "unet"):
        super().__init__()
        if encoder_name in RESNEST_OUT_CHANNELS:          # smp's ResNestEncoder over timm 0.4.12 (oracle/resnest_torch.py)
            assert topology not in ("deeplabv3", "deeplabv3plus", "pan"), "ResNeSt under the dilating decoders is not restated"
            self.encoder = ResNestEncoder(encoder_name, in_channels)
            out_channels = RESNEST_OUT_CHANNELS[encoder_name]
        elif encoder_name in EFFICIENTNET_OUT_CHANNELS:   # smp's EfficientNetEncoder (oracle/efficientnet_torch.py)
            assert topology != "linknet", "the EfficientNet encoders are not restated under smp.Linknet"
            self.encoder = EfficientNetEncoder(encoder_name, in_channels)
            out_channels = EFFICIENTNET_OUT_CHANNELS[encoder_name]
        else:
            self.encoder = ResNetEncoder(encoder_name, in_channels)
            out_channels = OUT_CHANNELS[encoder_name]
        self.decoder = {"unet": UnetDecoder, "unetplusplus": UnetPlusPlusDecoder, "linknet": LinknetDecoder,
                        "fpn": FPNDecoder, "deeplabv3plus": DeepLabV3PlusDecoder, "deeplabv3": DeepLabV3Decoder,
                        "manet": MAnetDecoder, "pan": PANDecoder}[topology](out_channels)
        if topology == "pan" and encoder_name in EFFICIENTNET_OUT_CHANNELS:
            from .efficientnet_torch import STAGE_IDXS
            replace_strides_with_dilation(self.encoder._blocks[STAGE_IDXS[encoder_name][2]:], 2)
        elif topology == "pan":             # encoder_dilation=True: make_dilated(stage_list=[5], dilation_list=[2])
            replace_strides_with_dilation(self.encoder.layer4, 2)
        if topology == "deeplabv3" and encoder_name in EFFICIENTNET_OUT_CHANNELS:     # get_stages()[4] / [5] = _blocks[s1:s2] / _blocks[s2:]
            from .efficientnet_torch import STAGE_IDXS
            _, s1, s2, _ = STAGE_IDXS[encoder_name]
            replace_strides_with_dilation(self.encoder._blocks[s1:s2], 2)
            replace_strides_with_dilation(self.encoder._blocks[s2:], 4)
        elif topology == "deeplabv3":       # encoder_output_stride = 8: make_dilated(stage_list=[4, 5], dilation_list=[2, 4])
            replace_strides_with_dilation(self.encoder.layer3, 2)
            replace_strides_with_dilation(self.encoder.layer4, 4)
        if topology == "deeplabv3plus":     # encoder_output_stride = 16: encoder.make_dilated(stage_list=[5], dilation_list=[2])
            if encoder_name in EFFICIENTNET_OUT_CHANNELS:     # get_stages()[5] = self._blocks[stage_idxs[2]:]
                from .efficientnet_torch import STAGE_IDXS
                replace_strides_with_dilation(self.encoder._blocks[STAGE_IDXS[encoder_name][2]:], 2)
            else:
                replace_strides_with_dilation(self.encoder.layer4, 2)
        if topology == "linknet":     # SegmentationHead(in_channels=32, out_channels=classes, kernel_size=1)
            self.segmentation_head = nn.Sequential(nn.Conv2d(32, classes, 1))
        elif topology == "pan":             # SegmentationHead(in_channels=32, out_channels=classes, kernel_size=3, upsampling=4)
            self.segmentation_head = nn.Sequential(nn.Conv2d(32, classes, 3, padding=1), nn.UpsamplingBilinear2d(scale_factor=4))
        elif topology == "deeplabv3":       # SegmentationHead(in_channels=256, out_channels=classes, kernel_size=1, upsampling=8)
            self.segmentation_head = nn.Sequential(nn.Conv2d(256, classes, 1), nn.UpsamplingBilinear2d(scale_factor=8))
        elif topology == "deeplabv3plus":   # SegmentationHead(in_channels=256, out_channels=classes, kernel_size=1, upsampling=4)
            self.segmentation_head = nn.Sequential(nn.Conv2d(256, classes, 1), nn.UpsamplingBilinear2d(scale_factor=4))
        elif topology == "fpn":       # SegmentationHead(in_channels=128, out_channels=classes, kernel_size=1, upsampling=4)
            self.segmentation_head = nn.Sequential(nn.Conv2d(128, classes, 1), nn.UpsamplingBilinear2d(scale_factor=4))
        else:
            self.segmentation_head = nn.Sequential(nn.Conv2d(DECODER_CHANNELS[-1], classes, 3, padding=1))
        for m in self.decoder.modules():   # smp initialisation (nn.ConvTranspose2d is not an nn.Conv2d: torch's default stays)
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        nn.init.xavier_uniform_(self.segmentation_head[0].weight)
        nn.init.constant_(self.segmentation_head[0].bias, 0)

    def forward(self, x):
        return self.segmentation_head(self.decoder(self.encoder(x)))


def seeded_oracle_unet(encoder_name: str, classes: int = 2, seed: int = 0, perturb_bn: bool = True, topology: str = "unet") -> OracleUnet:
    g = torch.Generator().manual_seed(seed)
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    try:
        net = OracleUnet(encoder_name, 1, classes, topology)
    finally:
        torch.set_rng_state(state)
    if perturb_bn:
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.weight.copy_(1.0 + 0.2 * (torch.rand(m.weight.shape, generator=g) - 0.5))
                    m.bias.copy_(0.2 * (torch.rand(m.bias.shape, generator=g) - 0.5))
                    m.running_mean.copy_(0.2 * (torch.rand(m.bias.shape, generator=g) - 0.5))
                    m.running_var.copy_(0.75 + 0.5 * torch.rand(m.bias.shape, generator=g))
    return net
