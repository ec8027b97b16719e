"""ORACLE (test infrastructure, never the product path).

Pure-``torch.nn`` CPU fp32 restatement of the network the reference builds with
``smp.Unet(encoder_name="resnet34", encoder_weights=..., in_channels=1, classes=K)``
(reference call site: volume_segmantics/model/model_2d.py:15-16; struct dict built
at volume_segmantics/model/operations/vol_seg_2d_trainer.py:78-84).

The arithmetic itself lives in the third-party, un-vendored package
segmentation-models-pytorch ``^0.2.1`` (reference pyproject.toml:24) on top of
torchvision's ResNet-34.  Neither is installed in this image, so the topology is
restated from the published architecture:

* encoder  = torchvision ResNet-34 (BasicBlock [3,4,6,3]) without avgpool/fc, first
  conv patched to one input channel; features taken after the stem (H/2), layer1
  (H/4), layer2 (H/8), layer3 (H/16), layer4 (H/32);
* decoder  = 5 blocks of [nearest x2 upsample -> cat(skip) -> conv3x3+BN+ReLU ->
  conv3x3+BN+ReLU], decoder_channels (256,128,64,32,16), no centre block, no
  attention;
* head     = conv3x3(16 -> K) with bias, no activation.

State-dict keys are the ones smp produces (``encoder.conv1.weight`` ...
``segmentation_head.0.bias``) so checkpoints interchange with the reference.

PARITY PINNING: smp's own forward cannot be imported here (ModuleNotFoundError,
no network), and the reference's tests assert no numerics for this boundary
 -> the *network arithmetic* is "parity unpinned" against
smp.  What is pinned: (i) structural known answers - key names, shapes and the
published parameter count of ``smp.Unet('resnet34')`` (24,436,369 for 3-channel
input / 1 class, i.e. 24,430,097 + head for 1-channel input), see
tests/test_oracle_topology.py; (ii) the primitives are torch's own CPU fp32
kernels; (iii) everything the reference itself computes around the network
(prediction loop, merges, losses, LR finder) is pinned by goldens generated from
the reference's code (oracle/gen_goldens.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

RESNET34_LAYERS = (3, 4, 6, 3)
ENCODER_CHANNELS = (1, 64, 64, 128, 256, 512)  # feature channels incl. the (dropped) input
DECODER_CHANNELS = (256, 128, 64, 32, 16)


class BasicBlock(nn.Module):
    """torchvision BasicBlock: relu(bn1(conv1 x)) -> bn2(conv2 .) -> + identity -> relu."""

    def __init__(self, inplanes: int, planes: int, stride: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                nn.BatchNorm2d(planes),
            )

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet34Encoder(nn.Module):
    def __init__(self, in_channels: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for i, (planes, blocks) in enumerate(zip((64, 128, 256, 512), RESNET34_LAYERS)):
            stride = 1 if i == 0 else 2
            layer = [BasicBlock(inplanes, planes, stride)]
            inplanes = planes
            layer += [BasicBlock(inplanes, planes) for _ in range(1, blocks)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*layer))
        # torchvision ResNet init
        for m in self.modules():
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
        x = self.layer2(x)
        feats.append(x)
        x = self.layer3(x)
        feats.append(x)
        x = self.layer4(x)
        feats.append(x)
        return feats


class Conv2dReLU(nn.Sequential):
    """smp.base.modules.Conv2dReLU with use_batchnorm=True: keys '0' conv, '1' bn, '2' relu."""

    def __init__(self, cin: int, cout: int):
        super().__init__(
            nn.Conv2d(cin, cout, 3, padding=1, bias=False),
            nn.BatchNorm2d(cout),
            nn.ReLU(inplace=True),
        )


class DecoderBlock(nn.Module):
    def __init__(self, cin: int, cskip: int, cout: int):
        super().__init__()
        self.conv1 = Conv2dReLU(cin + cskip, cout)
        self.conv2 = Conv2dReLU(cout, cout)

    def forward(self, x, skip=None):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        return self.conv2(self.conv1(x))


class UnetDecoder(nn.Module):
    def __init__(self):
        super().__init__()
        enc = ENCODER_CHANNELS[1:][::-1]  # (512, 256, 128, 64, 64)
        cins = [enc[0]] + list(DECODER_CHANNELS[:-1])
        cskips = list(enc[1:]) + [0]
        self.blocks = nn.ModuleList(
            DecoderBlock(i, s, o) for i, s, o in zip(cins, cskips, DECODER_CHANNELS)
        )

    def forward(self, feats):
        feats = feats[1:][::-1]
        x, skips = feats[0], feats[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class OracleUnetResnet34(nn.Module):
    """The network behind ``self.model`` in the reference for U_NET + resnet34."""

    def __init__(self, in_channels: int = 1, classes: int = 2):
        super().__init__()
        self.encoder = ResNet34Encoder(in_channels)
        self.decoder = UnetDecoder()
        self.segmentation_head = nn.Sequential(nn.Conv2d(DECODER_CHANNELS[-1], classes, 3, padding=1))
        # smp initialisation: decoder kaiming_uniform(fan_in, relu), BN 1/0; head xavier_uniform, bias 0
        for m in self.decoder.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        for m in self.segmentation_head.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        return self.segmentation_head(self.decoder(self.encoder(x)))


def seeded_oracle(classes: int = 2, seed: int = 0, perturb_bn: bool = True) -> OracleUnetResnet34:
    """Deterministic random-init weights (no ImageNet download is possible here).

    BN affine parameters and running statistics are perturbed so that an eval-mode
    BN is not the identity."""
    g = torch.Generator().manual_seed(seed)
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    try:
        net = OracleUnetResnet34(1, classes)
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


def conv_macs_per_slice(h: int, w: int, classes: int) -> int:
    """Convolution multiply-accumulates of one forward pass."""
    macs = (h // 2) * (w // 2) * 64 * 49
    plan = [(64, 64, 4, 6)]  # (cin, cout, downscale, n convs)
    macs += 6 * (h // 4) * (w // 4) * 64 * 64 * 9
    for cin, cout, ds, n in ((64, 128, 8, 8), (128, 256, 16, 12), (256, 512, 32, 6)):
        px = (h // ds) * (w // ds)
        macs += px * cin * cout * 9 + px * cin * cout  # first conv (stride 2) + 1x1 shortcut
        macs += (n - 1) * px * cout * cout * 9
    del plan
    cins = (512 + 256, 256 + 128, 128 + 64, 64 + 64, 32)
    for i, (cin, cout) in enumerate(zip(cins, DECODER_CHANNELS)):
        px = (h >> (4 - i)) * (w >> (4 - i))
        macs += px * cin * cout * 9 + px * cout * cout * 9
    macs += h * w * 16 * classes * 9
    return macs
