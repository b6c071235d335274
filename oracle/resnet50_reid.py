"""Oracle: ResNet-50-ReID forward/backward in plain fp32 torch (CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates ``Encoders.ResNet50ReID`` (Encoders.py:306-351) on top of a restatement of
torchvision's ResNet-50 v1.5 (third-party, absent from the image; reference call
sites Encoders.py:33,36).  Attribute names follow torchvision so ``state_dict()``
keys equal the reference's (``conv1.weight``, ``layer1.0.conv1.weight``,
``layer1.0.downsample.0.weight``, ``last_bn.weight`` ...).

The three ReID edits of the reference are reproduced exactly:
  * the stem ReLU is skipped: conv1 -> bn1 -> maxpool       (Encoders.py:332-335)
  * layer4[0].conv2 and layer4[0].downsample[0] run at stride 1 (Encoders.py:321-322)
  * head = global-avg-pool + global-max-pool, then BatchNorm1d(2048) (Encoders.py:341-350)
"""
import torch
from torch import nn


class Bottleneck(nn.Module):
    """torchvision v1.5 bottleneck: the stride sits on the 3x3 conv."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


def _make_layer(inplanes, planes, blocks, stride):
    downsample = None
    if stride != 1 or inplanes != planes * 4:
        downsample = nn.Sequential(
            nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
            nn.BatchNorm2d(planes * 4))
    layers = [Bottleneck(inplanes, planes, stride, downsample)]
    for _ in range(1, blocks):
        layers.append(Bottleneck(planes * 4, planes))
    return nn.Sequential(*layers)


class ResNet50ReID(nn.Module):
    """Oracle twin of Encoders.ResNet50ReID (Encoders.py:306-351).

    ``layers``/``width`` exist so tests can build a shallow/narrow net with the same
    topology; the defaults are the real ResNet-50.
    """

    def __init__(self, layers=(3, 4, 6, 3), width=64):
        super().__init__()
        w = width
        self.conv1 = nn.Conv2d(3, w, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(w)
        self.relu = nn.ReLU(inplace=False)     # held, never applied to the stem (Encoders.py:334)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = _make_layer(w, w, layers[0], 1)
        self.layer2 = _make_layer(w * 4, w * 2, layers[1], 2)
        self.layer3 = _make_layer(w * 8, w * 4, layers[2], 2)
        # torchvision builds layer4 at stride 2; the reference then forces conv2 and the
        # downsample conv of block 0 to stride 1 (Encoders.py:321-322).
        self.layer4 = _make_layer(w * 16, w * 8, layers[3], 1)
        self.global_avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.global_maxpool = nn.AdaptiveMaxPool2d((1, 1))
        self.last_bn = nn.BatchNorm1d(w * 32)
        # torchvision init: kaiming_normal_(fan_out, relu) on convs, BN weight=1 bias=0
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = self.conv1(x)
        x = self.bn1(x)
        # no ReLU here on purpose (Encoders.py:334)
        x = self.maxpool(x)
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        feature = getattr(self, "feature", "both")            # evaluateCleanATModels.py:335-340 (eval-script variant)
        if feature == "gap":
            x = self.global_avgpool(x)
        elif feature == "gmp":
            x = self.global_maxpool(x)
        else:
            x = self.global_avgpool(x) + self.global_maxpool(x)
        x = x.view(x.size(0), -1)
        return self.last_bn(x)


def conv_flops_per_image(height=256, width=128, layers=(3, 4, 6, 3), base=64):
    """2*MAC count of every conv for one image (8.107 GFLOP at 256x128)."""
    total = 0
    h, w = height // 2, width // 2
    total += 2 * h * w * base * 3 * 49
    h, w = h // 2, w // 2
    inpl = base
    for li, (planes, nblk, stride) in enumerate(
            zip((base, base * 2, base * 4, base * 8), layers, (1, 2, 2, 1))):
        for b in range(nblk):
            s = stride if b == 0 else 1
            total += 2 * h * w * inpl * planes                      # conv1 at input res
            ho, wo = h // s, w // s
            total += 2 * ho * wo * planes * planes * 9               # conv2
            total += 2 * ho * wo * planes * planes * 4               # conv3
            if b == 0:
                total += 2 * ho * wo * inpl * planes * 4             # downsample
            inpl = planes * 4
            h, w = ho, wo
    return total
