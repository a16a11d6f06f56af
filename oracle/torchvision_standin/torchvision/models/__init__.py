"""ResNet-18 topology stand-in for `torchvision.models` (TEST INFRASTRUCTURE ONLY).

Restates the published ResNet-18 architecture (He et al. 2015, as packaged by
torchvision; version unpinned by the reference) with
torchvision's child names, which the reference reaches into:
  conv1, bn1, relu, maxpool, layer1..layer4, avgpool, fc
  (/root/reference/Quadtree_from scratch/models.py:221-243,
   /root/reference/resnet/models.py:76-101,148-149).

`resnet18(weights=...)` IGNORES `weights` and fetches nothing: the pretrained
ImageNet file is not available offline, parity is checked with deterministic
synthetic weights loaded through load_state_dict.
"""
import enum

import torch.nn as nn


class ResNet18_Weights(enum.Enum):
    IMAGENET1K_V1 = "IMAGENET1K_V1"
    DEFAULT = "IMAGENET1K_V1"


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = out + identity
        return self.relu(out)


class ResNet(nn.Module):
    def __init__(self, layers=(2, 2, 2, 2), num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes, 1, stride, bias=False),
                nn.BatchNorm2d(planes),
            )
        seq = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            seq.append(BasicBlock(planes, planes))
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(self.avgpool(x).flatten(1))


def resnet18(weights=None, progress=True, **kwargs):
    del weights, progress  # never fetched
    return ResNet((2, 2, 2, 2), **kwargs)
