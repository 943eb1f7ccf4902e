"""ResNet-18 / ResNet-50 built on the fused operators.

The reference instantiates ``torchvision.models.resnet18(num_classes=10)``
(/root/reference/src/main.py:13,49).  torchvision is not a dependency here; these
definitions reproduce its architecture (v1.5: the stride of a bottleneck sits on the 3x3
convolution), its initialisation and - importantly - its parameter and buffer names, so a
``state_dict`` saved by this framework loads into a stock torchvision ResNet and back.

The forward pass is written in terms of *raw convolution outputs plus pending
normalisation*: a convolution returns its output together with the batch statistics its
kernel epilogue gathered, and the consumer decides how the normalisation is applied.  On
GPU (``ops.hip``): in front of a 1x1 convolution with at most 128 input channels,
BatchNorm+ReLU is applied to the operand tiles inside the GEMM and weight-gradient kernels
and the normalised activation is never written; in front of the other convolutions it is
one streaming pass (read ``y``, write ``relu(bn(y))``) at HBM rate, because there the
in-kernel transform was measured to cost more than the pass it saves; block tails fuse
BatchNorm, the shortcut add and ReLU into one pass, the stem adds the max-pool.
"""

from __future__ import annotations

from typing import List, Optional, Type, Union

import torch
import torch.nn as nn

from .. import ops


class Conv2d(nn.Conv2d):
    """Bias-free convolution parameter holder (square kernel, symmetric stride/padding)."""

    def __init__(self, cin: int, cout: int, kernel_size: int, stride: int = 1, padding: int = 0):
        super().__init__(cin, cout, kernel_size, stride=stride, padding=padding, bias=False)

    @property
    def s(self) -> int:
        return self.stride[0]

    @property
    def p(self) -> int:
        return self.padding[0]


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: Optional[nn.Sequential] = None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # both consumers of x hand their input gradients to one kernel: the shortcut's (block tail or projection
        # convolution) is added inside the data-gradient kernel of conv1
        link = ops.residual_link(x)
        y1, s1 = ops.conv2d_stats(x, self.conv1.weight, self.conv1.s, self.conv1.p, self.training, link)
        y2, s2 = ops.bn_relu_conv2d(y1, s1, self.bn1, self.conv2.weight, 1, 1)
        if self.downsample is not None:
            dconv, dbn = self.downsample[0], self.downsample[1]
            yd, sd = ops.conv2d_stats(x, dconv.weight, dconv.s, dconv.p, self.training, link_out=link)
            return ops.bn_add_relu(y2, s2, self.bn2, yd, sd, dbn)
        return ops.bn_add_relu(y2, s2, self.bn2, x, link=link)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: Optional[nn.Sequential] = None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * self.expansion, 1)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.downsample = downsample

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        link = ops.residual_link(x)  # see BasicBlock.forward
        y1, s1 = ops.conv2d_stats(x, self.conv1.weight, 1, 0, self.training, link)
        y2, s2 = ops.bn_relu_conv2d(y1, s1, self.bn1, self.conv2.weight, self.conv2.s, 1)
        y3, s3 = ops.bn_relu_conv2d(y2, s2, self.bn2, self.conv3.weight, 1, 0)
        if self.downsample is not None:
            dconv, dbn = self.downsample[0], self.downsample[1]
            yd, sd = ops.conv2d_stats(x, dconv.weight, dconv.s, dconv.p, self.training, link_out=link)
            return ops.bn_add_relu(y3, s3, self.bn3, yd, sd, dbn)
        return ops.bn_add_relu(y3, s3, self.bn3, x, link=link)


class ResNet(nn.Module):
    def __init__(self, block: Type[Union[BasicBlock, Bottleneck]], layers: List[int], num_classes: int = 1000,
                 zero_init_residual: bool = False):
        super().__init__()
        self.inplanes = 64
        self.conv1 = Conv2d(3, 64, 7, 2, 3)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._make_layer(block, 64, layers[0], 1)
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0.0)
                elif isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0.0)

    def _make_layer(self, block, planes: int, blocks: int, stride: int) -> nn.Sequential:
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                Conv2d(self.inplanes, planes * block.expansion, 1, stride, 0),
                nn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y0, s0 = ops.conv2d_stats(x, self.conv1.weight, 2, 3, self.training)
        a = ops.bn_relu_maxpool(y0, s0, self.bn1)
        a = self.layer1(a)
        a = self.layer2(a)
        a = self.layer3(a)
        a = self.layer4(a)
        pooled = ops.global_avgpool(a)
        return ops.linear(pooled, self.fc.weight, self.fc.bias)


def resnet18(num_classes: int = 1000, **kw) -> ResNet:
    return ResNet(BasicBlock, [2, 2, 2, 2], num_classes, **kw)


def resnet34(num_classes: int = 1000, **kw) -> ResNet:
    return ResNet(BasicBlock, [3, 4, 6, 3], num_classes, **kw)


def resnet50(num_classes: int = 1000, **kw) -> ResNet:
    return ResNet(Bottleneck, [3, 4, 6, 3], num_classes, **kw)


def resnet101(num_classes: int = 1000, **kw) -> ResNet:
    return ResNet(Bottleneck, [3, 4, 23, 3], num_classes, **kw)
