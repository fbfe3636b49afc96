"""Residual block of the sparse ResUNet, counterpart of the reference's model/residual_block.py:25-103:
conv3 -> norm -> ReLU -> conv3 -> norm -> (+ input) -> ReLU.  Attribute names (conv1, norm1, conv2,
norm2) are the state-dict names of the reference checkpoints."""
import torch.nn as nn

from .. import minkowski as ME
from .common import get_norm

MEF = ME.MinkowskiFunctional


class BasicBlockBase(nn.Module):
    expansion = 1
    NORM_TYPE = "BN"

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, D=3):
        super().__init__()
        conv = dict(kernel_size=3, dilation=dilation, bias=False, dimension=D)
        self.conv1 = ME.MinkowskiConvolution(inplanes, planes, stride=stride, **conv)
        self.norm1 = get_norm(self.NORM_TYPE, planes, bn_momentum=bn_momentum, D=D)
        self.conv2 = ME.MinkowskiConvolution(planes, planes, stride=1, **conv)
        self.norm2 = get_norm(self.NORM_TYPE, planes, bn_momentum=bn_momentum, D=D)
        self.downsample = downsample

    def forward(self, x):
        y = MEF.relu(self.norm1(self.conv1(x)))
        y = self.norm2(self.conv2(y))
        y += x if self.downsample is None else self.downsample(x)
        return MEF.relu(y)


class BasicBlockBN(BasicBlockBase):
    NORM_TYPE = "BN"


class BasicBlockIN(BasicBlockBase):
    NORM_TYPE = "IN"


def get_block(norm_type, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, D=3):
    cls = {"BN": BasicBlockBN, "IN": BasicBlockIN}.get(norm_type)
    if cls is None:
        raise ValueError(f"Type {norm_type}, not defined")
    return cls(inplanes, planes, stride, dilation, downsample, bn_momentum, D)
