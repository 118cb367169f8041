"""Sparse U-Net building blocks of the PointGroup backbone on d3net_amd.minkowski.

Module and attribute names (`conv_branch`, `downsample`, `blocks`, `conv`, `u`, `deconv`, `blocks_tail`,
`block{i}`) reproduce the reference's module tree so its checkpoints' state-dict keys load unchanged
(reference: model/common.py:22-53 ResidualBlock, :56-70 VGGBlock, :73-118 UBlock).
"""
from collections import OrderedDict

import torch.nn as nn

from . import minkowski as ME


def _norm_relu_conv(norm_fn, cin, cout, conv_cls=ME.MinkowskiConvolution, **conv_kw):
    """[norm, relu, conv] -- the pre-activation unit every block of the U-Net is made of."""
    return [norm_fn(cin), ME.MinkowskiReLU(inplace=True), conv_cls(cin, cout, bias=False, **conv_kw)]


class ResidualBlock(nn.Module):
    """x + conv3(relu(bn(conv3(relu(bn(x)))))), with a 1x1 conv on the skip when the width changes."""

    def __init__(self, in_channels, out_channels, dimension, norm_fn=None):
        super().__init__()
        norm_fn = norm_fn or ME.MinkowskiBatchNorm
        self.downsample = None
        if in_channels != out_channels:
            self.downsample = nn.Sequential(
                ME.MinkowskiConvolution(in_channels, out_channels, kernel_size=1, bias=False, dimension=dimension))
        self.conv_branch = nn.Sequential(
            *_norm_relu_conv(norm_fn, in_channels, out_channels, kernel_size=3, dimension=dimension),
            *_norm_relu_conv(norm_fn, out_channels, out_channels, kernel_size=3, dimension=dimension))

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        out = self.conv_branch(x)
        out += skip
        return out


class VGGBlock(nn.Module):
    def __init__(self, in_channels, out_channels, dimension, norm_fn=None):
        super().__init__()
        norm_fn = norm_fn or ME.MinkowskiBatchNorm
        self.conv_layers = nn.Sequential(
            *_norm_relu_conv(norm_fn, in_channels, out_channels, kernel_size=3, dimension=dimension))

    def forward(self, x):
        return self.conv_layers(x)


class UBlock(nn.Module):
    """One level of the U-Net: `block_reps` blocks, then (if deeper levels exist) stride-2 down conv,
    the recursive UBlock, stride-2 transposed conv back, concat with the skip, `block_reps` tail blocks."""

    def __init__(self, nPlanes, norm_fn, block_reps, block):
        super().__init__()
        self.nPlanes = list(nPlanes)
        self.D = 3
        c = self.nPlanes[0]
        self.blocks = nn.Sequential(OrderedDict(
            ("block%d" % i, block(c, c, self.D, norm_fn)) for i in range(block_reps)))
        if len(self.nPlanes) > 1:
            c1 = self.nPlanes[1]
            self.conv = nn.Sequential(*_norm_relu_conv(norm_fn, c, c1, kernel_size=2, stride=2, dimension=self.D))
            self.u = UBlock(self.nPlanes[1:], norm_fn, block_reps, block)
            self.deconv = nn.Sequential(*_norm_relu_conv(norm_fn, c1, c, conv_cls=ME.MinkowskiConvolutionTranspose,
                                                         kernel_size=2, stride=2, dimension=self.D))
            self.blocks_tail = nn.Sequential(OrderedDict(
                ("block%d" % i, block(c * (2 - i), c, self.D, norm_fn)) for i in range(block_reps)))

    def forward(self, x):
        skip = self.blocks(x)
        if len(self.nPlanes) == 1:
            return skip
        deeper = self.deconv(self.u(self.conv(skip)))
        return self.blocks_tail(ME.cat(skip, deeper))
