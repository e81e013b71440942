"""CIFAR ResNet backbone (He et al., section 4.2) as a PyTorch-ROCm module.

Architecture and layer naming follow the reference's Keras model
(reference: models/cifar_resnet.py:28-257 -- ``ChannelPadding``, ``simple_block``, ``unit``,
``SmallResNet``): 3x3 conv stem, three units of ``n`` two-conv blocks with 16/32/64 (or wider)
channels, stride-2 average-pool + zero channel padding on the identity shortcut when the width
changes (or a strided 1x1 convolution with ``conv_shortcut``), global average pooling and an
optional dense layer named ``embedding`` (no activation) or ``prob`` (softmax).

Keras semantics mirrored here: every convolution has a bias and Glorot-uniform weights, batch
normalisation uses eps = 1e-3 and a running-average momentum of 0.99, and the L2 kernel
regulariser (2e-4) is exposed through ``regularized_parameters()`` so the trainer can add it to
the loss before gradient clipping.  The module runs in channels_last memory format; MIOpen picks
the convolution kernels -- the backbone is plumbing, not the product (the custom HIP kernels are
the loss/metric/retrieval ones).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

KERAS_BN_EPS = 1e-3
KERAS_BN_MOMENTUM = 0.01   # torch convention: new = (1 - m) * old + m * batch; Keras momentum 0.99


def _glorot_uniform_(weight):
    """Keras' default kernel initialiser: U(-l, l), l = sqrt(6 / (fan_in + fan_out))."""
    if weight.dim() == 4:
        rf = weight.shape[2] * weight.shape[3]
        fan_in, fan_out = weight.shape[1] * rf, weight.shape[0] * rf
    else:
        fan_out, fan_in = weight.shape
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    with torch.no_grad():
        weight.uniform_(-limit, limit)


def keras_conv(cin, cout, k, stride=1):
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=True)
    _glorot_uniform_(conv.weight)
    nn.init.zeros_(conv.bias)
    return conv


def keras_bn(c):
    bn = nn.BatchNorm2d(c, eps=KERAS_BN_EPS, momentum=KERAS_BN_MOMENTUM)
    # Keras' BatchNormalization has no batch counter and a fixed momentum never reads torch's: dropping the buffer saves one
    # `num_batches_tracked += 1` launch per layer and step (109 launches = 0.5 ms of a 15 ms resnet-110 step)
    bn.register_buffer('num_batches_tracked', None)
    return bn


def keras_dense(cin, cout):
    fc = nn.Linear(cin, cout)
    _glorot_uniform_(fc.weight)
    nn.init.zeros_(fc.bias)
    return fc


class ChannelPadding(nn.Module):
    """Zero-padding on the channel axis: ``padding`` = int or (before, after)."""

    def __init__(self, padding=1):
        super().__init__()
        self.padding = (padding, padding) if isinstance(padding, int) else tuple(padding)

    def forward(self, x):  # NCHW logical layout
        return F.pad(x, (0, 0, 0, 0, self.padding[0], self.padding[1]))


_ACT = {"relu": nn.ReLU, "selu": nn.SELU}


class SimpleBlock(nn.Module):
    """conv-bn-act-conv-bn (+ shortcut) - act; layer names res<prefix>x/y/z, bn<prefix>x/y/z."""

    def __init__(self, cin, cout, stride=1, activation="relu", conv_shortcut=False, bn=True):
        super().__init__()
        self.resx = keras_conv(cin, cout, 3, stride)
        self.bnx = keras_bn(cout) if bn else nn.Identity()
        self.resy = keras_conv(cout, cout, 3, 1)
        self.bny = keras_bn(cout) if bn else nn.Identity()
        self.act = _ACT[activation]()
        self.resz = None
        self.pool = None
        self.pad = None
        if cin != cout and conv_shortcut:
            self.resz = keras_conv(cin, cout, 1, stride)
            self.resz.padding = (0, 0)
            self.bnz = keras_bn(cout) if bn else nn.Identity()
        else:
            if stride > 1:
                self.pool = nn.AvgPool2d(stride, stride)
            if cin < cout:
                extra = cout - cin
                self.pad = ChannelPadding((extra // 2, extra - extra // 2))

    def forward(self, x):
        y = self.act(self.bnx(self.resx(x)))
        y = self.bny(self.resy(y))
        s = x
        if self.resz is not None:
            s = self.bnz(self.resz(s))
        else:
            if self.pool is not None:
                s = self.pool(s)
            if self.pad is not None:
                s = self.pad(s)
        return self.act(y + s)


class SmallResNet(nn.Module):
    """``SmallResNet(n, filters, include_top, ..., classes, name)`` as in the reference
    (models/cifar_resnet.py:149-155); depth = 2 * len(filters) * n + 2."""

    def __init__(self, n=9, filters=(16, 32, 64), include_top=True, weights=None, input_tensor=None, input_shape=None,
                 pooling="avg", regularizer=2e-4, activation="relu", top_activation="softmax", conv_shortcut=False,
                 bn=True, classes=100, name=None, input_channels=None):
        super().__init__()
        if weights is not None:
            raise NotImplementedError("loading Keras .h5 weights is not supported (no h5py); use torch state_dicts")
        cin = input_channels or (input_shape[-1] if input_shape else 3)
        self.name = name or "cifar-resnet{}".format(2 * len(filters) * n)
        self.regularizer = float(regularizer or 0.0)
        self.include_top = include_top
        self.pooling = pooling
        self.top_activation = top_activation
        self.conv0 = keras_conv(cin, filters[0], 3)
        self.bn0 = keras_bn(filters[0]) if bn else nn.Identity()
        self.act = _ACT[activation]()
        blocks = []
        prev = filters[0]
        for u, width in enumerate(filters):
            for b in range(n):
                stride = 2 if (u > 0 and b == 0) else 1
                blocks.append(SimpleBlock(prev, width, stride, activation, conv_shortcut, bn))
                prev = width
        self.blocks = nn.Sequential(*blocks)
        self.num_features = prev
        self.avg_pool = nn.Identity()      # named tap on the pooled features (the reference's GlobalAveragePooling2D 'avg_pool': --cls_base avg_pool)
        if include_top:
            head = keras_dense(prev, classes)
            # the head is called 'embedding' when it has no activation and 'prob' otherwise
            if top_activation is None:
                self.embedding = head
            else:
                self.prob = head
        self.to(memory_format=torch.channels_last)

    @property
    def head(self):
        return getattr(self, "embedding", None) or getattr(self, "prob", None)

    def features(self, x):
        x = self.act(self.bn0(self.conv0(x)))
        x = self.blocks(x)
        if self.pooling == "avg":
            x = self.avg_pool(x.mean(dim=(2, 3)))
        elif self.pooling == "max":
            x = x.amax(dim=(2, 3))
        return x

    def forward(self, x):
        x = self.features(x)
        if self.include_top:
            x = self.head(x)
            if self.top_activation == "softmax":
                x = torch.softmax(x.float(), dim=-1)
        return x

    def regularized_parameters(self):
        """Kernels carrying the Keras L2 regulariser (conv + dense kernels; not biases, not BN)."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                yield m.weight
