"""ResNet-50 backbone with the layer layout of ``keras.applications.ResNet50`` (the "initial
implementation" the reference insists on, reference: utils.py:228-243): 7x7/2 stem, 3x3/2 max-pool,
bottleneck stages [3, 4, 6, 3] with the stride on the FIRST 1x1 convolution of each stage (ResNet
v1, unlike torchvision's v1.5), every convolution with a bias, BN eps 1e-3 / momentum 0.99,
followed by global average pooling (``avg_pool``) and a dense layer named ``embedding`` (or ``prob``
for classification).  Plumbing around the custom kernels: MIOpen runs the convolutions.
"""
import torch
import torch.nn as nn

from .cifar_resnet import keras_bn, keras_dense, _glorot_uniform_


def _conv(cin, cout, k, stride=1, padding=0):
    c = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=True)
    _glorot_uniform_(c.weight)
    nn.init.zeros_(c.bias)
    return c


class Bottleneck(nn.Module):
    def __init__(self, cin, widths, stride, project):
        super().__init__()
        w1, w2, w3 = widths
        self.a, self.bn_a = _conv(cin, w1, 1, stride), keras_bn(w1)
        self.b, self.bn_b = _conv(w1, w2, 3, 1, 1), keras_bn(w2)
        self.c, self.bn_c = _conv(w2, w3, 1), keras_bn(w3)
        self.proj = _conv(cin, w3, 1, stride) if project else None
        self.bn_proj = keras_bn(w3) if project else None
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        y = self.relu(self.bn_a(self.a(x)))
        y = self.relu(self.bn_b(self.b(y)))
        y = self.bn_c(self.c(y))
        s = self.bn_proj(self.proj(x)) if self.proj is not None else x
        return self.relu(y + s)


class ResNet50(nn.Module):
    STAGES = ((3, (64, 64, 256), 1), (4, (128, 128, 512), 2), (6, (256, 256, 1024), 2), (3, (512, 512, 2048), 2))

    def __init__(self, num_outputs, classification=False, no_softmax=False, input_channels=3, name=None):
        super().__init__()
        self.name = name or "resnet50"
        self.conv1 = _conv(input_channels or 3, 64, 7, 2, 3)
        self.bn_conv1 = keras_bn(64)
        self.relu = nn.ReLU(inplace=True)
        self.pool = nn.MaxPool2d(3, 2, padding=1)
        layers, cin = [], 64
        for count, widths, stride in self.STAGES:
            for i in range(count):
                layers.append(Bottleneck(cin, widths, stride if i == 0 else 1, project=(i == 0)))
                cin = widths[2]
        self.stages = nn.Sequential(*layers)
        self.num_features = cin
        self.avg_pool = nn.Identity()      # named tap on the pooled features (keras.applications.ResNet50's 'avg_pool': --cls_base avg_pool)
        head = keras_dense(cin, num_outputs)
        self.softmax = bool(classification and not no_softmax)
        if classification:
            self.prob = head
        else:
            self.embedding = head
        self.to(memory_format=torch.channels_last)

    @property
    def head(self):
        return getattr(self, "embedding", None) or getattr(self, "prob", None)

    def features(self, x):
        x = self.pool(self.relu(self.bn_conv1(self.conv1(x))))
        return self.avg_pool(self.stages(x).mean(dim=(2, 3)))

    def forward(self, x):
        x = self.head(self.features(x))
        return torch.softmax(x.float(), dim=-1) if self.softmax else x

    def regularized_parameters(self):
        return iter(())     # keras.applications.ResNet50 carries no kernel regulariser
