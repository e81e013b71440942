"""PyTorch-ROCm backbones behind ``utils.build_network`` (the reference's models/ factory)."""
from . import cifar_resnet, resnet50  # noqa: F401
