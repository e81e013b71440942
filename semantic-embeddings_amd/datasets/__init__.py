"""Data generators with the attribute/method interface of the reference's ``datasets`` package
(reference: datasets/__init__.py:21-166, datasets/common.py:210-331,673-844):
``num_classes, num_train, num_test, num_channels, labels_train, labels_test, classes`` and
``train_sequence / test_sequence / flow_train / flow_test``.

Batches are produced ON THE DEVICE (channels_last float tensors), so images/s is not bound by a
Python loader: the synthetic generators draw N(0,1) images with fixed seeds (BASELINE.json's
configs are all measured on synthetic batches), the CIFAR generator keeps the whole pickle-decoded
dataset in HBM and augments there.  File-based datasets of the reference (ILSVRC, NAB, CUB, ...)
are out of scope of this build (real-data input path is a "next" row).
"""
from .common import DeviceBatchSequence, InMemoryDatasetGenerator, SyntheticGenerator  # noqa: F401
from .cifar import CifarGenerator  # noqa: F401

SYNTHETIC_PRESETS = {
    # name: (num_classes, height/width, channels, num_train, num_test)
    'synthetic-cifar100': (100, 32, 3, 50000, 10000),
    'synthetic-cub': (200, 224, 3, 5994, 5794),
    'synthetic-ilsvrc': (1000, 224, 3, 1281167, 50000),
}


def get_data_generator(dataset, data_root, classes=None):
    """Shortcut for creating a data generator with default settings (datasets/__init__.py:21).

    Supported names: 'cifar-10', 'cifar-100', 'cifar-100-a', 'cifar-100-b' (python pickles under
    ``data_root``) and 'synthetic-cifar100' / 'synthetic-cub' / 'synthetic-ilsvrc' or the generic
    'synthetic:<classes>x<size>x<train>x<test>' (``data_root`` ignored)."""
    name = dataset.lower()
    if name in SYNTHETIC_PRESETS:
        c, hw, ch, ntr, nte = SYNTHETIC_PRESETS[name]
        return SyntheticGenerator(c if classes is None else len(classes), hw, ch, ntr, nte, classes=classes)
    if name.startswith('synthetic:'):
        c, hw, ntr, nte = (int(v) for v in name.split(':', 1)[1].split('x'))
        return SyntheticGenerator(c if classes is None else len(classes), hw, 3, ntr, nte, classes=classes)
    if name == 'cifar-10':
        return CifarGenerator(data_root, classes, reenumerate=True, cifar10=True)
    if name == 'cifar-100':
        return CifarGenerator(data_root, classes, reenumerate=True)
    if name.startswith('cifar-100-a'):
        return CifarGenerator(data_root, list(range(50)), reenumerate=name.endswith('-consec'))
    if name.startswith('cifar-100-b'):
        return CifarGenerator(data_root, list(range(50, 100)), reenumerate=name.endswith('-consec'))
    raise NotImplementedError('dataset "{}": file-based datasets of the reference are outside the scope of this build; '
                              'use cifar-10/100 or a synthetic-* generator'.format(dataset))
