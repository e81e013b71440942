"""CIFAR-10 / CIFAR-100 from the python-pickle distribution (reference: datasets/cifar.py:9-84)."""
import os
import pickle

import numpy as np

from .common import InMemoryDatasetGenerator


def _load(path, label_key):
    with open(path, 'rb') as f:
        dump = pickle.load(f, encoding='bytes')
    get = lambda k: dump[k.encode()] if k.encode() in dump else dump[k]
    return get('data').astype(np.float32), list(get(label_key))


class CifarGenerator(InMemoryDatasetGenerator):
    def __init__(self, root_dir, classes=None, reenumerate=False, cifar10=False, **kwargs):
        self.root_dir = root_dir
        if cifar10:
            parts = [_load(os.path.join(root_dir, 'data_batch_%d' % i), 'labels') for i in range(1, 6)]
            X_train = np.concatenate([p[0] for p in parts])
            y_train = [l for p in parts for l in p[1]]
            X_test, y_test = _load(os.path.join(root_dir, 'test_batch'), 'labels')
        else:
            X_train, y_train = _load(os.path.join(root_dir, 'train'), 'fine_labels')
            X_test, y_test = _load(os.path.join(root_dir, 'test'), 'fine_labels')
        if classes is not None:
            keep = set(classes)
            tr = np.array([l in keep for l in y_train])
            te = np.array([l in keep for l in y_test])
            X_train, y_train = X_train[tr], [l for l in y_train if l in keep]
            X_test, y_test = X_test[te], [l for l in y_test if l in keep]
            self.classes = classes
            if reenumerate:
                self.class_indices = {c: i for i, c in enumerate(classes)}
                y_train = [self.class_indices[l] for l in y_train]
                y_test = [self.class_indices[l] for l in y_test]
        else:
            self.classes = np.arange(max(y_train) + 1)
            self.class_indices = {c: c for c in self.classes}
        to_img = lambda a: a.reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1)
        super().__init__(to_img(X_train), to_img(X_test), y_train, y_test, **kwargs)
