"""Device-resident batch sequences (the role of datasets/common.py's DataSequence and
TinyDatasetGenerator in the reference, re-designed for one-process-per-GPU training)."""
import numpy as np
import torch


class DeviceBatchSequence(object):
    """Indexable sequence of batches, ``len()`` = batches per epoch, ``seq[i] -> (X, y)`` like a
    ``keras.utils.Sequence`` (datasets/common.py:26-122), but X/y are device tensors.

    ``rank``/``world_size`` shard every global batch across data-parallel processes: rank r takes
    rows ``r::world_size`` of the global batch, so the union over ranks is the reference's batch.
    ``batch_transform(X, y, **kwargs)`` is applied last (e.g. learn_image_embeddings.transform_inputs)."""

    def __init__(self, generator, indices, labels, batch_size=32, shuffle=False, train=False, augment=False,
                 batch_transform=None, batch_transform_kwargs={}, rank=0, world_size=1, seed=0):
        self.generator = generator
        self.indices = np.asarray(indices)
        self.labels = np.asarray(labels)
        self.batch_size, self.shuffle, self.train, self.augment = batch_size, shuffle, train, augment
        self.batch_transform, self.batch_transform_kwargs = batch_transform, batch_transform_kwargs
        self.rank, self.world_size = rank, world_size
        self.rng = np.random.default_rng(seed)      # same seed on every rank -> same permutation
        self.perm = np.arange(len(self.indices))
        self.on_epoch_end()

    def __len__(self):
        return int(np.ceil(len(self.indices) / self.batch_size))

    def on_epoch_end(self):
        if self.shuffle:
            self.rng.shuffle(self.perm)

    def __getitem__(self, idx):
        glob = self.perm[idx * self.batch_size:(idx + 1) * self.batch_size]
        sel = glob[self.rank::self.world_size]
        if len(sel) == 0 and len(glob):     # a short last batch with fewer rows than ranks: no rank may see an empty batch (its mean
            sel = glob[[self.rank % len(glob)]]   # would be NaN and the all-reduce would spread it): re-use one of the rows
        X = self.generator.compose_batch(self.indices[sel], train=self.train, augment=self.augment)
        y = torch.from_numpy(self.labels[sel].astype(np.int64)).to(X.device, non_blocking=True)
        if self.batch_transform is not None:
            return self.batch_transform(X, y, **self.batch_transform_kwargs)
        return X, y

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]
        self.on_epoch_end()


class _GeneratorBase(object):
    """Attribute interface shared by all generators (datasets/common.py:584-631,799-844)."""

    device = None

    def _dev(self):
        if self.device is None:
            self.device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
        return self.device

    @property
    def labels_train(self):
        return self.y_train

    @property
    def labels_test(self):
        return self.y_test

    @property
    def num_classes(self):
        return len(self.classes)

    @property
    def num_train(self):
        return len(self.y_train)

    @property
    def num_test(self):
        return len(self.y_test)

    def train_sequence(self, batch_size=32, shuffle=True, augment=True, batch_transform=None, batch_transform_kwargs={}, **dp):
        return DeviceBatchSequence(self, np.arange(self.num_train), self.y_train, batch_size, shuffle, True, augment,
                                   batch_transform, batch_transform_kwargs, **dp)

    def test_sequence(self, batch_size=32, shuffle=False, augment=False, batch_transform=None, batch_transform_kwargs={}, **dp):
        return DeviceBatchSequence(self, np.arange(self.num_test), self.y_test, batch_size, shuffle, False, augment,
                                   batch_transform, batch_transform_kwargs, **dp)

    def flow_train(self, batch_size=32, include_labels=True, shuffle=True, augment=True):
        for X, y in self.train_sequence(batch_size, shuffle, augment):
            yield (X, y) if include_labels else X

    def flow_test(self, batch_size=32, include_labels=True, shuffle=False, augment=False):
        for X, y in self.test_sequence(batch_size, shuffle, augment):
            yield (X, y) if include_labels else X


class SyntheticGenerator(_GeneratorBase):
    """N(0,1) images drawn on the device from per-batch seeds (reproducible, no host traffic);
    labels uniform over the classes with a fixed seed."""

    def __init__(self, num_classes, size, channels, num_train, num_test, classes=None, dtype=torch.float32):
        self.classes = list(range(num_classes)) if classes is None else list(classes)
        self.size, self.num_channels, self.dtype = size, channels, dtype
        rng = np.random.default_rng(1)
        self.y_train = rng.integers(0, num_classes, size=num_train).tolist()
        self.y_test = rng.integers(0, num_classes, size=num_test).tolist()

    def compose_batch(self, indices, train=True, augment=False):
        dev = self._dev()
        g = torch.Generator(device=dev)
        g.manual_seed(int(indices[0]) * 2 + int(train) if len(indices) else 0)
        x = torch.randn((len(indices), self.num_channels, self.size, self.size), generator=g, device=dev, dtype=self.dtype)
        return x.contiguous(memory_format=torch.channels_last)


class InMemoryDatasetGenerator(_GeneratorBase):
    """Small-image datasets held entirely in HBM (the reference's TinyDatasetGenerator, datasets/common.py:635-844).

    Pre-processing follows Keras' ``ImageDataGenerator(featurewise_center, featurewise_std_normalization).fit(X_train)`` +
    ``standardize`` [third party: keras_preprocessing 1.0.x]: PER-CHANNEL mean and standard deviation of the training set
    (reduced over samples, rows and columns), float32, ``x = (x - mean) / (std + 1e-6)``.  Training batches get a random
    horizontal flip and random width / height shifts drawn uniformly from +-15 % of the image size -- continuous offsets,
    bilinear interpolation (Keras ``order = 1``) with edge replication (``fill_mode = 'nearest'``) -- as tensor ops on the
    device.  (The random numbers come from torch's device generator, not NumPy's: same distribution, different draws.)"""

    def __init__(self, X_train, X_test, y_train, y_test, shift_range=0.15, horizontal_flip=True):
        self.X_train_h, self.X_test_h = X_train, X_test      # NHWC float32 host arrays
        self.y_train, self.y_test = list(y_train), list(y_test)
        self.shift_range, self.horizontal_flip = shift_range, horizontal_flip
        X32 = np.asarray(X_train, dtype=np.float32)
        self.mean = np.mean(X32, axis=(0, 1, 2), keepdims=True)                       # [1, 1, 1, C]
        self.std = np.std(X32 - self.mean, axis=(0, 1, 2), keepdims=True) + np.float32(1e-6)
        self.num_channels = X_train.shape[-1]
        self._dev_data = None

    def _data(self):
        if self._dev_data is None:
            dev = self._dev()
            prep = lambda a: torch.from_numpy(((np.asarray(a, dtype=np.float32) - self.mean) / self.std).transpose(0, 3, 1, 2).copy()).to(dev)
            self._dev_data = (prep(self.X_train_h), prep(self.X_test_h))
        return self._dev_data

    @staticmethod
    def apply_transform(x, row_shift, col_shift, flip):
        """The reference's per-image augmentation with GIVEN parameters, batched on the device: Keras' ``random_transform`` shifts
        first -- ``out[r, c] = in[r + row_shift, c + col_shift]``, bilinear (``order = 1``), edges replicated (``fill_mode =
        'nearest'``) -- and flips horizontally afterwards (datasets/common.py:786-787 -> ImageDataGenerator.random_transform).
        x [B, C, H, W]; row_shift / col_shift [B] float pixels; flip [B] bool."""
        b, _, h, w = x.shape
        rr = torch.arange(h, device=x.device, dtype=torch.float32)[None, :] + row_shift.to(torch.float32)[:, None]   # source row of every output row
        cc = torch.arange(w, device=x.device, dtype=torch.float32)[None, :] + col_shift.to(torch.float32)[:, None]
        gy = (rr / max(h - 1, 1) * 2 - 1)[:, :, None].expand(b, h, w)
        gx = (cc / max(w - 1, 1) * 2 - 1)[:, None, :].expand(b, h, w)
        x = torch.nn.functional.grid_sample(x, torch.stack((gx, gy), dim=-1), mode='bilinear', padding_mode='border', align_corners=True)
        return torch.where(flip[:, None, None, None], x.flip(3), x)

    def draw_transform(self, b, h, w, device):
        """(row_shift, col_shift, flip) like ImageDataGenerator.get_random_transform: shifts uniform in +-shift_range x size, flip
        with probability 1/2 -- from torch's device generator (same distribution as the reference's np.random draws, other numbers)."""
        zero = torch.zeros(b, device=device)
        row = (torch.rand(b, device=device) * 2 - 1) * (self.shift_range * h) if self.shift_range else zero
        col = (torch.rand(b, device=device) * 2 - 1) * (self.shift_range * w) if self.shift_range else zero
        flip = (torch.rand(b, device=device) < 0.5) if self.horizontal_flip else torch.zeros(b, dtype=torch.bool, device=device)
        return row, col, flip

    def compose_batch(self, indices, train=True, augment=False, return_params=False):
        data = self._data()[0 if train else 1]
        idx = torch.from_numpy(np.asarray(indices, dtype=np.int64)).to(data.device)
        x = data.index_select(0, idx)
        params = None
        if augment:
            b, _, h, w = x.shape
            params = self.draw_transform(b, h, w, x.device)
            x = self.apply_transform(x, *params)
        x = x.contiguous(memory_format=torch.channels_last)
        return (x, params) if return_params else x
