"""oracle/keras_stub.py -- TEST INFRASTRUCTURE ONLY (this container only).

A NumPy-backed stand-in for ``keras`` / ``keras.backend`` that is just large enough to import the
reference's ``utils.py``, ``learn_labelembedding.py``, ``clr_callback.py`` and ``sgdr_callback.py``
UNMODIFIED and to evaluate their loss / metric / schedule functions eagerly on NumPy arrays.  With it
the golden loss vectors under tests/golden/loss_ref_*.npz are produced by the reference's own source
lines (utils.py:34-127, learn_labelembedding.py:17-37); what stays a restatement is the handful of
third-party primitives those lines call.  Every such primitive is listed here with the documented
formula it restates (Keras 2.2.x ``tensorflow_backend.py`` / TensorFlow 1.x, the versions the
reference pins in README.md:315-316):

* ``K.sum/square/sqrt/abs/min/max/any/less/equal/cast/dot/relu/argmax/flatten/shape/one_hot/constant``
  -- elementwise / reduction ops with NumPy's semantics in ``floatx`` precision (``K.constant`` casts
  to ``floatx`` like Keras does);
* ``K.softmax``        -> ``exp(x - max) / sum(exp(x - max))`` (tf.nn.softmax);
* ``K.stop_gradient``  -> identity (forward value);
* ``K.sparse_categorical_crossentropy(target, output)`` -> Keras 2.2: clip ``output`` to
  ``[1e-7, 1 - 1e-7]``, take the log, hand the result to ``sparse_softmax_cross_entropy_with_logits``
  (i.e. ``-log_softmax(log(clip(p)))[target]`` -- the renormalisation is part of the definition);
* ``K.tf.nn.l2_normalize(x, axis, epsilon=1e-12)`` -> ``x * rsqrt(maximum(sum(x^2, axis), epsilon))``;
* ``K.tf.nn.top_k(x, k, sorted=False)``  -> the k largest values per row (returned in descending order;
  the reference only feeds them to an order-free ``any``);
* ``K.tf.nn.log_softmax`` -> ``x - max - log(sum(exp(x - max)))``.

``floatx`` is 'float32' (what the reference computes in) or 'float64' (used to produce tight
reference values for the oracle's float64 restatement).
"""
import collections
import types

import numpy as np


class Variable(object):
    """Mutable scalar (``model.optimizer.lr``) for ``K.set_value`` / ``K.get_value`` -- lets the reference's
    clr_callback.py / sgdr_callback.py be driven epoch by epoch without a model."""

    def __init__(self, value=0.0):
        self.value = float(value)


def make_keras(floatx="float32"):
    """Returns {module name: module} to be placed in ``sys.modules``."""
    fx = np.dtype(floatx)
    keras = types.ModuleType("keras")
    K = types.ModuleType("keras.backend")

    def _f(x):
        a = np.asarray(x)
        return a.astype(fx) if a.dtype.kind == "f" and a.dtype != fx else a

    K.floatx = lambda: fx.name
    K.epsilon = lambda: 1e-7
    K.image_data_format = lambda: "channels_last"
    K.constant = lambda value, dtype=None, shape=None, name=None: np.asarray(value, dtype=dtype or fx)
    K.sum = lambda x, axis=None, keepdims=False: np.sum(_f(x), axis=axis, keepdims=keepdims)
    K.mean = lambda x, axis=None, keepdims=False: np.mean(_f(x), axis=axis, keepdims=keepdims)
    K.max = lambda x, axis=None, keepdims=False: np.max(_f(x), axis=axis, keepdims=keepdims)
    K.min = lambda x, axis=None, keepdims=False: np.min(_f(x), axis=axis, keepdims=keepdims)
    K.any = lambda x, axis=None, keepdims=False: np.any(x, axis=axis, keepdims=keepdims)
    K.square = lambda x: np.square(_f(x))
    K.sqrt = lambda x: np.sqrt(np.clip(_f(x), 0, np.inf))          # Keras clips to [0, inf) before tf.sqrt
    K.abs = lambda x: np.abs(_f(x))
    K.less = lambda x, y: np.less(x, y)
    K.equal = lambda x, y: np.equal(x, y)
    K.cast = lambda x, dtype: np.asarray(x).astype(dtype)
    K.dot = lambda x, y: np.dot(_f(x), _f(y))
    K.relu = lambda x, alpha=0.0, max_value=None: np.maximum(_f(x), 0)
    K.argmax = lambda x, axis=-1: np.argmax(x, axis=axis).astype(np.int64)
    K.flatten = lambda x: np.reshape(x, [-1])
    K.shape = lambda x: np.asarray(np.shape(x))
    K.one_hot = lambda indices, num_classes: np.eye(num_classes, dtype=fx)[np.asarray(indices)]
    K.stop_gradient = lambda x: x
    K.get_value = lambda x: x.value if isinstance(x, Variable) else x
    K.set_value = lambda x, v: setattr(x, "value", float(v))
    K.variable = lambda value, dtype=None, name=None: Variable(value)

    def softmax(x, axis=-1):
        x = _f(x)
        e = np.exp(x - np.max(x, axis=axis, keepdims=True))
        return e / np.sum(e, axis=axis, keepdims=True)

    def log_softmax(logits, axis=-1):
        x = _f(logits)
        z = x - np.max(x, axis=axis, keepdims=True)
        return z - np.log(np.sum(np.exp(z), axis=axis, keepdims=True))

    def sparse_categorical_crossentropy(target, output, from_logits=False):
        output = _f(output)
        if not from_logits:
            eps = fx.type(1e-7)
            output = np.log(np.clip(output, eps, fx.type(1) - eps))
        t = np.asarray(target).reshape(-1).astype(np.int64)
        ls = log_softmax(output.reshape(-1, output.shape[-1]))
        return -ls[np.arange(len(t)), t]

    def l2_normalize(x, axis=None, epsilon=1e-12, name=None, dim=None):
        x = _f(x)
        axis = dim if axis is None else axis
        ss = np.sum(np.square(x), axis=axis, keepdims=True)
        return x * (fx.type(1) / np.sqrt(np.maximum(ss, fx.type(epsilon))))

    TopK = collections.namedtuple("TopKV2", ["values", "indices"])

    def top_k(x, k=1, sorted=True, name=None):
        x = _f(x)
        idx = np.argsort(-x, axis=-1, kind="stable")[..., :k]
        return TopK(np.take_along_axis(x, idx, axis=-1), idx.astype(np.int32))

    K.softmax = softmax
    K.sparse_categorical_crossentropy = sparse_categorical_crossentropy
    tf = types.ModuleType("tensorflow_stub")
    tf.nn = types.SimpleNamespace(l2_normalize=l2_normalize, top_k=top_k, log_softmax=log_softmax, softmax=softmax)
    K.tf = tf

    # ---- the non-numeric attributes the reference's modules touch at import time ----
    class Callback(object):
        def __init__(self):
            self.model = None
            self.params = {}

        def set_model(self, model):
            self.model = model

    class LearningRateScheduler(Callback):
        def __init__(self, schedule, verbose=0):
            super(LearningRateScheduler, self).__init__()
            self.schedule, self.verbose = schedule, verbose

    class ReduceLROnPlateau(Callback):
        def __init__(self, monitor="val_loss", **kw):
            super(ReduceLROnPlateau, self).__init__()
            self.monitor, self.kw = monitor, kw

    class ModelCheckpoint(Callback):
        def __init__(self, filepath, *a, **kw):
            super(ModelCheckpoint, self).__init__()
            self.filepath = filepath

    callbacks = types.ModuleType("keras.callbacks")
    callbacks.Callback = Callback
    callbacks.LearningRateScheduler = LearningRateScheduler
    callbacks.ReduceLROnPlateau = ReduceLROnPlateau
    callbacks.ModelCheckpoint = ModelCheckpoint
    mods = {"keras": keras, "keras.backend": K, "keras.callbacks": callbacks}
    for sub in ("layers", "models", "utils", "metrics", "applications", "optimizers", "regularizers", "preprocessing"):
        m = types.ModuleType("keras." + sub)
        mods["keras." + sub] = m
        setattr(keras, sub, m)
    keras.backend = K
    keras.callbacks = callbacks

    # ---- keras.utils.Sequence + keras.preprocessing.image.ImageDataGenerator, as far as the reference's datasets/common.py uses
    #      them for in-memory datasets (TinyDatasetGenerator, datasets/common.py:635-844) [third party: Keras 2.2 = keras_preprocessing
    #      1.0.x, restated from its documented behaviour: featurewise statistics over (samples, rows, columns); standardize =
    #      (x - mean) / (std + 1e-6), in place; random_transform = random height / width shift (uniform in +-range * size, bilinear
    #      scipy.ndimage.affine_transform, mode 'nearest') then random horizontal flip; np.random draws in that order] ----
    class Sequence(object):
        def on_epoch_end(self):
            pass

    mods["keras.utils"].Sequence = Sequence
    mods["keras.utils"].to_categorical = lambda y, num_classes=None: np.eye(num_classes or int(np.max(y)) + 1, dtype=fx)[np.asarray(y, dtype=np.int64)]

    class ImageDataGenerator(object):
        def __init__(self, featurewise_center=False, featurewise_std_normalization=False, horizontal_flip=False,
                     width_shift_range=0.0, height_shift_range=0.0, **unsupported):
            if unsupported:
                raise NotImplementedError("ImageDataGenerator stand-in: %s" % sorted(unsupported))
            self.featurewise_center, self.featurewise_std_normalization = featurewise_center, featurewise_std_normalization
            self.horizontal_flip, self.width_shift_range, self.height_shift_range = horizontal_flip, width_shift_range, height_shift_range
            self.mean = self.std = None
            self.last_transform = None      # (tx rows, ty columns, flip) of the last random_transform (fixture generation reads it)

        def fit(self, x, augment=False, rounds=1, seed=None):
            x = np.array(x, dtype=fx, copy=True)
            if self.featurewise_center:
                self.mean = np.mean(x, axis=(0, 1, 2)).reshape(1, 1, x.shape[3])
                x -= self.mean
            if self.featurewise_std_normalization:
                self.std = np.std(x, axis=(0, 1, 2)).reshape(1, 1, x.shape[3])
                x /= (self.std + 1e-6)

        def standardize(self, x):
            if self.featurewise_center:
                x -= self.mean
            if self.featurewise_std_normalization:
                x /= (self.std + 1e-6)
            return x

        def random_transform(self, x, seed=None):
            import scipy.ndimage
            tx = ty = 0.0
            if self.height_shift_range:
                tx = np.random.uniform(-self.height_shift_range, self.height_shift_range) * x.shape[0]
            if self.width_shift_range:
                ty = np.random.uniform(-self.width_shift_range, self.width_shift_range) * x.shape[1]
            flip = bool((np.random.random() < 0.5) * self.horizontal_flip)
            self.last_transform = (tx, ty, flip)
            if tx != 0 or ty != 0:
                # translation about the image centre == plain translation: out[r, c] = in[r + tx, c + ty]
                x = np.stack([scipy.ndimage.affine_transform(x[..., ch], np.eye(2), offset=(tx, ty), order=1, mode="nearest", cval=0.0)
                              for ch in range(x.shape[2])], axis=-1)
            if flip:
                x = x[:, ::-1, :]
            return x

    image = types.ModuleType("keras.preprocessing.image")
    image.ImageDataGenerator = ImageDataGenerator
    image.load_img = image.img_to_array = image.list_pictures = None      # the file-based generators import these names; never called here
    mods["keras.preprocessing.image"] = image
    mods["keras.preprocessing"].image = image
    return mods
