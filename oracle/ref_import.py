"""oracle/ref_import.py -- TEST INFRASTRUCTURE ONLY (this container only).

Imports the reference's *own* modules from /root/reference, unmodified, so that golden vectors
can be produced by the real implementation.  ``numexpr``, Keras/TensorFlow and
the reference's ``datasets`` package (which needs Keras) are not importable here, so they are
pre-seeded in ``sys.modules`` with minimal stand-ins:

* ``keras`` / ``keras.backend`` -> oracle/keras_stub.py, a NumPy backend covering exactly the calls of
  utils.py:34-127 and learn_labelembedding.py:17-37 (its header lists each third-party primitive and the
  documented formula it restates);
* ``numexpr.evaluate(expr, local_dict)`` -> ``eval(expr)`` on the NumPy arrays (float32 in,
  float32 out for ``A + B - 2 * C``; numexpr's own promotion rules are third-party and unverified
  -- "parity unpinned" at that one boundary).
* ``datasets.get_data_generator`` -> raises (never called by ``pairwise_retrieval``).

/root/reference does not exist on the GPU box: nothing under tests -m gpu, smoke() or bench.py
may import this module.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("SE_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "evaluate_retrieval.py"))


def _numexpr_stub():
    m = types.ModuleType("numexpr")

    def evaluate(expr, local_dict=None, global_dict=None, **kw):
        return eval(expr, {"__builtins__": {}}, dict(local_dict or {}))

    m.evaluate = evaluate
    return m


def _datasets_stub():
    m = types.ModuleType("datasets")

    def get_data_generator(*a, **k):
        raise RuntimeError("reference datasets package is stubbed (needs Keras)")

    m.get_data_generator = get_data_generator
    return m


def _empty_module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


# names the reference's training-side modules import that must not leak into (or be taken from) the caller's sys.modules
_SHADOWED = ("numexpr", "datasets", "keras", "keras.backend", "keras.callbacks", "keras.layers", "keras.models", "keras.utils",
             "keras.metrics", "keras.applications", "keras.optimizers", "keras.regularizers", "keras.preprocessing",
             "keras_applications", "keras_resnet", "models", "densenet", "clr_callback", "sgdr_callback", "utils", "class_hierarchy")


def import_reference_datasets(floatx="float32"):
    """The reference's own ``datasets`` package (datasets/__init__.py, common.py, cifar.py ...), unmodified, with ``keras``
    resolving to the stand-in (oracle/keras_stub.py: ``keras.utils.Sequence`` and the ``ImageDataGenerator`` subset the in-memory
    generators use).  Returns the package; ``datasets.cifar.CifarGenerator`` etc. are the reference's classes."""
    if not available():
        raise ImportError("reference tree not present at " + REFERENCE_ROOT)
    from oracle import keras_stub
    sys.dont_write_bytecode = True
    saved = {k: v for k, v in sys.modules.items() if k == "datasets" or k.startswith("datasets.") or k == "keras" or k.startswith("keras.")}
    saved_path = list(sys.path)
    try:
        for k in saved:
            sys.modules.pop(k, None)
        sys.modules.update(keras_stub.make_keras(floatx))
        sys.path.insert(0, REFERENCE_ROOT)
        mod = importlib.import_module("datasets")
        subs = {k: v for k, v in sys.modules.items() if k.startswith("datasets.")}
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k == "datasets" or k.startswith("datasets.") or k == "keras" or k.startswith("keras.")]:
            sys.modules.pop(k, None)
        sys.modules.update(saved)
    mod._submodules = subs
    return mod


def import_reference(name, floatx="float32"):
    """Import module ``name`` (e.g. 'evaluate_retrieval', 'class_hierarchy', 'utils', 'learn_labelembedding') from the
    reference tree, unmodified.  ``keras`` resolves to the NumPy stand-in of oracle/keras_stub.py evaluated in ``floatx``
    precision; the reference's ``models`` package and ``densenet`` (network definitions, never called here) are empty stubs;
    ``clr_callback`` / ``sgdr_callback`` / ``utils`` / ``class_hierarchy`` are the reference's own files."""
    if not available():
        raise ImportError("reference tree not present at " + REFERENCE_ROOT)
    from oracle import keras_stub
    sys.dont_write_bytecode = True  # the mount is read-only
    names = set(_SHADOWED) | {name}
    saved = {k: sys.modules.get(k) for k in names}
    saved_path = list(sys.path)
    try:
        for k in names:
            sys.modules.pop(k, None)
        sys.modules["numexpr"] = _numexpr_stub()
        sys.modules["datasets"] = _datasets_stub()
        sys.modules.update(keras_stub.make_keras(floatx))
        sys.modules["models"] = _empty_module("models", cifar_resnet=None, cifar_pyramidnet=None, plainnet=None,
                                              wide_residual_network=None)
        sys.modules["densenet"] = _empty_module("densenet")
        sys.path.insert(0, REFERENCE_ROOT)
        mod = importlib.import_module(name)
    finally:
        sys.path[:] = [p for p in saved_path]
        for k in set(sys.modules) & (names | {n for n in sys.modules if n.startswith("keras.")}):
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
    return mod
