"""oracle/ref_import.py -- TEST INFRASTRUCTURE ONLY (this container only).

Imports the reference's *own* modules from /root/reference, unmodified, so that golden vectors
can be produced by the real implementation.  ``numexpr`` and the
reference's ``datasets`` package (which needs Keras) are not importable here, so they are
pre-seeded in ``sys.modules`` with minimal stand-ins:

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


def import_reference(name):
    """Import module ``name`` (e.g. 'evaluate_retrieval', 'class_hierarchy') from the reference tree."""
    if not available():
        raise ImportError("reference tree not present at " + REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # the mount is read-only
    saved = {k: sys.modules.get(k) for k in ("numexpr", "datasets", name)}
    saved_path = list(sys.path)
    try:
        sys.modules["numexpr"] = _numexpr_stub()
        sys.modules["datasets"] = _datasets_stub()
        sys.modules.pop(name, None)
        sys.path.insert(0, REFERENCE_ROOT)
        mod = importlib.import_module(name)
    finally:
        sys.path[:] = saved_path
        for k in ("numexpr", "datasets"):
            if saved[k] is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = saved[k]
        sys.modules.pop(name, None)
    return mod
