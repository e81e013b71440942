"""Nearest-class-embedding classification on the MI355X distance + ranking kernels.

Mirrors the pieces of the reference's ``evaluate_classification_accuracy.py`` that sit on the hot path's kernels:

* ``nn_classification`` (evaluate_classification_accuracy.py:51-71): ``cdist(feat, centroids, 'sqeuclidean').argsort(-1)`` --
  here ``se_row_sqnorm`` + ``se_pairwise_dist(SE_METRIC_EUCLID)`` + ``se_rank_rows`` on device-resident features (straight
  from ``Trainer.predict(..., to_host=False)`` or any float32 ``[N, D]`` array).  SciPy evaluates ``sum((x - c)^2)`` in
  float64; the kernels use the canonical float32 arithmetic of the retrieval path (``|x|^2 + |c|^2 - 2 x.c``), so rankings
  agree wherever two class distances differ by more than float32 round-off -- ties and near-ties come back in canonical
  (distance, class index) order.
* ``evaluate`` (evaluate_classification_accuracy.py:88-108): accuracy, top-5 accuracy, class-balanced accuracy and
  hierarchical accuracy ``1 - lcs_height`` of a class ranking / prediction vector.

The SVM and softmax-prediction modes of the reference script (``train_and_predict``, ``extract_predictions``) use
scikit-learn / the classifier head and are outside this build's scope.
"""
from collections import OrderedDict

import numpy as np

METRICS = ['Accuracy', 'Top-5 Accuracy', 'Avg. Accuracy', 'Hierarchical Accuracy']


def nn_classification(features, centroids, return_device=False):
    """Class ranking ``[N, C]`` (nearest class embedding first) of every feature row.

    ``features``: float32 ``[N, D]`` ndarray or device tensor; ``centroids``: ``[C, D]`` array / tensor, a dict with an
    ``'embedding'`` item or the path of such a pickle (evaluate_classification_accuracy.py:55-58)."""
    import pickle
    import torch
    import sehip

    if isinstance(centroids, str):
        with open(centroids, 'rb') as f:
            centroids = pickle.load(f)
    if isinstance(centroids, dict):
        centroids = centroids['embedding']
    sehip._lib.require_gpu()
    dev = torch.device('cuda', torch.cuda.current_device())

    def to_dev(a):
        if torch.is_tensor(a):
            return a.detach().to(device=dev, dtype=torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

    f, c = to_dev(features), to_dev(centroids)
    pd = sehip.pairwise_dist(f, c, metric=sehip.METRIC_EUCLID, sqa=sehip.row_sqnorm(f), sqb=sehip.row_sqnorm(c))
    rank = sehip.rank_rows(pd)
    return rank if return_device else rank.cpu().numpy()


def evaluate(y_pred, data_generator, hierarchy=None):
    """Flat, top-5, class-balanced and hierarchical accuracy (evaluate_classification_accuracy.py:88-108).
    ``y_pred``: ``[N]`` predicted class indices or an ``[N, >= 1]`` class ranking."""
    perf = OrderedDict()
    y_true = np.asarray(data_generator.labels_test)
    y_pred = np.asarray(y_pred)
    if y_pred.ndim == 2:
        perf['Top-5 Accuracy'] = float(np.mean(np.any(y_pred[:, :5] == y_true[:, None], axis=-1)))
        y_pred = y_pred[:, 0]
    hit = (y_pred == y_true)
    perf['Accuracy'] = float(np.mean(hit))
    class_freq = np.bincount(y_true)
    perf['Avg. Accuracy'] = float((hit.astype(np.float64) / class_freq[y_true]).sum() / len(class_freq))
    if hierarchy is not None:
        classes = data_generator.classes
        total = 0.0
        for yp, yt in zip(y_pred, y_true):
            total += 1.0 - hierarchy.lcs_height(classes[int(yp)], classes[int(yt)])
        perf['Hierarchical Accuracy'] = total / len(y_true)
    return perf
