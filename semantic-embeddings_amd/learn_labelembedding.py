"""Host-side mirror of the loss functions and the model head of the reference's ``learn_labelembedding.py`` (lines 17-61),
executed by the MI355X kernels of libsehip.so.  Only the loss is on the hot path named by
BASELINE.json; ``labelembed_model`` / ``transform_inputs`` are mirrored so that the loss is usable the
way the reference uses it; the training CLI around them (argument parsing, callbacks) is out of scope.
"""
import torch


def cross_entropy(logit, prob):
    """``K.sum(prob * log_softmax(logit), axis=1)`` (learn_labelembedding.py:17-18).  Plain torch: it is only
    used inside ``labelembed_loss`` in the reference, where the fused kernel computes it."""
    return torch.sum(prob * torch.log_softmax(logit, dim=1), dim=1)


def labelembed_loss(out1, out2, tar, targets, tau=2., alpha=0.9, beta=0.5, num_classes=100):
    """Same signature and meaning as the reference (learn_labelembedding.py:21-37); returns the per-sample loss
    ``[B]`` (the reference's Lambda layer appends ``[:, None]``, learn_labelembedding.py:54).  ``num_classes`` is
    accepted for signature compatibility; the class count is the logits' last dimension."""
    import sehip  # raises SehipError when the HIP library / a ROCm device is missing -- no CPU fallback
    targets = targets.reshape(-1).to(torch.int64).contiguous()
    return sehip.labelembed_loss(out1.float(), out2.float(), tar.float(), targets, tau=tau, alpha=alpha, beta=beta)


class LabelEmbedModel(torch.nn.Module):
    """``labelembed_model(base_model, num_classes, **kwargs)`` (learn_labelembedding.py:40-56) as a module: the base network's
    embedding goes through ReLU -> BatchNorm (``embedding_bn``) into two heads -- ``prob`` (out1) and ``out2``, the latter
    behind a stop-gradient -- and a learnable ``[C, C]`` label-embedding table initialised to the identity
    (``labelembeddings``) supplies ``tar`` for the sample's label.  ``forward(x, labels)`` returns the reference model's three
    outputs ``(embedding, out1, loss[:, None])`` with the loss computed by the fused HIP kernel."""

    def __init__(self, base_model, num_classes, embed_dim=None, tau=2., alpha=0.9, beta=0.5):
        super().__init__()
        from models.cifar_resnet import KERAS_BN_EPS, KERAS_BN_MOMENTUM, keras_dense
        self.base_model = base_model
        if embed_dim is None:
            head = getattr(base_model, 'head', None)
            embed_dim = head.out_features if head is not None else base_model.num_features
        self.embedding_bn = torch.nn.BatchNorm1d(embed_dim, eps=KERAS_BN_EPS, momentum=KERAS_BN_MOMENTUM)
        self.prob = keras_dense(embed_dim, num_classes)
        self.out2 = keras_dense(embed_dim, num_classes)
        self.labelembeddings = torch.nn.Embedding(num_classes, num_classes)
        with torch.no_grad():
            self.labelembeddings.weight.copy_(torch.eye(num_classes))
        self.num_classes, self.kwargs = num_classes, dict(tau=tau, alpha=alpha, beta=beta)

    def forward(self, x, labels):
        embedding = self.base_model(x)
        out = self.embedding_bn(torch.relu(embedding.float()))
        out1 = self.prob(out)
        out2 = self.out2(out.detach())                                  # Lambda(K.stop_gradient)
        labels = labels.reshape(-1).to(torch.int64)
        tar = self.labelembeddings(labels)
        loss = labelembed_loss(out1, out2, tar, labels, num_classes=self.num_classes, **self.kwargs)
        return embedding, out1, loss[:, None]


def labelembed_model(base_model, num_classes, **kwargs):
    """Same call as the reference's factory (learn_labelembedding.py:40)."""
    return LabelEmbedModel(base_model, num_classes, **kwargs)


def transform_inputs(X, y, num_classes):
    """learn_labelembedding.py:59-61: inputs ``[X, y]``, targets for the two trained outputs (a dummy for the loss output, the
    labels -- instead of their one-hot encoding -- for ``prob``)."""
    return [X, y], {'labelembed_loss': torch.zeros((len(X), 1), device=X.device), 'prob': y}
