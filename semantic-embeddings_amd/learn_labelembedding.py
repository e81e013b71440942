"""Host-side mirror of the loss functions of the reference's ``learn_labelembedding.py`` (lines 17-37),
executed by the MI355X kernels of libsehip.so.  Only the loss is on the hot path named by
BASELINE.json; the label-embedding training CLI itself is out of scope.
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
