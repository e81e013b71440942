"""Host-side mirror of the reference's ``utils.py`` API on PyTorch-ROCm + the sehip HIP kernels.

Same names, argument meaning and conventions as the reference so its scripts and README snippets
keep working (reference: utils.py:26-465):

* ``ARCHITECTURES`` / ``LR_SCHEDULES``                                   utils.py:26-30
* losses / metrics with the Keras signature ``f(y_true, y_pred) -> [B]``   utils.py:34-122
* ``l2norm``                                                              utils.py:125-127
* ``build_network(num_outputs, architecture, ...)``                      utils.py:130-276
* ``get_custom_objects``                                                  utils.py:279-285
* ``get_lr_schedule`` / ``add_lr_schedule_arguments``                    utils.py:288-418
* ``TemplateModelCheckpoint``                                             utils.py:422-465

What changed underneath: tensors are torch device tensors; ``y_true`` may be EITHER the gathered
class embeddings ``[B, D]`` (the reference's convention, learn_image_embeddings.py:48-50) or the
raw integer labels ``[B]`` -- with labels the fused HIP kernels are used (gather + l2norm + loss in
one launch, MFMA contraction for the metric), which is what the trainer does.  Callbacks are small
framework-free objects driven by ``engine.Trainer`` instead of Keras callbacks.
"""
import math
import os
import warnings

import numpy as np
import torch

import sehip
from models import cifar_resnet, resnet50

ARCHITECTURES = ['simple', 'resnet-32', 'resnet-110', 'resnet-110-fc', 'resnet-110-wfc', 'wrn-28-10',
                 'densenet-100-12', 'densenet-100-24', 'densenet-bc-190-40', 'pyramidnet-272-200', 'pyramidnet-110-270',
                 'resnet-50', 'resnet-101', 'resnet-152', 'rn18', 'rn34', 'rn50', 'rn101', 'rn152', 'rn200', 'nasnet-a']

LR_SCHEDULES = ['SGD', 'SGDR', 'CLR', 'ResNet-Schedule']

# architectures with a PyTorch-ROCm body in this build (BASELINE.json configs); the other names are
# kept so CLI `choices` match the reference but raise NotImplementedError in build_network
IMPLEMENTED_ARCHITECTURES = ['resnet-32', 'resnet-110', 'resnet-110-fc', 'resnet-110-wfc', 'resnet-50']


def _is_labels(y_true):
    return y_true.dim() == 1 and not y_true.is_floating_point()


# ------------------------------------------------------------------------------------------------
# losses (Keras signature: y_true, y_pred -> per-sample tensor)
# ------------------------------------------------------------------------------------------------

def squared_distance(y_true, y_pred):
    """Squared Euclidean distance between corresponding rows (utils.py:34-36)."""
    return torch.sum(torch.square(y_pred.float() - y_true.float()), dim=-1)


def mean_distance(y_true, y_pred):
    """Euclidean distance between corresponding rows (utils.py:39-41)."""
    return torch.sqrt(squared_distance(y_true, y_pred))


def inv_correlation(y_true, y_pred):
    """1 - <y_true, y_pred> per row (utils.py:44-46); ``y_pred`` is expected L2-normalised."""
    return 1. - torch.sum(y_true.float() * y_pred.float(), dim=-1)


class CosineEmbeddingLoss(object):
    """The fused form of ``Lambda(l2norm)`` + ``transform_inputs`` + ``inv_correlation``:
    ``loss(labels[B] int64, raw_features[B, D]) -> [B]`` in ONE HIP launch forward and one backward
    (learn_image_embeddings.py:48-50,127-128; utils.py:44-46,125-127)."""

    name = 'inv_correlation'

    def __init__(self, embedding):
        self.embedding = embedding

    def __call__(self, y_true, y_pred):
        if not _is_labels(y_true):
            raise ValueError('CosineEmbeddingLoss expects integer class labels as y_true')
        loss_i, xhat = sehip.cosine_embedding_loss(y_pred, y_true, self.embedding, reduction='none', return_normalized=True)
        self.last_normalized = xhat      # what the reference's l2norm layer would have emitted; metrics use it
        return loss_i


class SquaredDistanceLoss(object):
    """The fused form of ``transform_inputs`` + ``squared_distance`` (the `--loss mse` training loss, learn_image_embeddings.py:48-50,
    160-163; utils.py:34-36): ``loss(labels[B] int64, features[B, D]) -> [B]``, one HIP launch forward and one backward.  With
    gathered embeddings as ``y_true`` (the reference's convention) it is ``squared_distance`` itself."""

    name = 'squared_distance'

    def __init__(self, embedding):
        self.embedding = embedding

    def __call__(self, y_true, y_pred):
        if not _is_labels(y_true):
            return squared_distance(y_true, y_pred)
        if y_pred.is_cuda and y_pred.dim() == 2 and y_pred.dtype in (torch.float32, torch.bfloat16):
            return sehip.squared_distance_loss(y_pred, y_true, self.embedding)
        return squared_distance(self.embedding[y_true], y_pred)


def l2norm(x):
    """L2-normalises a tensor along the last axis (utils.py:125-127), HIP kernel + autograd."""
    return sehip.l2norm(x)


def top_k_acc(k):
    """Top-k categorical accuracy metric on probability outputs (utils.py:49-54)."""

    def acc(y_true, y_pred):
        target = y_true if _is_labels(y_true) else y_true.argmax(dim=-1)
        topk = y_pred.topk(k, dim=-1).indices
        return (topk == target[:, None]).any(dim=-1).float()

    acc.name = 'acc{}'.format(k)
    return acc


def nn_accuracy(embedding, dot_prod_sim=False, k=1):
    """Accuracy of assigning samples to the class with the nearest embedding (utils.py:57-100).

    ``embedding``: [C, D] array/tensor.  Returns ``metric(y_true, y_pred) -> [B]`` of 0/1 with a
    ``.name`` like the reference.  ``y_true`` = integer labels (fast path) or gathered embeddings
    (reference convention; the labels are then recovered by exact row match)."""
    emb_dev = {}

    def _emb(device):
        if device not in emb_dev:
            e = embedding if torch.is_tensor(embedding) else torch.from_numpy(np.asarray(embedding, dtype=np.float32))
            emb_dev[device] = e.to(device=device, dtype=torch.float32).contiguous()
        return emb_dev[device]

    def _labels(y_true, emb):
        if _is_labels(y_true):
            return y_true.long()
        # Gathered rows -> class indices by EXACT row match (learn_image_embeddings.py:48-50 feeds embedding[y]).  Identical
        # rows of the embedding (e.g. the zero rows of nab.sim*) are interchangeable for the metric -- same true score, same
        # class scores -- so the first match is taken; a row that is no class embedding at all cannot use the label kernel.
        yt = y_true.float()
        d = torch.cdist(yt, emb, compute_mode='donot_use_mm_for_euclid_dist')
        idx = d.argmin(dim=-1)
        if not bool((emb[idx] == yt).all()):
            return None
        return idx

    def _generic(y_true, y_pred, dot, emb):
        # y_true is not a row of the embedding: evaluate utils.py:73-95 as written, on the kernel's score matrix
        yp, yt = y_pred.float().contiguous(), y_true.float()
        dummy = torch.zeros((yp.shape[0],), dtype=torch.int64, device=yp.device)
        _, scores = sehip.nn_accuracy(yp, dummy, emb, dot_prod_sim=dot, k=1, want_scores=True)
        if dot:
            true = torch.sum(yp * yt, dim=-1)
            top = scores.topk(min(max(k, 1), scores.shape[1]), dim=-1).values
        else:
            true = torch.sum(torch.square(yp - yt), dim=-1)
            top = -(-scores).topk(min(max(k, 1), scores.shape[1]), dim=-1).values
        return (torch.abs(top - true[:, None]) < 1e-6).any(dim=-1).float()

    def _run(y_true, y_pred, dot):
        emb = _emb(y_pred.device)
        labels = _labels(y_true, emb)
        if labels is None:
            return _generic(y_true, y_pred, dot, emb)
        return sehip.nn_accuracy(y_pred.float().contiguous(), labels.contiguous(), emb, dot_prod_sim=dot, k=k)

    def nn_accuracy(y_true, y_pred):
        return _run(y_true, y_pred, False)

    def max_sim_acc(y_true, y_pred):
        return _run(y_true, y_pred, True)

    metric = max_sim_acc if dot_prod_sim else nn_accuracy
    metric.name = metric.__name__ if k <= 1 else '{}{}'.format(metric.__name__, k)
    return metric


def devise_ranking_loss(embedding, margin=0.1):
    """DeViSE ranking loss (utils.py:103-122): fused HIP kernel (MFMA contraction with the hinge, its row sums and the
    active mask in the epilogue; the mask feeds the backward contraction).  ``y_true``: gathered embeddings [B, D] (the
    reference's convention) or integer labels [B]."""
    emb_dev = {}

    def _loss(y_true, y_pred):
        if y_pred.device not in emb_dev:
            e = embedding if torch.is_tensor(embedding) else torch.from_numpy(np.asarray(embedding, dtype=np.float32))
            emb_dev[y_pred.device] = e.to(device=y_pred.device, dtype=torch.float32).contiguous()
        return sehip.devise_ranking_loss(y_pred, y_true, emb_dev[y_pred.device], margin)

    return _loss


# ------------------------------------------------------------------------------------------------
# model factory
# ------------------------------------------------------------------------------------------------

def build_network(num_outputs, architecture, classification=False, no_softmax=False, input_channels=None, name=None):
    """Constructs a CNN (same arguments as the reference, utils.py:130-151).  Returns an
    ``nn.Module`` in channels_last format whose head is called ``embedding`` (or ``prob``)."""
    activation = 'relu'
    if architecture.lower().endswith('-selu'):
        activation, architecture = 'selu', architecture[:-5]
    top = 'softmax' if classification and (not no_softmax) else None

    if architecture in ('resnet-32', 'resnet-110'):
        # NB (utils.py:160-172): without -fc these emit the 64-d pooled features unless classifying
        return cifar_resnet.SmallResNet(5 if architecture == 'resnet-32' else 18, filters=[16, 32, 64],
                                        activation=activation, include_top=classification,
                                        top_activation=None if no_softmax else 'softmax', classes=num_outputs,
                                        name=name, input_channels=input_channels)
    if architecture == 'resnet-110-fc':
        return cifar_resnet.SmallResNet(18, filters=[16, 32, 64], activation=activation, include_top=True,
                                        top_activation=top, classes=num_outputs, name=name, input_channels=input_channels)
    if architecture == 'resnet-110-wfc':
        return cifar_resnet.SmallResNet(18, filters=[32, 64, 128], activation=activation, include_top=True,
                                        top_activation=top, classes=num_outputs, name=name, input_channels=input_channels)
    if architecture == 'resnet-50':
        return resnet50.ResNet50(num_outputs, classification=classification, no_softmax=no_softmax,
                                 input_channels=input_channels or 3, name=name)
    if architecture in ARCHITECTURES:
        raise NotImplementedError('architecture "{}" is accepted for CLI compatibility but has no PyTorch-ROCm body in '
                                  'this build (implemented: {})'.format(architecture, ', '.join(IMPLEMENTED_ARCHITECTURES)))
    raise ValueError('Unknown network architecture: {}'.format(architecture))


def get_custom_objects(architecture):
    """Keras needed this to deserialise ``ChannelPadding`` (utils.py:279-285); kept for callers."""
    if architecture in ('resnet-32', 'resnet-110', 'resnet-110-fc', 'resnet-110-wfc', 'pyramidnet-272-200', 'pyramidnet-110-270'):
        return {'ChannelPadding': cifar_resnet.ChannelPadding}
    return {}


# ------------------------------------------------------------------------------------------------
# learning-rate schedules: callbacks with on_train_begin / on_batch_end / on_epoch_end hooks that
# read and write ``trainer.lr`` (the role keras.callbacks played, utils.py:288-399)
# ------------------------------------------------------------------------------------------------

class Callback(object):
    def on_train_begin(self, trainer): pass
    def on_epoch_begin(self, trainer, epoch): pass
    def on_batch_end(self, trainer, batch, logs): pass
    def on_epoch_end(self, trainer, epoch, logs): pass


class SGDR(Callback):
    """Cosine annealing with warm restarts, per epoch (reference: sgdr_callback.py:6-87):
    lr(i) = min + (max - min)/2 * (1 + cos(pi * (i+1) / T)), T multiplied by ``mul_epochs`` after
    each cycle; counters restart from 0 on every ``fit`` like the reference's."""

    def __init__(self, min_lr=0.0, max_lr=0.05, base_epochs=10, mul_epochs=2):
        self.min_lr, self.max_lr, self.base_epochs, self.mul_epochs = min_lr, max_lr, base_epochs, mul_epochs
        self.cycles = 0.
        self.cycle_iterations = 0.
        self.trn_iterations = 0.

    def _lr(self):
        period = self.base_epochs * (self.mul_epochs ** self.cycles)
        return self.min_lr + 0.5 * (self.max_lr - self.min_lr) * (1 + np.cos(np.pi * (self.cycle_iterations + 1) / period))

    def on_train_begin(self, trainer):
        trainer.lr = self.max_lr if self.cycle_iterations == 0 else self._lr()

    def on_epoch_end(self, trainer, epoch, logs):
        logs['lr'] = trainer.lr
        self.trn_iterations += 1
        self.cycle_iterations += 1
        if self.cycle_iterations >= self.base_epochs * (self.mul_epochs ** self.cycles):
            self.cycles += 1
            self.cycle_iterations = 0
            trainer.lr = self.max_lr
        else:
            trainer.lr = self._lr()


class CyclicLR(Callback):
    """Triangular cyclical learning rate, per batch (reference: clr_callback.py, mode 'triangular')."""

    def __init__(self, base_lr=0.001, max_lr=0.006, step_size=2000., mode='triangular'):
        if mode != 'triangular':
            raise NotImplementedError('only the triangular policy is used by the reference CLI')
        self.base_lr, self.max_lr, self.step_size = base_lr, max_lr, float(step_size)
        self.clr_iterations = 0.

    def clr(self):
        cycle = np.floor(1 + self.clr_iterations / (2 * self.step_size))
        x = np.abs(self.clr_iterations / self.step_size - 2 * cycle + 1)
        return self.base_lr + (self.max_lr - self.base_lr) * np.maximum(0, (1 - x))

    def on_train_begin(self, trainer):
        trainer.lr = self.base_lr if self.clr_iterations == 0 else self.clr()

    def on_batch_end(self, trainer, batch, logs):
        self.clr_iterations += 1
        trainer.lr = self.clr()


class LearningRateScheduler(Callback):
    """lr = schedule(epoch[, current_lr]) at the start of each epoch (Keras semantics)."""

    def __init__(self, schedule):
        self.schedule = schedule

    def on_epoch_begin(self, trainer, epoch):
        try:
            trainer.lr = float(self.schedule(epoch, trainer.lr))
        except TypeError:
            trainer.lr = float(self.schedule(epoch))


class ReduceLROnPlateau(Callback):
    """Multiply the lr by ``factor`` after ``patience`` epochs without ``monitor`` improving by
    more than ``epsilon`` (Keras defaults: factor 0.1, mode min)."""

    def __init__(self, monitor='val_loss', factor=0.1, patience=10, epsilon=1e-4, min_lr=0., verbose=False):
        self.monitor, self.factor, self.patience, self.epsilon, self.min_lr, self.verbose = monitor, factor, patience, epsilon, min_lr, verbose
        self.best, self.wait = float('inf'), 0

    def on_epoch_end(self, trainer, epoch, logs):
        logs['lr'] = trainer.lr
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if cur < self.best - self.epsilon:
            self.best, self.wait = cur, 0
            return
        self.wait += 1
        if self.wait >= self.patience and trainer.lr > self.min_lr:
            trainer.lr = max(trainer.lr * self.factor, self.min_lr)
            self.wait = 0
            if self.verbose:
                print('\nEpoch %05d: reducing learning rate to %s.' % (epoch + 1, trainer.lr))


def get_lr_schedule(schedule, num_samples, batch_size, schedule_args={}):
    """Returns ``(callbacks, suggested number of epochs)`` for 'SGD' | 'SGDR' | 'CLR' |
    'ResNet-Schedule' with the reference's defaults (utils.py:288-399)."""
    kind = schedule.lower()
    if kind == 'sgd':
        spec = schedule_args.get('sgd_schedule')
        if spec:
            points = []
            for item in spec.split(','):
                parts = item.split(':')
                points.append((int(parts[0]) - 1, float(parts[1]) if len(parts) > 1 else None))
            points.sort()

            def piecewise(epoch, cur_lr):
                chosen = None
                for start, lr in points:
                    if start <= epoch:
                        chosen = lr
                return cur_lr if chosen is None else chosen

            return [LearningRateScheduler(piecewise)], points[-1][0] + 1
        schedule_args.setdefault('sgd_patience', 10)
        schedule_args.setdefault('sgd_min_lr', 1e-4)
        return [ReduceLROnPlateau('val_loss', patience=schedule_args['sgd_patience'], epsilon=1e-4,
                                  min_lr=schedule_args['sgd_min_lr'], verbose=True)], 200
    if kind == 'sgdr':
        schedule_args.setdefault('sgdr_base_len', 12)
        schedule_args.setdefault('sgdr_mul', 2)
        schedule_args.setdefault('sgdr_max_lr', 0.1)
        base, mul = schedule_args['sgdr_base_len'], schedule_args['sgdr_mul']
        return [SGDR(1e-6, schedule_args['sgdr_max_lr'], base, mul)], sum(base * (mul ** i) for i in range(5))
    if kind == 'clr':
        schedule_args.setdefault('clr_step_len', 12)
        schedule_args.setdefault('clr_min_lr', 1e-5)
        schedule_args.setdefault('clr_max_lr', 0.1)
        steps = schedule_args['clr_step_len'] * (num_samples // batch_size)
        return [CyclicLR(schedule_args['clr_min_lr'], schedule_args['clr_max_lr'], steps, mode='triangular')], schedule_args['clr_step_len'] * 20
    if kind == 'resnet-schedule':
        def he_schedule(epoch):
            return 0.001 if epoch >= 120 else 0.01 if epoch >= 80 else 0.1 if epoch >= 1 else 0.01
        return [LearningRateScheduler(he_schedule)], 164
    raise ValueError('Unknown learning rate schedule: {}'.format(schedule))


def add_lr_schedule_arguments(parser):
    """The shared --sgd_* / --sgdr_* / --clr_* flag groups (utils.py:402-418)."""
    g = parser.add_argument_group('Parameters for --lr_schedule=SGD')
    g.add_argument('--sgd_patience', type=int, default=None, help='Epochs without improvement before the LR is reduced.')
    g.add_argument('--sgd_lr', type=float, default=0.1, help='Initial learning rate.')
    g.add_argument('--sgd_min_lr', type=float, default=None, help='Lower bound of the learning rate.')
    g.add_argument('--sgd_schedule', type=str, default=None,
                   help='Comma-separated `epoch:lr` pairs (1-based epochs); a trailing bare number sets the total epoch count.')
    g = parser.add_argument_group('Parameters for --lr_schedule=SGDR')
    g.add_argument('--sgdr_base_len', type=int, default=None, help='Epochs in the first cycle.')
    g.add_argument('--sgdr_mul', type=int, default=None, help='Cycle-length multiplier.')
    g.add_argument('--sgdr_max_lr', type=float, default=None, help='Learning rate at the start of each cycle.')
    g = parser.add_argument_group('Parameters for --lr_schedule=CLR')
    g.add_argument('--clr_step_len', type=int, default=None, help='Epochs per half-cycle.')
    g.add_argument('--clr_min_lr', type=float, default=None, help='Lowest learning rate.')
    g.add_argument('--clr_max_lr', type=float, default=None, help='Highest learning rate.')


class ModelCheckpoint(Callback):
    """Saves ``{'model': state_dict, 'epoch': n}`` after every epoch, or only when ``monitor``
    improves (``save_best_only``); ``filepath`` may contain ``{epoch}`` / log-name placeholders."""

    def __init__(self, filepath, monitor='val_loss', save_best_only=False, mode='auto', verbose=0):
        self.filepath, self.monitor, self.save_best_only, self.verbose = filepath, monitor, save_best_only, verbose
        self.maximize = (mode == 'max') or (mode == 'auto' and ('acc' in monitor or monitor.startswith('fmeasure')))
        self.best = -float('inf') if self.maximize else float('inf')

    def _model(self, trainer):
        return trainer.model

    def on_epoch_end(self, trainer, epoch, logs):
        if not trainer.is_main_process:
            return
        path = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None:
                warnings.warn('Can save best model only with %s available, skipping.' % self.monitor, RuntimeWarning)
                return
            better = cur > self.best if self.maximize else cur < self.best
            if not better:
                return
            self.best = cur
        torch.save({'model': self._model(trainer).state_dict(), 'epoch': epoch + 1}, path)
        if self.verbose:
            print('Epoch %05d: saving model to %s' % (epoch + 1, path))


class TemplateModelCheckpoint(ModelCheckpoint):
    """Checkpoints a given template model instead of the data-parallel wrapper (utils.py:422-465).
    With one process per GPU every rank holds the same un-wrapped model, so this only pins which
    module is saved."""

    def __init__(self, tpl_model, filepath, *args, **kwargs):
        super().__init__(filepath, *args, **kwargs)
        self.tpl_model = tpl_model

    def _model(self, trainer):
        return self.tpl_model
