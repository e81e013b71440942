"""Drop-in for the reference's ``learn_image_embeddings.py``: learns to map images onto class
embeddings (cosine loss on hierarchy-based unit-sphere embeddings by default), same command line
(reference: learn_image_embeddings.py:57-96 + utils.py:402-418), on MI355X.

    # single GPU
    python learn_image_embeddings.py --dataset synthetic-cifar100 --data_root . \
        --embedding embeddings/cifar100.unitsphere.pickle --architecture resnet-110-fc --batch_size 128
    # data parallel, one process per GPU over RCCL (instead of keras.utils.multi_gpu_model)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 learn_image_embeddings.py ... --gpus 8

Differences from the reference a user can observe: ``--gpus N`` expects to be launched with N
processes (torchrun); ``--batch_size`` stays the GLOBAL batch and is split across the ranks like
``multi_gpu_model`` split it across towers; models / snapshots are torch ``state_dict`` files, not
Keras ``.h5``; ``--read_workers`` / ``--queue_size`` are accepted and ignored (batches are composed
on the device); ``--log_dir`` writes a JSON-lines log instead of TensorBoard events.
"""
import argparse
import json
import os
import pickle

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

import utils
from datasets import get_data_generator
from engine import Trainer, backbone_mode
from models.cifar_resnet import keras_bn, keras_dense


class ClsModel(nn.Module):
    """Embedding model + classifier head (ReLU -> BN -> Dense softmax named ``prob``), two outputs
    (reference: cls_model, learn_image_embeddings.py:16-45).  The reference appends the classifier to the model that
    already ENDS in the ``l2norm`` Lambda layer (``--loss inv_corr``) or the ``softmax`` activation (``softmax_corr``)
    (learn_image_embeddings.py:127-133), so its base is the normalised / soft-maxed output; that tensor is also the first
    output, which loss and metrics consume.  The head emits logits; the categorical cross-entropy is applied on them."""

    def __init__(self, embed_model, num_classes, cls_base=None, head=None, width=None):
        super().__init__()
        if head not in (None, 'l2norm', 'softmax'):
            raise ValueError('head must be None, "l2norm" or "softmax"')
        self.embed_model = embed_model
        self.head = head
        self._tap = None
        self.cls_base = None
        # cls_base 'l2norm' / 'softmax' names the LAST layer of the reference's embedding model (learn_image_embeddings.py:127-133):
        # its output is the first output itself, i.e. the default base -- nothing to tap.
        if cls_base is not None and str(cls_base) not in ('l2norm', 'softmax'):
            # reference: `embed_model.layers[int(cls_base)].output` / `embed_model.get_layer(cls_base).output`
            # (learn_image_embeddings.py:34-40).  Names are module names of the embedding model ('avg_pool' = the pooled backbone
            # features, 'embedding' = the dense layer in front of l2norm, dotted names for anything deeper); an integer indexes its
            # leaf modules in definition order (Keras numbers its own layer list: indices are not portable between the two).
            named = [(n, m) for n, m in embed_model.named_modules() if n]
            no_such = '--cls_base {!r}: no such layer; the embedding model has {}'.format(cls_base, ', '.join(n for n, _ in named))
            try:
                index = int(cls_base)
            except ValueError:
                index = None
            if index is not None:
                leaves = [(n, m) for n, m in named if not list(m.children())]
                if not -len(leaves) <= index < len(leaves):
                    raise ValueError(no_such + ' ({} leaf layers)'.format(len(leaves)))
                name, tap = leaves[index]
            else:
                found = dict(named)
                if str(cls_base) not in found:
                    raise ValueError(no_such)
                name, tap = str(cls_base), found[str(cls_base)]
            # the tap's width must follow from the layer itself -- the caller's `width` describes the embedding OUTPUT, not an inner layer
            if isinstance(tap, nn.Linear):
                width = tap.out_features
            elif isinstance(tap, (nn.BatchNorm1d, nn.BatchNorm2d)):
                width = tap.num_features
            elif name == 'avg_pool' or name.endswith('.avg_pool'):
                width = embed_model.num_features
            else:
                raise ValueError('--cls_base {!r} ({}): cannot tell the width of that layer\'s output (dense, batch-norm and avg_pool '
                                 'layers are supported)'.format(cls_base, type(tap).__name__))
            self.cls_base = name
            tap.register_forward_hook(self._remember)
        if width is None:       # width of what the embedding model emits (resnet-32 / -110 without -fc: the pooled features)
            width = embed_model.head.out_features if getattr(embed_model, 'head', None) is not None else None
        if width is None:
            raise ValueError('ClsModel needs the output width of the embedding model')
        self.bn = nn.BatchNorm1d(width, eps=1e-3, momentum=0.01)
        self.prob = keras_dense(width, num_classes)
        self.cls_l2 = 5e-4

    def _remember(self, module, inputs, output):
        self._tap = output

    def forward(self, x):
        emb = self.embed_model(x)
        if self.head == 'l2norm':
            base = utils.l2norm(emb)                    # HIP kernel with autograd (utils.py:125-127)
        elif self.head == 'softmax':
            base = torch.softmax(emb.float(), -1)
        else:
            base = emb.float()
        cls_in = base
        if self.cls_base is not None:                   # the classifier hangs off an inner layer; the first output stays the embedding
            cls_in, self._tap = self._tap, None
            if cls_in is None or cls_in.dim() != 2:
                raise ValueError('--cls_base {!r} does not produce a [batch, features] tensor'.format(self.cls_base))
            cls_in = cls_in.float()
        return base, self.prob(self.bn(torch.relu(cls_in)))


def transform_inputs(X, y, embedding=None, num_classes=None):
    """reference: learn_image_embeddings.py:48-50.  The reference gathers ``embedding[y]`` on the
    host; here the labels travel to the fused kernel, which gathers on the device."""
    return (X, y) if num_classes is None else (X, [y, y])


def categorical_crossentropy(y_true, logits):
    return nn.functional.cross_entropy(logits.float(), y_true, reduction='none')


def accuracy(y_true, out):
    return (out.argmax(dim=-1) == y_true).float()


accuracy.name = 'acc'


def build_parser():
    parser = argparse.ArgumentParser(description='Learns to map images onto class embeddings (MI355X build).',
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    g = parser.add_argument_group('Data parameters')
    g.add_argument('--dataset', type=str, required=True, help='Dataset name (see datasets.get_data_generator).')
    g.add_argument('--data_root', type=str, required=True, help='Dataset root directory.')
    g.add_argument('--embedding', type=str, required=True,
                   help='Pickle written by compute_class_embedding.py ({"embedding", "ind2label", ...}) or "onehot".')
    g = parser.add_argument_group('Training parameters')
    g.add_argument('--architecture', type=str, default='simple', choices=utils.ARCHITECTURES, help='Network architecture.')
    g.add_argument('--loss', type=str, default='inv_corr', choices=['mse', 'inv_corr', 'unnorm_corr', 'softmax_corr'],
                   help='"inv_corr": cosine loss on L2-normalised outputs (fused HIP kernel); "mse": squared distance; '
                        '"unnorm_corr"/"softmax_corr": negated dot product without normalisation / after softmax.')
    g.add_argument('--cls_weight', type=float, default=0.0, help='Weight of an additional softmax classification loss (0 = off).')
    g.add_argument('--cls_base', type=str, default=None, help='Name or index of the layer that the classification layer should be based on. If not specified, the final embedding layer will be used.')
    g.add_argument('--lr_schedule', type=str, default='SGDR', choices=utils.LR_SCHEDULES, help='Learning-rate schedule.')
    g.add_argument('--clipgrad', type=float, default=10.0, help='Global gradient-norm clip.')
    g.add_argument('--max_decay', type=float, default=0.0, help='Learning-rate decay reached at the end of training.')
    g.add_argument('--nesterov', action='store_true', default=False, help='Nesterov momentum.')
    g.add_argument('--epochs', type=int, default=None, help='Number of training epochs.')
    g.add_argument('--batch_size', type=int, default=100, help='Global batch size.')
    g.add_argument('--val_batch_size', type=int, default=None, help='Validation batch size.')
    g.add_argument('--snapshot', type=str, default=None, help='Checkpoint written after every epoch; resumed from if present.')
    g.add_argument('--snapshot_best', type=str, nargs='?', default=None, const='val_loss', help='Only keep the best checkpoint w.r.t. this metric.')
    g.add_argument('--initial_epoch', type=int, default=0, help='First epoch when resuming.')
    g.add_argument('--finetune', type=str, default=None, help='state_dict with pre-trained weights (matched by name, mismatches skipped).')
    g.add_argument('--finetune_init', type=int, default=8, help='Epochs training only the new layers first.')
    g.add_argument('--gpus', type=int, default=1, help='Number of GPUs = number of launched processes.')
    g.add_argument('--read_workers', type=int, default=8, help='Ignored (device-side batches).')
    g.add_argument('--queue_size', type=int, default=100, help='Ignored (device-side batches).')
    g.add_argument('--gpu_merge', action='store_true', default=False, help='Ignored (weights always live on the GPUs).')
    g = parser.add_argument_group('Output parameters')
    g.add_argument('--model_dump', type=str, default=None, help='Where to save the whole model (torch.save of the module).')
    g.add_argument('--weight_dump', type=str, default=None, help='Where to save the state_dict.')
    g.add_argument('--feature_dump', type=str, default=None, help='Where to save test-image embeddings ({"feat": {i: vec}} pickle).')
    g.add_argument('--log_dir', type=str, default=None, help='Directory for a JSON-lines training log.')
    g.add_argument('--no_progress', action='store_true', default=False, help='Only print the final performance.')
    g.add_argument('--top_k_acc', type=int, nargs='+', default=[], help='Also report these top-k accuracies.')
    utils.add_lr_schedule_arguments(parser)
    return parser


class JsonLogger(utils.Callback):
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, 'training_log.jsonl')
        open(self.path, 'w').close()

    def on_epoch_end(self, trainer, epoch, logs):
        if trainer.is_main_process:
            with open(self.path, 'a') as f:
                f.write(json.dumps(dict(logs, epoch=epoch + 1)) + '\n')


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.val_batch_size is None:
        args.val_batch_size = args.batch_size

    # ---- process group: one process per GPU over RCCL
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if not torch.cuda.is_available():
        raise RuntimeError('learn_image_embeddings.py needs a ROCm GPU (no CPU fallback for the HIP loss kernels)')
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print('note: --gpus {} but {} process(es) were launched; using {}'.format(args.gpus, world, world))
    dev = torch.device('cuda', torch.cuda.current_device())

    # ---- class embeddings (learn_image_embeddings.py:105-117)
    if args.embedding == 'onehot':
        embed_labels, embedding = None, None
    else:
        with open(args.embedding, 'rb') as pf:
            dump = pickle.load(pf)
        embed_labels, embedding = dump['ind2label'], dump['embedding']
    data_generator = get_data_generator(args.dataset, args.data_root, classes=embed_labels)
    if embedding is None:
        embedding = np.eye(data_generator.num_classes)
    emb_dev = torch.from_numpy(np.asarray(embedding, dtype=np.float32)).to(dev).contiguous()   # f64 -> f32 like the TF feed

    # ---- model (learn_image_embeddings.py:120-148)
    torch.manual_seed(0)   # identical initial weights on every rank
    embed_model = utils.build_network(embedding.shape[1], args.architecture, input_channels=data_generator.num_channels).to(dev)
    model = embed_model
    if args.cls_weight > 0:
        with torch.no_grad():    # output width of the embedding model (not every architecture ends in a Dense layer)
            was = embed_model.training
            embed_model.eval()
            width = int(embed_model(torch.zeros((1, data_generator.num_channels, 32, 32), device=dev)).shape[-1])
            embed_model.train(was)
        model = ClsModel(embed_model, data_generator.num_classes, args.cls_base,
                         head={'inv_corr': 'l2norm', 'softmax_corr': 'softmax'}.get(args.loss), width=width).to(dev)
    if args.snapshot and os.path.exists(args.snapshot):
        print('Resuming from snapshot {}'.format(args.snapshot))
        model.load_state_dict(torch.load(args.snapshot, map_location=dev)['model'])
    if args.finetune:
        print('Loading pre-trained weights from {}'.format(args.finetune))
        state = torch.load(args.finetune, map_location=dev)
        state = state.get('model', state)
        own = model.state_dict()
        model.load_state_dict({k: v for k, v in state.items() if k in own and own[k].shape == v.shape}, strict=False)

    # ---- loss / metrics (learn_image_embeddings.py:160-180)
    onehot_like = (args.loss == 'softmax_corr') or (args.embedding == 'onehot')
    if args.loss == 'inv_corr' and args.cls_weight > 0:
        # the classifier branch needs the normalised embedding as a tensor of the graph (ClsModel applies the l2norm kernel);
        # the cosine loss is then the plain inv_correlation on it, like the reference (utils.py:44-46)
        loss = lambda y, o: utils.inv_correlation(emb_dev[y], o)
        metrics = [accuracy if onehot_like else utils.nn_accuracy(emb_dev, dot_prod_sim=True)]
        metrics += [utils.top_k_acc(k) if onehot_like else utils.nn_accuracy(emb_dev, dot_prod_sim=True, k=k) for k in args.top_k_acc]
    elif args.loss == 'inv_corr':
        loss = utils.CosineEmbeddingLoss(emb_dev)          # fused l2norm + gather + 1 - <.,.>
        metrics = [accuracy if onehot_like else utils.nn_accuracy(emb_dev, dot_prod_sim=True)]
        metrics += [utils.top_k_acc(k) if onehot_like else utils.nn_accuracy(emb_dev, dot_prod_sim=True, k=k) for k in args.top_k_acc]
    elif args.loss.endswith('_corr'):
        # (with --cls_weight the model already emits the soft-maxed output, see ClsModel)
        post = (lambda o: torch.softmax(o.float(), -1)) if (args.loss == 'softmax_corr' and args.cls_weight <= 0) else (lambda o: o.float())
        loss = lambda y, o: utils.inv_correlation(emb_dev[y], post(o))
        metrics = [accuracy if onehot_like else utils.nn_accuracy(emb_dev, dot_prod_sim=True)]
        metrics += [utils.top_k_acc(k) if onehot_like else utils.nn_accuracy(emb_dev, dot_prod_sim=True, k=k) for k in args.top_k_acc]
    else:
        loss = utils.SquaredDistanceLoss(emb_dev)          # fused gather + squared distance (se_sqdist_loss_fwd / bwd)
        metrics = [accuracy if args.embedding == 'onehot' else utils.nn_accuracy(emb_dev, dot_prod_sim=False)]
        metrics += [utils.top_k_acc(k) if args.embedding == 'onehot' else utils.nn_accuracy(emb_dev, dot_prod_sim=False, k=k) for k in args.top_k_acc]
    embedding_layer_name = {'inv_corr': 'l2norm', 'softmax_corr': 'softmax'}.get(args.loss, 'embedding')
    losses = {embedding_layer_name: (loss, 1.0)}
    all_metrics = {embedding_layer_name: metrics}
    if args.cls_weight > 0:
        losses['prob'] = (categorical_crossentropy, args.cls_weight)
        all_metrics['prob'] = [accuracy] + [utils.top_k_acc(k) for k in args.top_k_acc]

    # Keras kernel regularisers folded into the update: backbone 2e-4, classifier 5e-4
    l2_of = {id(p): embed_model.regularizer for p in embed_model.regularized_parameters()} if getattr(embed_model, 'regularizer', 0) else {}
    if args.cls_weight > 0:
        l2_of[id(model.prob.weight)] = model.cls_l2

    dp = dict(rank=rank, world_size=world)
    kw = {'embedding': embedding, 'num_classes': data_generator.num_classes if args.cls_weight > 0 else None}
    train_seq = lambda: data_generator.train_sequence(args.batch_size, batch_transform=transform_inputs, batch_transform_kwargs=kw, **dp)
    val_seq = lambda: data_generator.test_sequence(args.val_batch_size, batch_transform=transform_inputs, batch_transform_kwargs=kw, **dp)

    mode = backbone_mode(args.architecture)       # (autocast dtype, memory format) of the PyTorch-ROCm backbone
    # ---- optional warm-up of the new layers only (learn_image_embeddings.py:183-207)
    if args.finetune and args.finetune_init > 0:
        print('Pre-training new layers')
        pre = Trainer(model, losses, all_metrics, lr=args.sgd_lr, momentum=0.9, nesterov=args.nesterov, clipnorm=args.clipgrad,
                      autocast_dtype=mode[0], memory_format=mode[1], l2_of=l2_of, trainable=lambda n: ('embedding' in n) or ('prob' in n))
        pre.fit(train_seq(), val_seq(), epochs=args.finetune_init, verbose=not args.no_progress)
        pre.close()            # drop its gradient hooks before the second trainer registers its own
        for p in model.parameters():
            p.requires_grad_(True)
        print('Full model training')

    # ---- main training (learn_image_embeddings.py:209-243)
    sched_args = {k: v for k, v in vars(args).items() if v is not None}
    callbacks, num_epochs = utils.get_lr_schedule(args.lr_schedule, data_generator.num_train, args.batch_size, schedule_args=sched_args)
    epochs = args.epochs if args.epochs else num_epochs
    if args.log_dir:
        callbacks.append(JsonLogger(args.log_dir))
    if args.snapshot:
        ck = {'save_best_only': True, 'monitor': args.snapshot_best} if args.snapshot_best else {}
        callbacks.append(utils.ModelCheckpoint(args.snapshot, **ck) if world <= 1 else utils.TemplateModelCheckpoint(model, args.snapshot, **ck))
    decay = (1.0 / args.max_decay - 1) / ((data_generator.num_train // args.batch_size) * epochs) if args.max_decay > 0 else 0.0
    trainer = Trainer(model, losses, all_metrics, lr=args.sgd_lr, momentum=0.9, nesterov=args.nesterov, clipnorm=args.clipgrad,
                      decay=decay, l2_of=l2_of, autocast_dtype=mode[0], memory_format=mode[1])
    trainer.fit(train_seq(), val_seq(), epochs=epochs, initial_epoch=args.initial_epoch, callbacks=callbacks, verbose=not args.no_progress)

    # ---- final evaluation (learn_image_embeddings.py:246-255)
    final = trainer.evaluate(val_seq())
    if rank == 0:
        print([final[k] for k in sorted(final)], sorted(final))
    if (args.cls_weight > 0) or (args.embedding == 'onehot'):
        pred = trainer.predict(data_generator.test_sequence(args.val_batch_size))
        pred = (pred[1] if args.cls_weight > 0 else pred).argmax(axis=-1)
        y = np.asarray(data_generator.labels_test)
        freq = np.bincount(y)
        if rank == 0 and world == 1:
            print('Average Accuracy: {:.4f}'.format(((pred == y).astype(float) / freq[y]).sum() / len(freq)))

    # ---- dumps (learn_image_embeddings.py:258-275)
    if rank == 0:
        if args.weight_dump:
            torch.save(model.state_dict(), args.weight_dump)
        if args.model_dump:
            torch.save(model, args.model_dump)
        if args.feature_dump:
            feats = trainer.predict(data_generator.test_sequence(max(args.val_batch_size, 256)))
            feats = feats[0] if args.cls_weight > 0 else feats
            if args.cls_weight > 0:
                pass                        # ClsModel's first output already is the l2norm / softmax layer's
            elif args.loss == 'inv_corr':   # the reference's model ends in the l2norm layer
                feats = utils.l2norm(torch.from_numpy(feats).to(dev)).cpu().numpy()
            elif args.loss == 'softmax_corr':
                feats = torch.softmax(torch.from_numpy(feats), -1).numpy()
            with open(args.feature_dump, 'wb') as f:
                pickle.dump({'feat': dict(enumerate(feats))}, f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return final


if __name__ == '__main__':
    main()
