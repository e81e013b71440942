"""Training-step benchmark used by bench.py: ResNet-110-fc (or resnet-50) cosine-embedding
training on synthetic batches -- forward (fp32 for the CIFAR nets, bf16 autocast for resnet-50), fused HIP loss fwd/bwd + MFMA accuracy
metric, backward, RCCL gradient all-reduce, Keras-style SGD update.  Nothing is skipped inside the
timed region (BASELINE.json configs[1] / configs[3])."""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))


def load_embedding(num_classes):
    """CIFAR-100 / CUB unit-sphere class embeddings from the committed fixtures; for other sizes a
    seeded random orthonormal-ish unit-sphere matrix of the same shape (C x C)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "embeddings.npz"))
    if num_classes == 100:
        return g["cifar100_unitsphere"]
    if num_classes == 200:
        return g["cub_balanced_unitsphere"]
    rng = np.random.default_rng(0)
    e = np.linalg.qr(rng.standard_normal((num_classes, num_classes)))[0]
    return e


def bench_train(args, rank, world):
    import utils
    from datasets import SyntheticGenerator
    from engine import Trainer, backbone_mode

    dev = torch.device("cuda", torch.cuda.current_device())
    arch = args.arch
    if arch == "resnet-50":
        classes, size = 200, 224
    else:
        classes, size = 100, 32
    emb = load_embedding(classes)
    emb_dev = torch.from_numpy(emb.astype(np.float32)).to(dev)
    torch.manual_seed(0)
    model = utils.build_network(classes, arch, input_channels=3).to(dev)
    loss = utils.CosineEmbeddingLoss(emb_dev)
    metric = utils.nn_accuracy(emb_dev, dot_prod_sim=True)
    l2_of = {id(p): model.regularizer for p in model.regularized_parameters()} if getattr(model, "regularizer", 0) else {}
    adt, fmt = backbone_mode(arch)                   # fp32 NCHW for the CIFAR ResNets, bf16 channels_last for resnet-50
    trainer = Trainer(model, {"l2norm": (loss, 1.0)}, {"l2norm": [metric]}, lr=0.1, momentum=0.9, clipnorm=10.0, l2_of=l2_of,
                      autocast_dtype=adt, memory_format=fmt)
    B = args.batch                                   # per-GPU batch: weak scaling
    gen = SyntheticGenerator(classes, size, 3, B * 64 * world, B * world)
    seq = gen.train_sequence(B * world, shuffle=False, rank=rank, world_size=world)
    batches = [seq[i] for i in range(8)]             # pre-generated, resident in HBM
    batches = [(x.contiguous(memory_format=fmt), y) for x, y in batches]
    # The product default (Trainer.fit): fp32 steps are replayed as two HIP graphs (fwd+loss+bwd | update) after the
    # capture-time validation against the eager gradient; SE_TRAIN_GRAPHS=0 times eager launches.  bf16-autocast replays
    # fail that validation on torch 2.10 / ROCm 7.0 and stay eager.
    graphs = False
    if adt is None and os.environ.get("SE_TRAIN_GRAPHS", "1") != "0":
        graphs = trainer.enable_graphs(*batches[0])
    logs = {}
    steps, warm = max(args.steps, 20) if args.workload != "train" else args.steps, max(args.warmup, 5)
    for i in range(warm):
        trainer.train_step(*batches[i % len(batches)], logs)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        trainer.train_step(*batches[i % len(batches)], logs)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(torch.as_tensor(logs["loss"]).item()) / (steps + warm)
    if not np.isfinite(loss_val):
        raise FloatingPointError("non-finite mean training loss %r (graphs=%s)" % (loss_val, graphs))
    return {"metric": "train_images_per_sec", "value": B * world * steps / dt, "unit": "images/s", "n_gpus": world,
            "steps": steps, "warmup": warm, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if adt is None else "bf16", "data": "synthetic",
            "config": {"workload": "%s cosine-embedding training step, %dx%dx3, %d classes, per-GPU batch %d" % (arch, size, size, classes, B),
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "backbone": "%s, %s" % ("fp32" if adt is None else "bf16 autocast", "NCHW" if fmt == torch.contiguous_format else "channels_last"),
                       "step": "2 HIP graphs (fwd+loss+bwd | update) + eager RCCL all-reduce" if graphs else "eager launches"},
            "mean_loss": loss_val}


def cpu_baseline_train(args, steps=3):
    """CPU restatement of the reference training step on the host cores (Keras/TF are absent, so this is the
    same PyTorch backbone + the plain-PyTorch cosine loss, fp32, all host threads; "port", not "reference")."""
    import torch.nn.functional as F
    import utils
    from engine import Trainer
    arch = args.arch
    classes, size = (200, 224) if arch == "resnet-50" else (100, 32)
    emb = torch.from_numpy(load_embedding(classes).astype(np.float32))
    torch.manual_seed(0)
    model = utils.build_network(classes, arch, input_channels=3)

    def loss(y, x):                                   # utils.py:125-127 + :44-46 in plain PyTorch
        return 1.0 - (F.normalize(x.float(), dim=-1, eps=1e-6) * emb[y]).sum(-1)
    l2_of = {id(p): model.regularizer for p in model.regularized_parameters()} if getattr(model, "regularizer", 0) else {}
    tr = Trainer(model, {"l2norm": (loss, 1.0)}, {}, lr=0.1, momentum=0.9, clipnorm=10.0, l2_of=l2_of, autocast_dtype=None)
    B = args.batch
    g = torch.Generator().manual_seed(0)
    X = torch.randn(B, 3, size, size, generator=g).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, classes, (B,), generator=g)
    tr.train_step(X, y, {})                           # warm-up (allocations, oneDNN primitive caches)
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(X, y, {})
    dt = time.perf_counter() - t0
    return {"value": B * steps / dt, "unit": "images/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d steps of the same %s step (batch %d, fp32, PyTorch CPU, %d threads), %.1f s" % (steps, arch, B, torch.get_num_threads(), dt)}
