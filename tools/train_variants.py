"""Tuning aid: ms/step of the ResNet-110-fc cosine-embedding training step under backbone layout / precision variants
(the backbone is PyTorch-ROCm plumbing; this only decides which of its modes the trainer defaults to).
usage: python tools/train_variants.py [variant ...]   variants: nhwc_bf16 nchw_bf16 nhwc_fp32 nchw_fp32 nhwc_bf16_bench nhwc_fp16"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-embeddings_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def run(variant, B=int(os.environ.get("B", "128")), steps=20, warm=6):
    import utils
    from datasets import SyntheticGenerator
    from engine import Trainer
    from train_bench import load_embedding
    dev = torch.device("cuda", 0)
    arch = os.environ.get("ARCH", "resnet-110-fc")
    classes, size = (200, 224) if arch == "resnet-50" else (100, 32)
    torch.backends.cudnn.benchmark = variant.endswith("_bench")
    emb_dev = torch.from_numpy(load_embedding(classes).astype(np.float32)).to(dev)
    torch.manual_seed(0)
    model = utils.build_network(classes, arch, input_channels=3).to(dev)
    nchw = variant.startswith("nchw")
    if nchw:
        model = model.to(memory_format=torch.contiguous_format)
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": None}[variant.split("_")[1]]
    loss = utils.CosineEmbeddingLoss(emb_dev)
    metric = utils.nn_accuracy(emb_dev, dot_prod_sim=True)
    l2_of = {id(p): model.regularizer for p in model.regularized_parameters()} if getattr(model, "regularizer", 0) else {}
    tr = Trainer(model, {"l2norm": (loss, 1.0)}, {"l2norm": [metric]}, lr=0.1, momentum=0.9, clipnorm=10.0, l2_of=l2_of, autocast_dtype=dt)
    gen = SyntheticGenerator(classes, size, 3, B * 16, B)
    seq = gen.train_sequence(B, shuffle=False, rank=0, world_size=1)
    batches = [seq[i] for i in range(8)]
    if nchw:
        batches = [(x.contiguous(), y) for x, y in batches]
    logs = {}
    for i in range(warm):
        tr.train_step(*batches[i % 8], logs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.train_step(*batches[i % 8], logs)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # host-only time of a step (launch cost): same loop without the final sync, measured by the wall clock of enqueueing
    t0 = time.perf_counter()
    for i in range(steps):
        tr.train_step(*batches[i % 8], logs)
    host = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    print("%-18s %7.2f ms/step  %8.0f img/s   (host enqueue %.2f ms/step)  loss %.4f" % (variant, ms, B / ms * 1e3, host,
          float(logs["loss"]) / max(float(logs.get("_n", 1)), 1.0)), flush=True)      # logs hold per-sample sums, '_n' the sample count


if __name__ == "__main__":
    vs = sys.argv[1:] or ["nhwc_bf16", "nchw_bf16", "nhwc_fp32", "nchw_fp32", "nhwc_bf16_bench"]
    for v in vs:
        run(v)
