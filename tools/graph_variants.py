"""Tuning aid: does the HIP-graph replay of the training step stay finite under a given backbone mode, and how fast is it?
usage: python tools/graph_variants.py <layout>_<dtype>[_bench][_det]   e.g. nhwc_bf16_bench"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-embeddings_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main(variant, B=int(os.environ.get("B", "128"))):
    import utils
    from datasets import SyntheticGenerator
    from engine import Trainer
    from train_bench import load_embedding
    dev = torch.device("cuda", 0)
    arch = os.environ.get("ARCH", "resnet-110-fc")
    classes, size = (200, 224) if arch == "resnet-50" else (100, 32)
    parts = variant.split("_")
    torch.backends.cudnn.benchmark = "bench" in parts
    torch.backends.cudnn.deterministic = "det" in parts
    emb_dev = torch.from_numpy(load_embedding(classes).astype(np.float32)).to(dev)
    torch.manual_seed(0)
    model = utils.build_network(classes, arch, input_channels=3).to(dev)
    nchw = parts[0] == "nchw"
    if nchw:
        model = model.to(memory_format=torch.contiguous_format)
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": None}[parts[1]]
    loss = utils.CosineEmbeddingLoss(emb_dev)
    metric = utils.nn_accuracy(emb_dev, dot_prod_sim=True)
    l2_of = {id(p): model.regularizer for p in model.regularized_parameters()} if getattr(model, "regularizer", 0) else {}
    tr = Trainer(model, {"l2norm": (loss, 1.0)}, {"l2norm": [metric]}, lr=0.1, momentum=0.9, clipnorm=10.0, l2_of=l2_of, autocast_dtype=dt)
    gen = SyntheticGenerator(classes, size, 3, B * 16, B)
    seq = gen.train_sequence(B, shuffle=False, rank=0, world_size=1)
    batches = [seq[i] for i in range(8)]
    if nchw:
        batches = [(x.contiguous(), y) for x, y in batches]
    ok = tr.enable_graphs(*batches[0])
    finite = []
    logs = {}
    for i in range(40):
        tr.train_step(*batches[i % 8], logs)
        torch.cuda.synchronize()
        finite.append(bool(torch.isfinite(tr.flat.flat_g).all()) and bool(torch.isfinite(tr.flat.flat_p).all()))
    first_bad = finite.index(False) if False in finite else -1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        tr.train_step(*batches[i % 8], logs)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print("%-22s capture %s  first non-finite replay %d  %.2f ms/step %.0f img/s  mean loss %.4f" % (
        variant, ok, first_bad, ms, B / ms * 1e3, float(logs["loss"]) / max(float(logs.get("_n", 1)), 1.0)), flush=True)   # per-sample sums / sample count


if __name__ == "__main__":
    main(sys.argv[1])
