import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-embeddings_amd")); sys.path.insert(0, ROOT)
import utils
from engine import Trainer
from datasets import SyntheticGenerator
from train_bench import load_embedding
dev = torch.device("cuda")
emb = torch.from_numpy(load_embedding(100).astype(np.float32)).to(dev)
def make():
    torch.manual_seed(0)
    m = utils.build_network(100, "resnet-110-fc", input_channels=3).to(dev)
    loss = utils.CosineEmbeddingLoss(emb); metric = utils.nn_accuracy(emb, dot_prod_sim=True)
    l2_of = {id(p): m.regularizer for p in m.regularized_parameters()} if getattr(m, "regularizer", 0) else {}
    return Trainer(m, {"l2norm": (loss, 1.0)}, {"l2norm": [metric]}, lr=0.1, momentum=0.9, clipnorm=10.0, l2_of=l2_of)
gen = SyntheticGenerator(100, 32, 3, 128 * 64, 128)
seq = gen.train_sequence(128, shuffle=False, rank=0, world_size=1)
batches = [seq[i] for i in range(8)]
for mode in ():
    t = make()
    if mode == "graph":
        print("capture:", t.enable_graphs(*batches[0]))
    for i in range(12):
        logs = {}
        l = t.train_step(*batches[i % 8], logs)
        torch.cuda.synchronize()
        print(mode, i, float(l), {k: float(v) for k, v in logs.items()}, float(t.flat.flat_p.abs().max()), float(t.flat.flat_g.abs().max()))


import torch.nn.functional as F
class TorchLoss(object):
    name = "inv_correlation"
    def __init__(self, e): self.e = e
    def __call__(self, y, x):
        xh = F.normalize(x.float(), dim=-1, eps=1e-6)
        self.last_normalized = xh.detach()
        return 1.0 - (xh * self.e[y]).sum(-1)
def make2(kind, autocast=True, metric_on=True):
    torch.manual_seed(0)
    m = utils.build_network(100, "resnet-110-fc", input_channels=3).to(dev)
    loss = utils.CosineEmbeddingLoss(emb) if kind == "hip" else TorchLoss(emb)
    metric = utils.nn_accuracy(emb, dot_prod_sim=True)
    l2_of = {id(p): m.regularizer for p in m.regularized_parameters()} if getattr(m, "regularizer", 0) else {}
    return Trainer(m, {"l2norm": (loss, 1.0)}, {"l2norm": [metric]} if metric_on else {}, lr=0.1, momentum=0.9, clipnorm=10.0, l2_of=l2_of,
                   autocast_dtype=torch.bfloat16 if autocast else None)

class HipLossNoShare(utils.CosineEmbeddingLoss):
    def __call__(self, y, x):
        li = super().__call__(y, x)
        self.last_normalized = F.normalize(x.detach().float(), dim=-1)    # metric input from torch, not from the kernel
        return li
class TorchMetric(object):
    name = "max_sim_acc"
    def __init__(self, e): self.e = e
    def __call__(self, y, xh):
        s = xh.float() @ self.e.t()
        return (s.argmax(-1) == y).float()
def run(tag, loss, metrics):
    torch.manual_seed(0)
    m = utils.build_network(100, "resnet-110-fc", input_channels=3).to(dev)
    l2_of = {id(p): m.regularizer for p in m.regularized_parameters()} if getattr(m, "regularizer", 0) else {}
    t = Trainer(m, {"l2norm": (loss, 1.0)}, {"l2norm": metrics}, lr=0.1, momentum=0.9, clipnorm=10.0, l2_of=l2_of)
    ok = t.enable_graphs(*batches[0])
    ga, gb = t._graph
    res = []
    for i in range(6):
        X, y = batches[i % 8]
        t._sX.copy_(X); t._sy[0].copy_(y); t._lr_t.fill_(0.1)
        ga.replay(); torch.cuda.synchronize()
        res.append(bool(torch.isfinite(t.flat.flat_g).all()))
        gb.replay(); torch.cuda.synchronize()
    print(tag, "capture", ok, "grads finite per replay:", res, flush=True)

hipm = utils.nn_accuracy(emb, dot_prod_sim=True)
for rep in range(3):
    run("T torch loss + torch metric", TorchLoss(emb), [TorchMetric(emb)])
    run("N torch loss, no metric", TorchLoss(emb), [])
    run("A hip loss + hip metric", utils.CosineEmbeddingLoss(emb), [hipm])
    run("D torch loss + hip metric", TorchLoss(emb), [hipm])
