"""Why is the replayed bf16 gradient NaN in the Trainer when a plain torch capture of the same network replays fine?
Round 4: the flat parameter buffer now aligns every slice to 256 bytes (engine.FlatState); SE_FLAT_ALIGN=1 restores the packed layout.
    python tools/graph_nan_probe.py            # every configuration with aligned, then with packed slices"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-embeddings_amd")); sys.path.insert(0, ROOT)
import utils
from engine import Trainer
from datasets import SyntheticGenerator
from train_bench import load_embedding
import torch.nn.functional as F
dev = torch.device("cuda")
emb = torch.from_numpy(load_embedding(100).astype(np.float32)).to(dev)

class TorchLoss(object):
    name = "inv_correlation"
    def __init__(self, e): self.e = e
    def __call__(self, y, x):
        xh = F.normalize(x.float(), dim=-1, eps=1e-6)
        self.last_normalized = xh.detach()
        return 1.0 - (xh * self.e[y]).sum(-1)

def make(loss_kind, metric_on, steal, fmt, align):
    os.environ["SE_FLAT_ALIGN"] = str(align)
    torch.manual_seed(0)
    m = utils.build_network(100, "resnet-110-fc", input_channels=3).to(dev)
    loss = utils.CosineEmbeddingLoss(emb) if loss_kind == "hip" else TorchLoss(emb)
    metrics = {"l2norm": [utils.nn_accuracy(emb, dot_prod_sim=True)]} if metric_on else {}
    t = Trainer(m, {"l2norm": (loss, 1.0)}, metrics, lr=0.1, momentum=0.9, clipnorm=10.0, autocast_dtype=torch.bfloat16, memory_format=fmt)
    if not steal:
        t.flat.all_contiguous = False
    return t

gen = SyntheticGenerator(100, 32, 3, 128 * 8, 128)
seq = gen.train_sequence(128, shuffle=False)
X, y = seq[0]
for align in (64, 1):
  for loss_kind, metric_on, steal, fmt in (("hip", True, True, torch.channels_last), ("torch", False, True, torch.channels_last),
                                           ("torch", False, False, torch.channels_last), ("hip", True, False, torch.channels_last),
                                           ("torch", False, True, torch.contiguous_format)):
    for rep in range(2):
        t = make(loss_kind, metric_on, steal, fmt, align)
        ok = t.enable_graphs(X.contiguous(memory_format=fmt), y, validate=3, allow_autocast=True)
        info = getattr(t, "graph_validation", None)
        print("align=%d loss=%s metric=%s steal=%s fmt=%s rep=%d -> graphs=%s %s" % (align, loss_kind, metric_on, steal, "nhwc" if fmt == torch.channels_last else "nchw", rep, ok, info), flush=True)
        if not ok:
            g = t.flat.flat_g
            bad = ~torch.isfinite(g)
            names = [n for n, p in t.model.named_parameters() if p.requires_grad]
            hit = [names[i] for i, (off, n) in enumerate(t.flat.offsets) if bool(bad[off:off + n].any())]
            print("   non-finite in %d of %d parameters, first: %s" % (len(hit), len(names), hit[:4]), flush=True)
