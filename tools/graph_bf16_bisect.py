"""Round 5 bisect of the bf16 + HIP-graph fault: the Trainer's capture of the bf16-autocast ResNet-110 step returns
1-3 non-finite conv-bias gradients; rounds 2-4 suspected the flat parameter buffer.  This walks from a bare capture to the Trainer one
difference at a time (result, profiles/r05_g_bf16_graph_bisect.txt: EVERY variant fails, the bare capture included -- the minimal form is
tools/graph_wrw_bf16_repro.py).  SPLIT_BIAS=1: conv biases applied outside the MIOpen call; NHWC=1: channels_last.

  plain        ordinary parameter tensors; gradients stolen (p.grad = None), packed by torch.cat into a NEW tensor
  views        parameters re-homed as views of one flat fp32 buffer (engine.FlatState), everything else as `plain`
  views+out    ... and the gradients packed by torch.cat(out=flat_g)
  views+upd    ... and the warm-up steps run the whole-buffer SGD update, the state is restored by copy_ before the capture
  trainer      engine.Trainer.enable_graphs(allow_autocast=True), pure-torch loss head, no metric
  trainer/plain-params   the same Trainer over a FlatState that leaves every parameter an ordinary tensor (flat buffers exist, nothing
               aliases them): the bisect the round-4 review asked for

    python tools/graph_bf16_bisect.py [reps]
"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-embeddings_amd")); sys.path.insert(0, ROOT)
import utils
import engine
from engine import Trainer, FlatState
import torch.nn.functional as F

if os.environ.get("SPLIT_BIAS"):
    # the conv bias leaves the MIOpen call: y = conv(x, w) + b  (its gradient is then an at::sum, not MIOpen's backward-bias)
    _orig = torch.nn.Conv2d._conv_forward
    def _split(self, x, w, b):
        y = _orig(self, x, w, None)
        return y if b is None else y + b.view(1, -1, 1, 1).to(y.dtype)
    torch.nn.Conv2d._conv_forward = _split
FMT = torch.channels_last if os.environ.get("NHWC") else torch.contiguous_format
dev = torch.device("cuda")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ARCH, SIZE, CLASSES, B = "resnet-110-fc", 32, 100, 128
torch.manual_seed(1)
E = F.normalize(torch.randn(CLASSES, CLASSES, device=dev), dim=-1)
X = torch.randn(B, 3, SIZE, SIZE, device=dev).contiguous(memory_format=FMT)
Y = torch.randint(0, CLASSES, (B,), device=dev)


class TorchLoss(object):
    name = "inv_correlation"

    def __call__(self, y, x):
        xh = F.normalize(x.float(), dim=-1, eps=1e-6)
        self.last_normalized = xh.detach()
        return 1.0 - (xh * E[y]).sum(-1)


class PlainState(FlatState):
    """FlatState's buffers and bookkeeping, but the parameters stay the ordinary tensors the model was built with."""

    def __init__(self, model, l2_of=None, align=None):
        params = [p for p in model.parameters() if p.requires_grad]
        self.params = params
        self.align = 64
        starts, off = [], 0
        for p in params:
            off = (off + 63) // 64 * 64
            starts.append(off)
            off += p.numel()
        total = (off + 63) // 64 * 64
        z = lambda: torch.zeros(total, dtype=torch.float32, device=params[0].device)
        self.flat_p, self.flat_g, self.flat_v, self.flat_l2 = z(), z(), z(), z()
        self.offsets = [(o, p.numel()) for p, o in zip(params, starts)]
        for p, (o, n) in zip(params, self.offsets):
            self.flat_p[o:o + n] = p.data.reshape(-1)
            p.grad = None
        self.total = total
        self.frozen_l2 = []
        self.has_l2 = False
        self.all_contiguous = True
        self.packed = False
        self._grad_views = [self.flat_g[o:o + n].view(p.shape) for p, (o, n) in zip(params, self.offsets)]


def report(tag, rep, names, offsets, g, ref, noise, loss):
    bad = ~torch.isfinite(g)
    hit = [names[i] for i, (o, n) in enumerate(offsets) if bool(bad[o:o + n].any())]
    err = float((g - ref).norm() / ref.norm())
    print("%-22s rep %d: eager-vs-eager %.2e  replay-vs-eager %.2e  loss %.6f  non-finite parameters: %d %s"
          % (tag, rep, noise, err, float(loss), len(hit), hit[:3]), flush=True)
    return len(hit)


def manual(tag, rep, views, cat_out, upd):
    torch.manual_seed(0)
    m = utils.build_network(CLASSES, ARCH, input_channels=3).to(dev).to(memory_format=FMT)
    fs = FlatState(m) if views else None
    params = [p for p in m.parameters() if p.requires_grad]
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    offs, o = [], 0
    for p in params:
        offs.append((o, p.numel())); o += p.numel()
    out_buf = fs.flat_g if (views and cat_out) else None
    loss_fn = TorchLoss()

    def step():
        for p in params:
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            out = m(X)
        loss = loss_fn(Y, out).mean()
        loss.backward()
        if out_buf is not None:
            torch._foreach_copy_([fs.flat_g[o_:o_ + n].view(p.shape) for p, (o_, n) in zip(params, fs.offsets)], [p.grad.contiguous() for p in params])
            return loss.detach(), out_buf
        return loss.detach(), torch.cat([p.grad.reshape(-1) for p in params])

    def flat_of(g):      # the aligned flat buffer -> packed order, for the report
        if out_buf is None:
            return g
        return torch.cat([g[o_:o_ + n] for o_, n in fs.offsets])

    snap = fs.flat_p.clone() if upd else None
    snap_buf = [b.clone() for b in m.buffers()]
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            _, g = step()
            if upd:
                fs.flat_v.mul_(0.9).sub_(fs.flat_g * 0.1)
                fs.flat_p.add_(fs.flat_v)
    torch.cuda.current_stream().wait_stream(side)
    if upd:
        fs.flat_p.copy_(snap); fs.flat_v.zero_()
    for b, sb in zip(m.buffers(), snap_buf):
        b.copy_(sb)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        l, gcap = step()
    _, r1 = step(); ref = flat_of(r1).clone()
    _, r2 = step(); noise = float((flat_of(r2) - ref).norm() / ref.norm())
    gr.replay(); torch.cuda.synchronize()
    return report(tag, rep, names, offs, flat_of(gcap), ref, noise, l)


def trainer(tag, rep, plain_params):
    torch.manual_seed(0)
    m = utils.build_network(CLASSES, ARCH, input_channels=3).to(dev).to(memory_format=FMT)
    saved = engine.FlatState
    if plain_params:
        engine.FlatState = PlainState
    try:
        t = Trainer(m, {"l2norm": (TorchLoss(), 1.0)}, {}, lr=0.1, momentum=0.9, clipnorm=10.0, autocast_dtype=torch.bfloat16)
    finally:
        engine.FlatState = saved
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    ok = t.enable_graphs(X, Y, validate=1, allow_autocast=True)
    info = getattr(t, "graph_validation", None)
    g = t.flat.flat_g
    bad = ~torch.isfinite(g)
    hit = [names[i] for i, (o, n) in enumerate(t.flat.offsets) if bool(bad[o:o + n].any())]
    print("%-22s rep %d: capture accepted=%s %s  non-finite parameters: %d %s" % (tag, rep, ok, info, len(hit), hit[:3]), flush=True)
    return 0 if ok else 1


tot = {}
for rep in range(REPS):
    for tag, fn in (("plain", lambda r: manual("plain", r, False, False, False)),
                    ("views", lambda r: manual("views", r, True, False, False)),
                    ("views+out", lambda r: manual("views+out", r, True, True, False)),
                    ("views+upd", lambda r: manual("views+upd", r, True, True, True)),
                    ("trainer", lambda r: trainer("trainer", r, False)),
                    ("trainer/plain-params", lambda r: trainer("trainer/plain-params", r, True))):
        try:
            bad = fn(rep)
        except Exception as e:
            print("%-22s rep %d: FAILED %s: %s" % (tag, rep, type(e).__name__, str(e)[:300]), flush=True)
            bad = -1
        tot.setdefault(tag, []).append(bad)
print("summary (per repetition: parameters with non-finite replayed gradients; trainer rows: 1 = capture rejected):", tot)
