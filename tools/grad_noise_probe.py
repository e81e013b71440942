"""Where does the run-to-run difference of the bf16 ResNet-50 gradient come from?  Per-parameter relative difference of two
eager backward passes on the same batch (autocast bf16, channels_last)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-embeddings_amd")); sys.path.insert(0, ROOT)
import utils
import torch.nn.functional as F
torch.manual_seed(0)
B, classes = 32, 200
m = utils.build_network(classes, "resnet-50", input_channels=3).cuda().to(memory_format=torch.channels_last)
x = torch.randn(B, 3, 224, 224, device="cuda").contiguous(memory_format=torch.channels_last)
E = F.normalize(torch.randn(classes, classes, device="cuda"), dim=-1)
y = torch.randint(0, classes, (B,), device="cuda")
names = [n for n, p in m.named_parameters()]
params = [p for n, p in m.named_parameters()]
def step(dtype):
    for p in params: p.grad = None
    if dtype is None: out = m(x)
    else:
        with torch.autocast("cuda", dtype=dtype, cache_enabled=False): out = m(x)
    loss = (1 - (F.normalize(out.float(), dim=-1) * E[y]).sum(-1)).mean()
    loss.backward()
    return [p.grad.detach().float().clone() for p in params]
for dtype in (None, torch.bfloat16):
    for _ in range(2): step(dtype)
    a, b = step(dtype), step(dtype)
    tot_d = sum(float(((u - v) ** 2).sum()) for u, v in zip(a, b)) ** 0.5
    tot_n = sum(float((u ** 2).sum()) for u in a) ** 0.5
    print("dtype", dtype, "total rel diff %.3e  |g| %.3e" % (tot_d / tot_n, tot_n))
    rows = []
    for n, u, v in zip(names, a, b):
        rows.append((float((u - v).norm()), float(u.norm()), n, tuple(u.shape)))
    rows.sort(reverse=True)
    for d, nn_, n, shp in rows[:8]:
        print("   %-40s %-18s |diff| %.3e  |g| %.3e  rel %.2e" % (n, shp, d, nn_, d / max(nn_, 1e-30)))
    if dtype is not None:
        f32 = step(None)
        tot = sum(float(((u - v) ** 2).sum()) for u, v in zip(a, f32)) ** 0.5
        print("   bf16 vs fp32 gradient: rel %.3e" % (tot / sum(float((u ** 2).sum()) for u in f32) ** 0.5))
