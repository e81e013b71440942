"""Does a HIP-graph replay of a bf16 backward reproduce the eager gradient on this torch / ROCm stack?
Variants: autocast (weight-cast cache on / off) vs a model whose weights ARE bf16 (no autocast casts at all)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-embeddings_amd")); sys.path.insert(0, ROOT)
import utils
import torch.nn.functional as F

def run(arch, size, classes, B, mode):
    torch.manual_seed(0)
    m = utils.build_network(classes, arch, input_channels=3).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(B, 3, size, size, device="cuda").contiguous(memory_format=torch.channels_last)
    E = F.normalize(torch.randn(classes, classes, device="cuda"), dim=-1)
    y = torch.randint(0, classes, (B,), device="cuda")
    if mode == "pure":
        m = m.bfloat16()
        for mod in m.modules():            # BatchNorm statistics / affine in fp32 like autocast keeps them
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.float()
        x = x.bfloat16()
    params = [p for p in m.parameters() if p.requires_grad]
    def step():
        for p in params: p.grad = None
        if mode.startswith("autocast"):
            with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=(mode == "autocast_cache")):
                out = m(x)
        else:
            out = m(x)
        loss = (1 - (F.normalize(out.float(), dim=-1) * E[y]).sum(-1)).mean()
        loss.backward()
        return loss.detach(), torch.cat([p.grad.reshape(-1).float() for p in params])
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side)
    _, ref = step(); ref = ref.clone()
    _, ref2 = step()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        l, flat = step()
    errs = []
    for _ in range(4):
        g.replay(); torch.cuda.synchronize()
        errs.append(float((flat - ref).norm() / ref.norm()))
    print("%-12s %-16s eager-vs-eager %.2e   replay-vs-eager %s   loss %.4f finite=%s" % (arch, mode, float((ref2 - ref).norm() / ref.norm()),
          ["%.2e" % e for e in errs], float(l), bool(torch.isfinite(flat).all())), flush=True)

for arch, size, classes, B in (("resnet-110-fc", 32, 100, 128), ("resnet-50", 224, 200, 32)):
    for mode in ("fp32", "autocast_cache", "autocast_nocache", "pure"):
        try:
            run(arch, size, classes, B, mode)
        except Exception as e:
            print(arch, mode, "FAILED:", type(e).__name__, str(e)[:200], flush=True)
