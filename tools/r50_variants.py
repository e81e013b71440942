"""Tuning aid (round 5): forward + backward of the ResNet-50 embedding step, batch 128, channels_last, under
  autocast      bf16 autocast over fp32 parameters (what the trainer runs)
  pure          bf16 parameters (BatchNorm affine / statistics fp32), bf16 input, no autocast: no per-use weight casts
  *_nativebn    the same with BatchNorm through PyTorch's own kernels instead of MIOpen's (cudnn flag off around the call)
usage: python tools/r50_variants.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-embeddings_amd")); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import utils
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
CL = torch.channels_last
E = F.normalize(torch.randn(200, 200, device=dev), dim=-1)
y = torch.randint(0, 200, (B,), device=dev)

_bn_fwd = torch.nn.modules.batchnorm._BatchNorm.forward
def native_bn(self, x):
    with torch.backends.cudnn.flags(enabled=False):
        return _bn_fwd(self, x)

def run(name, pure, nativebn):
    torch.nn.modules.batchnorm._BatchNorm.forward = native_bn if nativebn else _bn_fwd
    torch.manual_seed(0)
    m = utils.build_network(200, "resnet-50", input_channels=3).to(dev).to(memory_format=CL)
    x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=CL)
    if pure:
        m = m.bfloat16()
        for mod in m.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm): mod.float()
        x = x.bfloat16()
    params = [p for p in m.parameters() if p.requires_grad]
    def step():
        for p in params: p.grad = None
        if pure:
            out = m(x)
        else:
            with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
                out = m(x)
        loss = (1 - (F.normalize(out.float(), dim=-1) * E[y]).sum(-1)).mean()
        loss.backward()
        return loss
    for _ in range(6): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(15): l = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 15 * 1e3
    print("%-22s %7.2f ms fwd+bwd  %7.0f img/s  loss %.4f" % (name, ms, B / ms * 1e3, float(l)), flush=True)
    torch.nn.modules.batchnorm._BatchNorm.forward = _bn_fwd

for name, pure, nb in (("autocast", False, False), ("pure", True, False), ("autocast_nativebn", False, True), ("pure_nativebn", True, True)):
    try:
        run(name, pure, nb)
    except Exception as e:
        print(name, "FAILED", type(e).__name__, str(e)[:300], flush=True)
