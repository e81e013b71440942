"""Micro-repro for the bf16 + HIP-graph fault (tools/graph_bf16_bisect.py: even a plain capture of the bf16-autocast ResNet-110 step
returns non-finite conv-bias gradients): which single operation misbehaves under replay?
    python tools/graph_bias_grad_repro.py [replays]"""
import sys
import torch
import torch.nn.functional as F
dev = torch.device("cuda")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 200

def trial(name, make):
    """make() -> (fn, check): fn() runs the op on static inputs and returns the output tensor; captured once, replayed R times."""
    fn = make()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): ref = fn().clone()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    bad = wrong = 0
    for _ in range(R):
        g.replay()
        torch.cuda.synchronize()
        o = out.float()
        if not bool(torch.isfinite(o).all()): bad += 1
        elif float((o - ref.float()).abs().max()) > 0.05 * float(ref.float().abs().max()) + 1e-3: wrong += 1
    eager_bad = 0
    for _ in range(R // 4):
        o = fn().float()
        if not bool(torch.isfinite(o).all()): eager_bad += 1
    print("%-58s replays: %3d non-finite, %3d wrong of %d   eager non-finite: %d of %d" % (name, bad, wrong, R, eager_bad, R // 4), flush=True)

def sum_case(N, C, H, dtype, fmt):
    def make():
        t = torch.randn(N, C, H, H, device=dev).to(dtype).contiguous(memory_format=fmt)
        return lambda: t.sum(dim=(0, 2, 3))
    return make

def conv_case(N, C, H, dtype, fmt, autocast):
    def make():
        conv = torch.nn.Conv2d(C, C, 3, padding=1, bias=True).to(dev).to(memory_format=fmt)
        x = torch.randn(N, C, H, H, device=dev).contiguous(memory_format=fmt)
        if not autocast and dtype != torch.float32:
            conv = conv.to(dtype); x = x.to(dtype)
        def fn():
            conv.bias.grad = None; conv.weight.grad = None
            if autocast:
                with torch.autocast("cuda", dtype=dtype, cache_enabled=False):
                    y = conv(x)
            else:
                y = conv(x)
            y.float().square().mean().backward()
            return conv.bias.grad
        return fn
    return make

CL, NC = torch.channels_last, torch.contiguous_format
for (N, C, H) in ((128, 16, 32), (128, 32, 16), (128, 64, 8)):
    for dtype in (torch.bfloat16, torch.float32):
        for fmt in (NC, CL):
            tag = "%dx%dx%dx%d %s %s" % (N, C, H, H, str(dtype).split(".")[1], "nhwc" if fmt == CL else "nchw")
            trial("sum(dim=(0,2,3))        " + tag, sum_case(N, C, H, dtype, fmt))
    for fmt in (NC, CL):
        tag = "%dx%dx%dx%d %s" % (N, C, H, H, "nhwc" if fmt == CL else "nchw")
        trial("conv bias grad autocast bf16 " + tag, conv_case(N, C, H, torch.bfloat16, fmt, True))
        trial("conv bias grad pure bf16     " + tag, conv_case(N, C, H, torch.bfloat16, fmt, False))
        trial("conv bias grad fp32          " + tag, conv_case(N, C, H, torch.float32, fmt, False))
