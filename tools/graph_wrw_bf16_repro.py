"""Minimal form of the bf16 + HIP-graph fault (profiles/r05_g_bf16_graph_bisect.txt): ONE call of
aten::convolution_backward on bf16 channels_last tensors, captured in a torch.cuda.graph.  With grad_weight requested its replay returns a
wrong grad_bias from the second replay on -- but only if ordinary tensor work runs between the replays."""
import torch
dev = torch.device("cuda")
CL = torch.channels_last
N, C, H = 128, 32, 16

def run(name, dtype, fmt, keep_all, check_every, mask=(True, True, True)):
    torch.manual_seed(0)
    x = torch.randn(N, C, H, H, device=dev).to(dtype).contiguous(memory_format=fmt)
    w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).to(dtype).contiguous(memory_format=fmt)
    go = torch.randn(N, C, H, H, device=dev).to(dtype).contiguous(memory_format=fmt)
    keep = []
    def fn():
        outs = torch.ops.aten.convolution_backward(go, x, w, [C], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, list(mask))
        if keep_all: keep.append(outs)
        return outs[2]
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): ref = fn().clone()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    keep.clear()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    errs = []
    reff = ref.float()
    for r in range(6):
        g.replay(); torch.cuda.synchronize()
        if check_every or r == 5:
            errs.append("%.1e" % float((out.float() - reff).abs().max() / (reff.abs().max() + 1e-30)))
    print("%-64s %s" % (name, errs), flush=True)

bf, f32 = torch.bfloat16, torch.float32
run("bf16 nhwc, other outputs dropped, checked after every replay", bf, CL, False, True)
run("bf16 nhwc, other outputs dropped, checked after 6 replays", bf, CL, False, False)
run("bf16 nhwc, all outputs kept alive", bf, CL, True, True)
run("bf16 nhwc, mask = bias only", bf, CL, False, True, (False, False, True))
run("bf16 nhwc, mask = weight + bias", bf, CL, False, True, (False, True, True))
run("bf16 nhwc, mask = input + bias", bf, CL, False, True, (True, False, True))
run("bf16 nchw, other outputs dropped", bf, torch.contiguous_format, False, True)
run("fp32 nhwc, other outputs dropped", f32, CL, False, True)
run("fp16 nhwc, other outputs dropped", torch.float16, CL, False, True)
