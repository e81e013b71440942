import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semantic-embeddings_amd"), ROOT]
import torch, sehip, numpy as np
def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps*1e3
for C,B in ((100,128),(1000,128),(1000,1024)):
    E=torch.nn.functional.normalize(torch.randn(C,C,device="cuda"),dim=-1)
    yp=torch.nn.functional.normalize(torch.randn(B,C,device="cuda"),dim=-1)
    yl=torch.randint(0,C,(B,),device="cuda")
    L=sehip.lib(); from sehip._lib import ptr, stream_ptr
    import ctypes
    loss=torch.empty(B,device="cuda"); aux=torch.empty(int(L.se_devise_aux_floats(B,C)),device="cuda"); dp=torch.empty(B,C,device="cuda")
    f=lambda: L.se_devise_loss_fwd(ptr(yp),C,ptr(yl),None,0,ptr(E),C,B,C,C,ctypes.c_float(0.1),ptr(loss),ptr(aux),stream_ptr())
    g=lambda: L.se_devise_loss_bwd(ptr(yl),None,0,ptr(E),C,None,ctypes.c_float(1.0/B),B,C,C,ptr(aux),ptr(dp),C,stream_ptr())
    print(C,B,"fwd %.1f us  bwd %.1f us"%(timeit(f),timeit(g)))
