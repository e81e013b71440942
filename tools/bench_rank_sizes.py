import sys; sys.path[:0]=["semantic-embeddings_amd","."]
import torch, sehip
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps
for q,n in ((32768,5794),(32768,10000),(16384,15000),(16384,24633),(16384,26624),(16384,30000),(16384,32768),(16384,40960),(16384,50000),(16384,53248)):
    x=torch.randn(q,n,device="cuda")
    ms=timeit(lambda: sehip.rank_rows(x))
    print("rank_rows %d x %d: %.2f ms = %.2f ps/key"%(q,n,ms,ms*1e9/(q*n)),flush=True)
