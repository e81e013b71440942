"""Row pitch alignment: all-pairs distances + ranking at N = 24,633 (NABirds test set: odd) with a contiguous [N, N] output against a
pitch rounded up to a multiple of 4 elements (16 bytes)."""
import sys; sys.path[:0]=["semantic-embeddings_amd","."]
import torch, sehip
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps
for n in (24633, 5794, 10001):
    x=torch.randn(n,128,device="cuda"); sehip.normalize_rows_(x)
    np_=(n+3)//4*4
    for name,pitch in (("contiguous",n),("padded pitch",np_)):
        pd=torch.empty((n,pitch),device="cuda")[:, :n]
        rk=torch.empty((n,pitch),dtype=torch.int32,device="cuda")[:, :n]
        t1=timeit(lambda: sehip.pairwise_dist(x,x,metric=sehip.METRIC_COSINE,out=pd))
        t2=timeit(lambda: sehip.rank_rows(pd,out=rk))
        print("n=%d %-13s pdist %.3f ms  rank %.3f ms"%(n,name,t1,t2),flush=True)
