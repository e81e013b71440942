// rownorm.hip -- bit-exact float32 row square-norms / row normalisation.
//
// Replaces `features /= np.linalg.norm(features, axis=-1, keepdims=True)` (evaluate_retrieval.py:58)
// and `sqnorm = np.sum(features ** 2, axis=-1)` (evaluate_retrieval.py:61).
//
// NumPy reduces a contiguous float32 row with its pairwise summation (numpy 2.2,
// loops_utils.h.src: blocks of <= 128 elements summed with 8 interleaved accumulators, combined
// as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), tail added sequentially, larger rows split recursively
// at n/2 rounded down to a multiple of 8).  The order of the additions decides the rounding, so
// the kernel reproduces exactly that tree: the 8 accumulators of a leaf block are 8 lanes'
// private chains... kept simple here: one lane walks one row in NumPy's order.  The rows are
// staged through LDS with coalesced loads first, so HBM sees each byte once, streaming.
// x*x, the adds, sqrtf and the division are all IEEE round-to-nearest (this file is compiled with
// -ffp-contract=off; hipcc's default keeps fp32 sqrt/div correctly rounded).
#include "se_common.h"

namespace se {

constexpr int RN_ROWS = 64;     // rows per workgroup == threads per workgroup
constexpr int RN_CHUNK = 128;   // staged columns per pass (NumPy's PW_BLOCKSIZE)
constexpr int RN_LD = RN_CHUNK + 1;

// --- NumPy's leaf block (n <= 128) over LDS-resident data with stride 1 ---
__device__ __forceinline__ float np_leaf_sqsum(const float *a, int n)
{
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; i++) res = res + a[i] * a[i];
        return res;
    }
    float r[8];
#pragma unroll
    for (int u = 0; u < 8; u++) r[u] = a[u] * a[u];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) r[u] = r[u] + a[i + u] * a[i + u];
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + a[i] * a[i];
    return res;
}

// Walks NumPy's recursion tree over [0, d) for one row.  Leaves are visited left to right; each
// leaf (<= 128 elements, arbitrary offset) is staged into this lane's LDS row by the whole
// workgroup (coalesced), then summed by the lane.  All lanes of the workgroup execute the same
// tree (it depends on d only), so the barriers inside are uniform.
struct Frame { int off, n, stage; float left; };

__device__ float np_row_sqsum_staged(const float *__restrict__ x, int64_t ldx, int64_t row0, int64_t nrows,
                                     int d, float *lds /* [RN_ROWS][RN_LD] */)
{
    const int tid = threadIdx.x;
    Frame st[32];
    int sp = 0;
    st[0].off = 0; st[0].n = d; st[0].stage = 0; st[0].left = 0.f;
    float ret = 0.f;
    while (sp >= 0) {
        Frame &f = st[sp];
        if (f.stage == 0) {
            if (f.n <= RN_CHUNK) {
                // stage columns [off, off+n) of the 64 rows
                __syncthreads();
                for (int idx = tid; idx < RN_ROWS * f.n; idx += RN_ROWS) {
                    const int r = idx / f.n, c = idx - r * f.n;
                    lds[r * RN_LD + c] = (row0 + r < nrows) ? x[(row0 + r) * ldx + f.off + c] : 0.f;
                }
                __syncthreads();
                ret = np_leaf_sqsum(lds + tid * RN_LD, f.n);
                sp--;
            } else {
                int n2 = f.n / 2;
                n2 -= n2 % 8;
                f.stage = 1;
                st[sp + 1].off = f.off; st[sp + 1].n = n2; st[sp + 1].stage = 0;
                sp++;
            }
        } else if (f.stage == 1) {
            f.left = ret;
            f.stage = 2;
            int n2 = f.n / 2;
            n2 -= n2 % 8;
            st[sp + 1].off = f.off + n2; st[sp + 1].n = f.n - n2; st[sp + 1].stage = 0;
            sp++;
        } else {
            ret = f.left + ret;
            sp--;
        }
    }
    return ret;
}

template <bool NORMALIZE>
__global__ __launch_bounds__(RN_ROWS) void rownorm_kernel(float *__restrict__ x, int64_t ldx, int64_t n, int d,
                                                          float *__restrict__ sq)
{
    __shared__ float lds[RN_ROWS * RN_LD];
    __shared__ float snorm[RN_ROWS];
    const int64_t row0 = (int64_t)blockIdx.x * RN_ROWS;
    const float ss = np_row_sqsum_staged(x, ldx, row0, n, d, lds);
    if (!NORMALIZE) {
        if (row0 + threadIdx.x < n) sq[row0 + threadIdx.x] = ss;
        return;
    }
    snorm[threadIdx.x] = sqrtf(ss);
    __syncthreads();
    const int64_t rows = (n - row0 < RN_ROWS) ? (n - row0) : RN_ROWS;
    for (int64_t r = 0; r < rows; r++) {
        const float nrm = snorm[r];
        float *p = x + (row0 + r) * ldx;
        for (int c = threadIdx.x; c < d; c += RN_ROWS) p[c] = p[c] / nrm;
    }
}

}  // namespace se

using namespace se;

extern "C" int se_row_sqnorm(const float *x, int64_t ldx, int64_t n, int64_t d, float *sq, se_stream_t stream)
{
    if (n < 0 || d <= 0 || d > 0x7FFFFFFF) return fail(SE_ERR_INVALID, "se_row_sqnorm: bad shape");
    if (n == 0) return SE_OK;
    if (!x || !sq || ldx < d) return fail(SE_ERR_INVALID, "se_row_sqnorm: bad argument");
    hipLaunchKernelGGL(rownorm_kernel<false>, dim3((unsigned)((n + RN_ROWS - 1) / RN_ROWS)), dim3(RN_ROWS), 0,
                       (hipStream_t)stream, const_cast<float *>(x), ldx, n, (int)d, sq);
    SE_LAUNCH_CHECK();
    return SE_OK;
}

extern "C" int se_normalize_rows(float *x, int64_t ldx, int64_t n, int64_t d, se_stream_t stream)
{
    if (n < 0 || d <= 0 || d > 0x7FFFFFFF) return fail(SE_ERR_INVALID, "se_normalize_rows: bad shape");
    if (n == 0) return SE_OK;
    if (!x || ldx < d) return fail(SE_ERR_INVALID, "se_normalize_rows: bad argument");
    hipLaunchKernelGGL(rownorm_kernel<true>, dim3((unsigned)((n + RN_ROWS - 1) / RN_ROWS)), dim3(RN_ROWS), 0,
                       (hipStream_t)stream, x, ldx, n, (int)d, (float *)nullptr);
    SE_LAUNCH_CHECK();
    return SE_OK;
}
