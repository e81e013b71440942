// rownorm.hip -- bit-exact float32 row square-norms / row normalisation.
//
// Replaces `features /= np.linalg.norm(features, axis=-1, keepdims=True)` (evaluate_retrieval.py:58)
// and `sqnorm = np.sum(features ** 2, axis=-1)` (evaluate_retrieval.py:61).
//
// NumPy reduces a contiguous float32 row with its pairwise summation (numpy 2.2,
// loops_utils.h.src: blocks of <= 128 elements summed with 8 interleaved accumulators, combined
// as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), tail added sequentially, larger rows split recursively
// at n/2 rounded down to a multiple of 8).  The order of the additions decides the rounding, so
// the kernels reproduce exactly that tree.  Two kernels:
//   * rownorm_kernel (D < 8, D > 4096): one LANE walks one row in NumPy's order; the rows of a workgroup are staged
//     through LDS with coalesced loads first, so HBM sees each byte once, streaming.
//   * rownorm_small_kernel (8 <= D <= 248): 8 or 4 rows per wave at a time, one 8-lane group per NumPy leaf (below).
//   * rownorm_wave_kernel (249 <= D <= 4096): one WAVE per row.  The row is staged in LDS with coalesced 16-byte
//     loads; the tree's leaves (host-computed from D: the tree depends on nothing else) are dealt eight at a time to
//     the eight 8-lane groups of the wave, lane u of a group running accumulator r[u] of its leaf (<= 16 sequential
//     adds); the ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) combination is a 3-step xor butterfly inside the group (IEEE
//     addition commutes, the association is NumPy's), the group's first lane adds the leaf's tail, and lane 0 folds
//     the <= 64 leaf sums with the recursion's postfix program.  The one-lane-per-row kernel needed 132 VGPRs +
//     scratch for its recursion stack and crawled at ~0.36 TB/s on D = 1000 rows (one lane = 4 KB of sequential work).
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
                wg_barrier();
                for (int idx = tid; idx < RN_ROWS * f.n; idx += RN_ROWS) {
                    const int r = idx / f.n, c = idx - r * f.n;
                    lds[r * RN_LD + c] = (row0 + r < nrows) ? x[(row0 + r) * ldx + f.off + c] : 0.f;
                }
                wg_barrier();
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
    wg_barrier();
    const int64_t rows = (n - row0 < RN_ROWS) ? (n - row0) : RN_ROWS;
    for (int64_t r = 0; r < rows; r++) {
        const float nrm = snorm[r];
        float *p = x + (row0 + r) * ldx;
        for (int c = threadIdx.x; c < d; c += RN_ROWS) p[c] = p[c] / nrm;
    }
}

// ---- one wave per row (256 <= D <= 4096) ------------------------------------------------------------------------------------
constexpr int RW_MAX_LEAVES = 64, RW_MAX_D = 4096, RW_MIN_D = 249;   // (from 249 columns on a row has three leaves or more, all >= 64 long)
struct RowTree {
    int nleaves, nprog;
    uint16_t off[RW_MAX_LEAVES];           // first column of leaf i
    uint8_t len[RW_MAX_LEAVES];            // its length (8 ... 128: a row of >= 256 columns has no shorter leaf)
    uint8_t prog[2 * RW_MAX_LEAVES];       // postfix fold of the recursion: 0 = push the next leaf sum, 1 = pop b, pop a, push a + b
};

static void row_tree_build(int off, int n, RowTree &t)
{
    if (n <= RN_CHUNK) {
        t.off[t.nleaves] = (uint16_t)off;
        t.len[t.nleaves] = (uint8_t)n;
        t.nleaves++;
        t.prog[t.nprog++] = 0;
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    row_tree_build(off, n2, t);
    row_tree_build(off + n2, n - n2, t);
    t.prog[t.nprog++] = 1;
}

template <bool NORMALIZE, bool VEC>
__global__ __launch_bounds__(256) void rownorm_wave_kernel(float *__restrict__ x, int64_t ldx, int64_t n, int d, float *__restrict__ sq,
                                                           const RowTree t)
{
    extern __shared__ __attribute__((aligned(16))) float rw_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int dpad = (d + 3) & ~3;
    float *row = rw_lds + wave * (dpad + RW_MAX_LEAVES + 64);        // [dpad] staged row | [64] leaf sums | [64] fold stack
    float *leaf = row + dpad, *stack = leaf + RW_MAX_LEAVES;
    const int grp = lane >> 3, u = lane & 7;
    for (int64_t r = (int64_t)blockIdx.x * nw + wave; r < n; r += (int64_t)gridDim.x * nw) {
        float *xr = x + r * ldx;
        // ---- stage the row (coalesced) ----
        if (VEC) {
            for (int c = lane * 4; c < d; c += 256) *(float4 *)(row + c) = *(const float4 *)(xr + c);
        } else {
            for (int c = lane; c < d; c += 64) row[c] = xr[c];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ---- leaves, eight at a time: lane (grp, u) runs accumulator u of leaf l0 + grp ----
        for (int l0 = 0; l0 < t.nleaves; l0 += 8) {
            const int li = l0 + grp;
            const bool have = li < t.nleaves;
            const int lo = have ? t.off[li] : 0, ln = have ? t.len[li] : 8;
            const float *a = row + lo;
            const int body = ln - (ln % 8);
            float acc = a[u] * a[u];
            for (int i = 8; i < body; i += 8) { const float v = a[i + u]; acc = acc + v * v; }
            // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)): xor butterfly inside the 8-lane group
            acc = acc + __shfl_xor(acc, 1, 64);
            acc = acc + __shfl_xor(acc, 2, 64);
            acc = acc + __shfl_xor(acc, 4, 64);
            if (u == 0 && have) {
                for (int i = body; i < ln; i++) { const float v = a[i]; acc = acc + v * v; }
                leaf[li] = acc;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ---- fold the leaf sums in the recursion's order (lane 0; <= 127 steps) ----
        float ss = 0.f;
        if (lane == 0) {
            int sp = 0, next = 0;
            for (int i = 0; i < t.nprog; i++) {
                if (t.prog[i] == 0) stack[sp++] = leaf[next++];
                else { const float b = stack[sp - 1], aa = stack[sp - 2]; sp--; stack[sp - 1] = aa + b; }
            }
            ss = stack[0];
        }
        ss = __shfl(ss, 0, 64);
        if (!NORMALIZE) {
            if (lane == 0) sq[r] = ss;
        } else {
            const float nrm = sqrtf(ss);
            if (VEC) {
                for (int c = lane * 4; c < d; c += 256) {
                    const float4 v = *(const float4 *)(row + c);
                    *(float4 *)(xr + c) = make_float4(v.x / nrm, v.y / nrm, v.z / nrm, v.w / nrm);
                }
            } else {
                for (int c = lane; c < d; c += 64) xr[c] = row[c] / nrm;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the next row overwrites the stage
    }
}

// ---- short rows (8 <= D <= 248; the headline's D = 100): the same leaf arithmetic with SEVERAL rows per wave -------------------------
// Such a row is one NumPy leaf (D <= 128) or two (split at (D / 2) rounded down to a multiple of 8; the right one is <= 128 long up to
// D = 248, from 249 on it splits again: rownorm_wave_kernel), so the eight
// 8-lane groups of a wave take 8 or 4 rows at a time: group g runs leaf g % nl of row g / nl exactly like rownorm_wave_kernel does
// (accumulator u per lane, xor butterfly, sequential tail), a two-leaf row adds left + right, and all 64 lanes stream the rows in and
// out.  (The one-lane-per-row kernel it replaces here kept its recursion stack in scratch -- 132 VGPRs + 528 B -- and moved 0.6 TB/s.)
constexpr int RS_MIN_D = 8;
template <bool NORMALIZE>
__global__ __launch_bounds__(256) void rownorm_small_kernel(float *__restrict__ x, int64_t ldx, int64_t n, int d, float *__restrict__ sq)
{
    extern __shared__ __attribute__((aligned(16))) float rs_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int dpad = (d + 3) & ~3;
    const int nl = d > RN_CHUNK ? 2 : 1, rpi = 8 / nl;
    int n2 = d / 2;
    n2 -= n2 % 8;
    float *stage = rs_lds + wave * (8 * dpad);                          // [rpi][dpad]
    const int grp = lane >> 3, u = lane & 7;
    const int rr = grp / nl, li = grp - rr * nl;
    const int lo = li ? n2 : 0, ln = nl == 1 ? d : (li ? d - n2 : n2);
    const int body = ln - (ln % 8);
    for (int64_t r0 = ((int64_t)blockIdx.x * nw + wave) * rpi; r0 < n; r0 += (int64_t)gridDim.x * nw * rpi) {
        const int rows = (int)((n - r0 < rpi) ? (n - r0) : rpi);
        for (int q = 0; q < rows; q++) {
            const float *xr = x + (r0 + q) * ldx;
            for (int c = lane; c < d; c += 64) stage[q * dpad + c] = xr[c];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const float *a = stage + (rr < rows ? rr : 0) * dpad + lo;
        float acc = a[u] * a[u];
        for (int i = 8; i < body; i += 8) { const float v = a[i + u]; acc = acc + v * v; }
        acc = acc + __shfl_xor(acc, 1, 64);                              // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
        acc = acc + __shfl_xor(acc, 2, 64);
        acc = acc + __shfl_xor(acc, 4, 64);
        if (u == 0) for (int i = body; i < ln; i++) { const float v = a[i]; acc = acc + v * v; }
        float ss = __shfl(acc, lane & ~7, 64);                           // the leaf's sum, on every lane of its group
        if (nl == 2) ss = __shfl(ss, (grp & ~1) * 8, 64) + __shfl(ss, (grp | 1) * 8, 64);   // left + right
        if (!NORMALIZE) {
            if (u == 0 && li == 0 && rr < rows) sq[r0 + rr] = ss;
        } else {
            for (int q = 0; q < rows; q++) {
                const float nrm = sqrtf(__shfl(ss, q * nl * 8, 64));
                float *xr = x + (r0 + q) * ldx;
                for (int c = lane; c < d; c += 64) xr[c] = stage[q * dpad + c] / nrm;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // the next rows overwrite the stage
    }
}

template <bool NORMALIZE>
static int launch_rownorm_small(float *x, int64_t ldx, int64_t n, int d, float *sq, hipStream_t s)
{
    const int dpad = (d + 3) & ~3, waves = 4;
    const size_t lds = (size_t)waves * 8 * dpad * sizeof(float);          // <= 32 KB
    const int rpi = d > RN_CHUNK ? 4 : 8;
    int64_t grid = (n + (int64_t)waves * rpi - 1) / ((int64_t)waves * rpi);
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL((rownorm_small_kernel<NORMALIZE>), dim3((unsigned)grid), dim3(waves * 64), lds, s, x, ldx, n, d, sq);
    SE_LAUNCH_CHECK();
    return SE_OK;
}

template <bool NORMALIZE>
static int launch_rownorm_wave(float *x, int64_t ldx, int64_t n, int d, float *sq, hipStream_t s)
{
    RowTree t;
    t.nleaves = t.nprog = 0;
    row_tree_build(0, d, t);
    const int dpad = (d + 3) & ~3;
    const int waves = d <= 1024 ? 4 : (d <= 2048 ? 2 : 1);             // <= ~36 KB of LDS per workgroup
    const size_t lds = (size_t)waves * (dpad + RW_MAX_LEAVES + 64) * sizeof(float);
    const bool vec = (d % 4 == 0) && (ldx % 4 == 0) && ((((uintptr_t)x) & 15) == 0);
    int64_t grid = (n + waves - 1) / waves;
    if (grid > 256 * 16) grid = 256 * 16;
    if (vec) hipLaunchKernelGGL((rownorm_wave_kernel<NORMALIZE, true>), dim3((unsigned)grid), dim3(waves * 64), lds, s, x, ldx, n, d, sq, t);
    else hipLaunchKernelGGL((rownorm_wave_kernel<NORMALIZE, false>), dim3((unsigned)grid), dim3(waves * 64), lds, s, x, ldx, n, d, sq, t);
    SE_LAUNCH_CHECK();
    return SE_OK;
}

}  // namespace se

using namespace se;

extern "C" int se_row_sqnorm(const float *x, int64_t ldx, int64_t n, int64_t d, float *sq, se_stream_t stream)
{
    if (n < 0 || d <= 0 || d > 0x7FFFFFFF) return fail(SE_ERR_INVALID, "se_row_sqnorm: bad shape");
    if (n == 0) return SE_OK;
    if (!x || !sq || ldx < d) return fail(SE_ERR_INVALID, "se_row_sqnorm: bad argument");
    if (d >= RW_MIN_D && d <= RW_MAX_D) return launch_rownorm_wave<false>(const_cast<float *>(x), ldx, n, (int)d, sq, (hipStream_t)stream);
    if (d >= RS_MIN_D && d < RW_MIN_D) return launch_rownorm_small<false>(const_cast<float *>(x), ldx, n, (int)d, sq, (hipStream_t)stream);
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
    if (d >= RW_MIN_D && d <= RW_MAX_D) return launch_rownorm_wave<true>(x, ldx, n, (int)d, (float *)nullptr, (hipStream_t)stream);
    if (d >= RS_MIN_D && d < RW_MIN_D) return launch_rownorm_small<true>(x, ldx, n, (int)d, (float *)nullptr, (hipStream_t)stream);
    hipLaunchKernelGGL(rownorm_kernel<true>, dim3((unsigned)((n + RN_ROWS - 1) / RN_ROWS)), dim3(RN_ROWS), 0,
                       (hipStream_t)stream, x, ldx, n, (int)d, (float *)nullptr);
    SE_LAUNCH_CHECK();
    return SE_OK;
}
