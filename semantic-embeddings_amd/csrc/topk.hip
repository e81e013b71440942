// topk.hip -- k nearest columns per row, per-shard top-k merge, and the fused
// query-tile driver (SURVEY.md section 2.2 K9 "fused top-k", K12; section 8e sharded gallery).
//
// se_topk_rows : one workgroup per row; exact radix SELECT of the k-th smallest canonical key
//                (3 histogram passes of 11/11/10 bits, LDS atomics), then one collection pass
//                that takes every key below the threshold plus the lowest-index ties, then an
//                in-LDS bitonic sort of the k survivors on the 64-bit composite (key, index).
//                The result is identical to the first k entries of the canonical full ranking.
// se_topk_merge: the k-way merge that follows the RCCL all-gather of per-shard lists: all
//                parts*k candidates of a query are sorted in LDS on (key, global index).
#include "se_common.h"

namespace se {

constexpr int TK_THREADS = 512;
constexpr int TK_WAVES = TK_THREADS / WAVE;
constexpr int TK_BITS = 11;
constexpr int TK_NB = 1 << TK_BITS;

__device__ __forceinline__ float key_to_float(uint32_t key)
{
    const uint32_t u = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key;
    return __uint_as_float(u);
}

// In-LDS bitonic sort of P (power of two) uint64 values, ascending, by the whole workgroup.
__device__ __forceinline__ void bitonic_sort_u64(uint64_t *v, int P, int nthreads)
{
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < P / 2; t += nthreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j clear
                const int l = i | j;
                const bool up = ((i & k) == 0);
                const uint64_t a = v[i], b = v[l];
                if ((a > b) == up) { v[i] = b; v[l] = a; }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(TK_THREADS) void topk_rows_kernel(const float *__restrict__ pdist, int64_t ldp,
                                                               int64_t Q, int N, int64_t col_offset, int k, int P,
                                                               float *__restrict__ out_d, int32_t *__restrict__ out_i)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t tk_lds64[];
    uint64_t *cand = tk_lds64;                          // [P]
    uint32_t *hist = (uint32_t *)(tk_lds64 + P);        // [TK_NB]
    uint32_t *wcnt = hist + TK_NB;                      // [TK_WAVES + 1]
    uint32_t *ctl = wcnt + TK_WAVES + 1;                // [4]: prefix, remaining k, n_lt cursor
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int chunk = (N + TK_WAVES - 1) / TK_WAVES;
    chunk = (chunk + WAVE - 1) / WAVE * WAVE;
    const int beg = wave * chunk;
    const int end = (beg + chunk < N) ? (beg + chunk) : N;

    for (int64_t row = blockIdx.x; row < Q; row += gridDim.x) {
        const float *drow = pdist + row * ldp;

        // ---------- radix select: find the canonical key of the k-th smallest ----------
        uint32_t prefix = 0, prefix_mask = 0;
        uint32_t remaining = (uint32_t)k;  // rank (1-based) of the wanted key among keys matching prefix
        for (int pass = 0; pass < 3; pass++) {
            const int shift = (pass == 0) ? 21 : (pass == 1 ? 10 : 0);
            const int nb = (pass == 2) ? 10 : 11;
            const uint32_t dmask = (1u << nb) - 1u;
            __syncthreads();
            for (int i = tid; i < TK_NB; i += TK_THREADS) hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < N; i += TK_THREADS) {
                const uint32_t key = canon_key(drow[i]);
                if ((key & prefix_mask) == prefix) atomicAdd(&hist[(key >> shift) & dmask], 1u);
            }
            __syncthreads();
            // wave 0 scans the histogram to find the digit holding the `remaining`-th key
            if (wave == 0) {
                constexpr int PER = TK_NB / WAVE;  // 32 bins per lane
                uint32_t local = 0;
                for (int d = 0; d < PER; d++) local += hist[lane * PER + d];
                uint32_t incl = local;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t v = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += v;
                }
                const uint32_t excl = incl - local;
                if (remaining > excl && remaining <= incl) {  // exactly one lane
                    uint32_t run = excl;
                    for (int d = 0; d < PER; d++) {
                        const uint32_t c = hist[lane * PER + d];
                        if (remaining <= run + c) {
                            ctl[0] = (uint32_t)(lane * PER + d);
                            ctl[1] = remaining - run;
                            break;
                        }
                        run += c;
                    }
                }
            }
            __syncthreads();
            prefix |= ctl[0] << shift;
            prefix_mask |= dmask << shift;
            remaining = ctl[1];
        }
        const uint32_t kth = prefix;          // exact key of the k-th smallest element
        const uint32_t need_ties = remaining; // how many elements equal to kth belong to the top-k
        const uint32_t n_lt = (uint32_t)k - need_ties;

        // ---------- collect: keys < kth (any order) + the `need_ties` lowest-index ties ----------
        __syncthreads();
        if (tid == 0) ctl[2] = 0;
        // per-wave tie counts (index order matters for ties)
        uint32_t myties = 0;
        for (int i = beg + lane; i < end; i += WAVE) myties += (canon_key(drow[i]) == kth) ? 1u : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) myties += __shfl_xor(myties, off, 64);
        if (lane == 0) wcnt[wave] = myties;
        for (int i = tid; i < P; i += TK_THREADS) cand[i] = ~0ull;
        __syncthreads();
        uint32_t tie_base = 0;
        for (int w = 0; w < wave; w++) tie_base += wcnt[w];
        const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        for (int i0 = beg; i0 < end; i0 += WAVE) {
            const int i = i0 + lane;
            const bool valid = i < end;
            const uint32_t key = valid ? canon_key(drow[i]) : 0xFFFFFFFFu;
            const bool is_lt = valid && key < kth;
            const bool is_tie = valid && key == kth;
            const uint64_t tie_ballot = __ballot(is_tie);
            if (is_lt) {
                const uint32_t slot = atomicAdd(&ctl[2], 1u);
                cand[slot] = ((uint64_t)key << 32) | (uint32_t)i;
            }
            if (is_tie) {
                const uint32_t ord = tie_base + (uint32_t)__popcll(tie_ballot & lt_mask);
                if (ord < need_ties) cand[n_lt + ord] = ((uint64_t)key << 32) | (uint32_t)i;
            }
            tie_base += (uint32_t)__popcll(tie_ballot);
        }
        __syncthreads();

        // ---------- sort the k survivors on (key, index) and emit ----------
        bitonic_sort_u64(cand, P, TK_THREADS);
        for (int r = tid; r < k; r += TK_THREADS) {
            const uint64_t c = cand[r];
            out_d[row * k + r] = key_to_float((uint32_t)(c >> 32));
            out_i[row * k + r] = (int32_t)(col_offset + (int64_t)(uint32_t)c);
        }
    }
}

__global__ __launch_bounds__(256) void topk_merge_kernel(const float *__restrict__ d, const int32_t *__restrict__ idx,
                                                         int parts, int64_t Q, int k, int P,
                                                         float *__restrict__ out_d, int32_t *__restrict__ out_i)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t mg_lds[];
    const int m = parts * k;
    for (int64_t row = blockIdx.x; row < Q; row += gridDim.x) {
        __syncthreads();
        for (int t = threadIdx.x; t < P; t += 256) {
            uint64_t v = ~0ull;
            if (t < m) {
                const int p = t / k, r = t - p * k;
                const int64_t src = ((int64_t)p * Q + row) * k + r;
                v = ((uint64_t)canon_key(d[src]) << 32) | (uint32_t)idx[src];
            }
            mg_lds[t] = v;
        }
        bitonic_sort_u64(mg_lds, P, 256);
        for (int r = threadIdx.x; r < k; r += 256) {
            const uint64_t c = mg_lds[r];
            out_d[row * k + r] = key_to_float((uint32_t)(c >> 32));
            out_i[row * k + r] = (int32_t)(uint32_t)c;
        }
    }
}

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

}  // namespace se

using namespace se;

extern "C" int se_topk_rows(const float *pdist, int64_t ldp, int64_t q, int64_t n, int64_t col_offset, int k,
                            float *out_d, int32_t *out_i, se_stream_t stream)
{
    if (q < 0 || n < 0 || n > 0x7FFFFFFFll) return fail(SE_ERR_INVALID, "se_topk_rows: bad shape");
    if (k < 1 || k > SE_TOPK_MAX || k > n) return fail(SE_ERR_INVALID, "se_topk_rows: need 1 <= k <= min(n, %d), got k=%d n=%lld", SE_TOPK_MAX, k, (long long)n);
    if (q == 0) return SE_OK;
    if (!pdist || !out_d || !out_i || ldp < n) return fail(SE_ERR_INVALID, "se_topk_rows: bad argument");
    const int P = next_pow2(k);
    const size_t lds = (size_t)P * 8 + (TK_NB + TK_WAVES + 1 + 4) * sizeof(uint32_t);
    const int64_t grid = q < 2048 ? q : 2048;
    hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)grid), dim3(TK_THREADS), lds, (hipStream_t)stream,
                       pdist, ldp, q, (int)n, col_offset, k, P, out_d, out_i);
    SE_LAUNCH_CHECK();
    return SE_OK;
}

extern "C" int se_topk_merge(const float *d, const int32_t *idx, int parts, int64_t q, int k, float *out_d,
                             int32_t *out_i, se_stream_t stream)
{
    if (parts < 1 || q < 0 || k < 1) return fail(SE_ERR_INVALID, "se_topk_merge: bad shape");
    if ((int64_t)parts * k > (int64_t)SE_TOPK_MAX * 4) return fail(SE_ERR_UNSUPPORTED, "se_topk_merge: parts*k = %lld exceeds %d", (long long)parts * k, SE_TOPK_MAX * 4);
    if (q == 0) return SE_OK;
    if (!d || !idx || !out_d || !out_i) return fail(SE_ERR_INVALID, "se_topk_merge: null pointer");
    const int P = next_pow2(parts * k);
    const size_t lds = (size_t)P * 8;
    const int64_t grid = q < 4096 ? q : 4096;
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, d, idx, parts, q, k, P, out_d, out_i);
    SE_LAUNCH_CHECK();
    return SE_OK;
}

// ------------------------------------------------------------------------------------------------
// fused driver: distance slab -> top-k, query tile by query tile
// ------------------------------------------------------------------------------------------------
static int64_t topk_qtile(int64_t q, int64_t n)
{
    // keep the slab <= 2 GiB and a multiple of 128 rows
    int64_t rows = ((int64_t)2 << 30) / (n * 4);
    rows = rows / 128 * 128;
    if (rows < 128) rows = 128;
    if (rows > q) rows = q;
    return rows;
}

extern "C" int64_t se_retrieve_topk_workspace_bytes(int64_t q, int64_t n, int k)
{
    (void)k;
    if (q <= 0 || n <= 0) return 0;
    return topk_qtile(q, n) * n * 4;
}

extern "C" int se_retrieve_topk(const float *queries, int64_t ldq, const float *gallery, int64_t ldg,
                                const float *sqq, const float *sqg, int64_t q, int64_t n, int64_t d,
                                int metric, int64_t col_offset, int k, float *out_d, int32_t *out_i,
                                void *workspace, int64_t workspace_bytes, se_stream_t stream)
{
    if (q < 0 || n <= 0 || d <= 0) return fail(SE_ERR_INVALID, "se_retrieve_topk: bad shape");
    if (q == 0) return SE_OK;
    const int64_t need = se_retrieve_topk_workspace_bytes(q, n, k);
    if (!workspace || workspace_bytes < need) return fail(SE_ERR_WORKSPACE, "se_retrieve_topk: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    const int64_t qt = topk_qtile(q, n);
    float *slab = (float *)workspace;
    for (int64_t q0 = 0; q0 < q; q0 += qt) {
        const int64_t rows = (q - q0 < qt) ? (q - q0) : qt;
        int rc = se_pairwise_dist(queries + q0 * ldq, ldq, gallery, ldg, sqq ? sqq + q0 : nullptr, sqg, rows, n, d,
                                  metric, nullptr, 0, slab, n, stream);
        if (rc != SE_OK) return rc;
        rc = se_topk_rows(slab, n, rows, n, col_offset, k, out_d + q0 * k, out_i + q0 * k, stream);
        if (rc != SE_OK) return rc;
    }
    return SE_OK;
}
