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
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>

namespace se {

constexpr int TK_THREADS = 512;
constexpr int TK_WAVES = TK_THREADS / WAVE;
constexpr int TK_BITS = 11;
constexpr int TK_NB = 1 << TK_BITS;
constexpr int32_t TK_REDO = -1;   // out_i[row, 0] marker: row to be redone by the exact radix-select kernel

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
            wg_barrier();
            for (int t = threadIdx.x; t < P / 2; t += nthreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j clear
                const int l = i | j;
                const bool up = ((i & k) == 0);
                const uint64_t a = v[i], b = v[l];
                if ((a > b) == up) { v[i] = b; v[l] = a; }
            }
        }
    }
    wg_barrier();
}

// One row: exact radix SELECT of the k-th smallest canonical key, tie-aware collection, sort, output (whole workgroup).
// lds: P * 8 + (TK_NB + TK_WAVES + 1 + 4) * 4 bytes.
__device__ __forceinline__ void topk_select_row(const float *__restrict__ drow, int N, int64_t col_offset, int k, int P,
                                                float *__restrict__ od, int32_t *__restrict__ oi, uint64_t *lds)
{
    uint64_t *cand = lds;                               // [P]
    uint32_t *hist = (uint32_t *)(lds + P);             // [TK_NB]
    uint32_t *wcnt = hist + TK_NB;                      // [TK_WAVES + 1]
    uint32_t *ctl = wcnt + TK_WAVES + 1;                // [4]: prefix, remaining k, n_lt cursor
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int chunk = (N + TK_WAVES - 1) / TK_WAVES;
    chunk = (chunk + WAVE - 1) / WAVE * WAVE;
    const int beg = wave * chunk;
    const int end = (beg + chunk < N) ? (beg + chunk) : N;
    {
        // ---------- radix select: find the canonical key of the k-th smallest ----------
        uint32_t prefix = 0, prefix_mask = 0;
        uint32_t remaining = (uint32_t)k;  // rank (1-based) of the wanted key among keys matching prefix
        for (int pass = 0; pass < 3; pass++) {
            const int shift = (pass == 0) ? 21 : (pass == 1 ? 10 : 0);
            const int nb = (pass == 2) ? 10 : 11;
            const uint32_t dmask = (1u << nb) - 1u;
            wg_barrier();
            for (int i = tid; i < TK_NB; i += TK_THREADS) hist[i] = 0;
            wg_barrier();
            for (int i = tid; i < N; i += TK_THREADS) {
                const uint32_t key = canon_key(drow[i]);
                if ((key & prefix_mask) == prefix) atomicAdd(&hist[(key >> shift) & dmask], 1u);
            }
            wg_barrier();
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
            wg_barrier();
            prefix |= ctl[0] << shift;
            prefix_mask |= dmask << shift;
            remaining = ctl[1];
        }
        const uint32_t kth = prefix;          // exact key of the k-th smallest element
        const uint32_t need_ties = remaining; // how many elements equal to kth belong to the top-k
        const uint32_t n_lt = (uint32_t)k - need_ties;

        // ---------- collect: keys < kth (any order) + the `need_ties` lowest-index ties ----------
        wg_barrier();
        if (tid == 0) ctl[2] = 0;
        // per-wave tie counts (index order matters for ties)
        uint32_t myties = 0;
        for (int i = beg + lane; i < end; i += WAVE) myties += (canon_key(drow[i]) == kth) ? 1u : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) myties += __shfl_xor(myties, off, 64);
        if (lane == 0) wcnt[wave] = myties;
        for (int i = tid; i < P; i += TK_THREADS) cand[i] = ~0ull;
        wg_barrier();
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
        wg_barrier();

        // ---------- sort the k survivors on (key, index) and emit ----------
        bitonic_sort_u64(cand, P, TK_THREADS);
        for (int r = tid; r < k; r += TK_THREADS) {
            const uint64_t c = cand[r];
            od[r] = key_to_float((uint32_t)(c >> 32));
            oi[r] = (int32_t)(col_offset + (int64_t)(uint32_t)c);
        }
    }
}

__global__ __launch_bounds__(TK_THREADS) void topk_rows_kernel(const float *__restrict__ pdist, int64_t ldp,
                                                               int64_t Q, int N, int64_t col_offset, int k, int P,
                                                               float *__restrict__ out_d, int32_t *__restrict__ out_i, int only_flagged)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t tk_lds64[];
    for (int64_t row = blockIdx.x; row < Q; row += gridDim.x) {
        if (only_flagged && out_i[row * k] != TK_REDO) continue;   // repair pass behind topk_sample_kernel (uniform per workgroup)
        topk_select_row(pdist + row * ldp, N, col_offset, k, P, out_d + row * k, out_i + row * k, tk_lds64);
    }
}

// ------------------------------------------------------------------------------------------------
// Register-blocked bitonic sort of P = TK_THREADS * PER values held PER per thread (value e = tid * PER + r):
// partners closer than PER live in the same thread, closer than 64 * PER in the same wave (shuffle, no barrier);
// only the log2(P / (64 PER))-deep far strides go through LDS.  A 2048-value sort needs 6 barrier stages
// instead of the 66 of the all-LDS version above.
template <typename T>
__device__ __forceinline__ T shfl_xor_any(T v, int mask)
{
    if constexpr (sizeof(T) == 8) {
        const uint32_t lo = (uint32_t)v, hi = (uint32_t)((uint64_t)v >> 32);
        return (T)(((uint64_t)(uint32_t)__shfl_xor((int)hi, mask, 64) << 32) | (uint32_t)__shfl_xor((int)lo, mask, 64));
    } else {
        return (T)__shfl_xor((int)v, mask, 64);
    }
}

template <typename T, int PER>
__device__ __forceinline__ void blocked_bitonic_sort(T (&v)[PER], T *lds)
{
    constexpr int P = TK_THREADS * PER;
    const int tid = threadIdx.x;
#define TK_CE_REMOTE(O, R)                                                     \
    {                                                                          \
        const int e = tid * PER + (R);                                         \
        const bool take_min = (((e & k) == 0) == ((e & j) == 0));              \
        v[R] = take_min ? ((O) < v[R] ? (O) : v[R]) : ((O) > v[R] ? (O) : v[R]); \
    }
#pragma unroll 1
    for (int k = 2; k <= P; k <<= 1) {
        int j = k >> 1;
#pragma unroll 1
        for (; j >= 64 * PER; j >>= 1) {               // partner in another wave: through LDS
            wg_barrier();
#pragma unroll
            for (int r = 0; r < PER; r++) lds[tid * PER + r] = v[r];
            wg_barrier();
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const T o = lds[(tid * PER + r) ^ j];
                TK_CE_REMOTE(o, r)
            }
        }
#pragma unroll 1
        for (; j >= PER; j >>= 1) {                    // partner thread in the same wave: shuffle, no barrier
            const int lm = j / PER;
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const T o = shfl_xor_any(v[r], lm);
                TK_CE_REMOTE(o, r)
            }
        }
#pragma unroll
        for (int jj = PER >> 1; jj > 0; jj >>= 1) {    // partner register of the same thread (static indices)
            if (jj <= j) {
#pragma unroll
                for (int r = 0; r < PER; r++) {
                    if ((r & jj) == 0) {
                        const bool up = (((tid * PER + r) & k) == 0);
                        const T a = v[r], b = v[r | jj];
                        const bool sw = (a > b) == up;
                        v[r] = sw ? b : a;
                        v[r | jj] = sw ? a : b;
                    }
                }
            }
        }
    }
#undef TK_CE_REMOTE
}

// ------------------------------------------------------------------------------------------------
// Sample-select variant (the fast path for long rows): the k smallest of N keys with ONE streaming read of
// the row and almost no atomics.
//   1. 2048 evenly spaced keys of the row are sorted in LDS; the sample at rank r(k, N) -- chosen 4 sigma above
//      the k/N quantile -- is the pivot, so that   k <= #{key <= pivot} <= TK_CAP   with overwhelming probability;
//   2. one coalesced pass over the row collects every (key, index) with key <= pivot (about 2 % of the row takes
//      the one LDS atomic; the radix select above does 3 atomics per key on a handful of hot histogram bins);
//   3. the candidates are sorted on the 64-bit (key, index) composite -- exactly the canonical order -- and the
//      first k are written out.
// A row whose candidate count misses [k, TK_CAP] (a huge tie group straddling rank k, or bad luck) is flagged
// with out_i[row, 0] = TK_REDO and redone by the exact radix-select kernel launched right behind this one.
constexpr int TK_SAMPLES = 2048;
constexpr int TK_CAP = 8192;

__device__ __forceinline__ void bitonic_sort_u32(uint32_t *v, int P, int nthreads)
{
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            wg_barrier();
            for (int t = threadIdx.x; t < P / 2; t += nthreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = ((i & k) == 0);
                const uint32_t a = v[i], b = v[l];
                if ((a > b) == up) { v[i] = b; v[l] = a; }
            }
        }
    }
    wg_barrier();
}

__global__ __launch_bounds__(TK_THREADS, 4) void topk_sample_kernel(const float *__restrict__ pdist, int64_t ldp, int64_t Q, int N,
                                                                 int64_t col_offset, int k, int pivot_rank,
                                                                 float *__restrict__ out_d, int32_t *__restrict__ out_i)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t ts_lds64[];
    uint64_t *cand = ts_lds64;                                  // [TK_CAP]
    uint32_t *smp = (uint32_t *)(ts_lds64 + TK_CAP);            // [TK_SAMPLES]
    uint32_t *ctl = smp + TK_SAMPLES;                           // [2]
    const int tid = threadIdx.x;
    for (int64_t row = blockIdx.x; row < Q; row += gridDim.x) {
        const float *drow = pdist + row * ldp;
        wg_barrier();
        {   // 2048 evenly spaced sample keys, 4 per thread, sorted in registers (+ 2 LDS stages); rank r is the pivot
            uint32_t sv[TK_SAMPLES / TK_THREADS];
#pragma unroll
            for (int r = 0; r < TK_SAMPLES / TK_THREADS; r++)
                sv[r] = canon_key(drow[(int64_t)(tid * (TK_SAMPLES / TK_THREADS) + r) * N / TK_SAMPLES]);
            blocked_bitonic_sort<uint32_t, TK_SAMPLES / TK_THREADS>(sv, smp);
            wg_barrier();
#pragma unroll
            for (int r = 0; r < TK_SAMPLES / TK_THREADS; r++) smp[tid * (TK_SAMPLES / TK_THREADS) + r] = sv[r];
            if (tid == 0) ctl[0] = 0;
            wg_barrier();
        }
        const uint32_t pivot = smp[pivot_rank];
        // ---- one pass: collect (key, index) with key <= pivot ----
        // (16 coalesced loads in flight per thread: with 4 the pass is HBM-latency-bound, 25 round trips per row)
        constexpr int TK_PF = 16;
        for (int i0 = 0; i0 < N; i0 += TK_THREADS * TK_PF) {
            uint32_t key[TK_PF];
#pragma unroll
            for (int e = 0; e < TK_PF; e++) {
                const int i = i0 + e * TK_THREADS + tid;
                key[e] = canon_key(drow[i < N ? i : N - 1]);
            }
#pragma unroll
            for (int e = 0; e < TK_PF; e++) {
                const int i = i0 + e * TK_THREADS + tid;
                if (i < N && key[e] <= pivot) {
                    const uint32_t slot = atomicAdd(&ctl[0], 1u);
                    if (slot < TK_CAP) cand[slot] = ((uint64_t)key[e] << 32) | (uint32_t)i;
                }
            }
        }
        wg_barrier();
        const uint32_t total = ctl[0];
        if (total < (uint32_t)k || total > (uint32_t)TK_CAP) {      // uniform: hand the row to the exact kernel
            if (tid == 0) out_i[row * k] = TK_REDO;
            continue;
        }
        // sort the candidates on (key, index): registers + shuffles, padded with ~0 to TK_THREADS * PER
#define TK_SORT_CAND(PER)                                                                     \
    {                                                                                         \
        uint64_t cv[PER];                                                                     \
        _Pragma("unroll") for (int r = 0; r < PER; r++) {                                     \
            const int e = tid * PER + r;                                                      \
            cv[r] = (e < (int)total) ? cand[e] : ~0ull;                                       \
        }                                                                                     \
        blocked_bitonic_sort<uint64_t, PER>(cv, cand);                                        \
        wg_barrier();                                                                      \
        _Pragma("unroll") for (int r = 0; r < PER; r++) cand[tid * PER + r] = cv[r];          \
        wg_barrier();                                                                      \
    }
        if (total <= TK_THREADS) TK_SORT_CAND(1)
        else if (total <= 2 * TK_THREADS) TK_SORT_CAND(2)
        else if (total <= 4 * TK_THREADS) TK_SORT_CAND(4)
        else if (total <= 8 * TK_THREADS) TK_SORT_CAND(8)
        else TK_SORT_CAND(16)
#undef TK_SORT_CAND
        for (int r = tid; r < k; r += TK_THREADS) {
            const uint64_t c = cand[r];
            out_d[row * k + r] = key_to_float((uint32_t)(c >> 32));
            out_i[row * k + r] = (int32_t)(col_offset + (int64_t)(uint32_t)c);
        }
    }
}

__global__ __launch_bounds__(256) void topk_merge_kernel(const float *__restrict__ d, const int32_t *__restrict__ idx, int64_t part_stride,
                                                         int parts, int64_t Q, int k, int P,
                                                         float *__restrict__ out_d, int32_t *__restrict__ out_i)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t mg_lds[];
    const int m = parts * k;
    for (int64_t row = blockIdx.x; row < Q; row += gridDim.x) {
        wg_barrier();
        for (int t = threadIdx.x; t < P; t += 256) {
            uint64_t v = ~0ull;
            if (t < m) {
                const int p = t / k, r = t - p * k;
                const int64_t src = (int64_t)p * part_stride + row * k + r;
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
    hipStream_t s = (hipStream_t)stream;
    // sample-select fast path: pivot = sample of rank r, r - 4 sqrt(r) >= m = k S / N  (r = (2 + sqrt(4 + m))^2 + 4);
    // taken when the 4-sigma upper bound of the candidate count fits the LDS candidate buffer
    static const bool exact_only = tuning_env("SE_TOPK_EXACT") != nullptr;   // -DSE_TUNING build only: exact radix select for every row
    const double m = (double)k * TK_SAMPLES / (double)n;
    const double rr = (2.0 + sqrt(4.0 + m)) * (2.0 + sqrt(4.0 + m)) + 4.0;
    const int r = (int)ceil(rr);
    const double upper = (rr + 4.0 * sqrt(rr) + 8.0) / TK_SAMPLES * (double)n;
    int only_flagged = 0;
    if (!exact_only && n >= 4 * TK_SAMPLES && r < TK_SAMPLES && upper <= (double)TK_CAP) {
        const size_t lds2 = (size_t)TK_CAP * 8 + (TK_SAMPLES + 4) * sizeof(uint32_t);
        SE_HIP_CHECK(hipFuncSetAttribute((const void *)topk_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        const int64_t grid2 = q < 1024 ? q : 1024;
        hipLaunchKernelGGL(topk_sample_kernel, dim3((unsigned)grid2), dim3(TK_THREADS), lds2, s, pdist, ldp, q, (int)n, col_offset, k, r, out_d, out_i);
        SE_LAUNCH_CHECK();
        only_flagged = 1;
    }
    hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)grid), dim3(TK_THREADS), lds, s,
                       pdist, ldp, q, (int)n, col_offset, k, P, out_d, out_i, only_flagged);
    SE_LAUNCH_CHECK();
    return SE_OK;
}


// ------------------------------------------------------------------------------------------------
// se_retrieve_topk: distances + top-k without the distance matrix (SURVEY.md section 8d "fused top-k": bytes =
// 4 (Q + N) D + 8 Q k).  Replaces the head of evaluate_retrieval.py:57-67 for consumers that only need the first k
// entries of every ranking (clipped AHP / P@k without AP; the per-shard step of the sharded-gallery split).
//
// Three steps on the tile loop of pdist_mfma.hip (gallery = row operand, queries = column operand: a lane owns one query
// per 32-column block), no [Q, N] slab in between:
//   1. SAMPLE pass (EPI_GROUPMIN): S evenly spaced gallery rows x all queries; every lane writes the minimum of the 16
//      values it holds of its query -> gm[query, S / 16 group minima].
//   2. THRESHOLD: tau[query] = the j-th smallest group minimum (one wave per query, bitonic sort in registers).  A
//      group minimum is <= x as soon as ONE of its 16 samples is, so the number of groups below the k/N quantile is
//      Binomial(G, 1 - (1 - k/N)^16): j is chosen on the host so that fewer than k gallery items lie below tau with
//      probability <= 1e-6 per query, and the list capacity so that more than `cap` do with the same probability.
//   3. MAIN pass (EPI_FILTER): all N gallery rows; values <= tau[query] (about 2-3 k of the N) are appended to the
//      query's candidate list (one returning atomic per lane and tile for the slots), then a per-query kernel sorts the
//      candidates on the canonical (key, index) composite and writes the first k.
// Every key of the true top-k is <= the k-th smallest key <= tau whenever the list holds >= k entries, so the result is
// EXACTLY the head of the canonical ranking; queries whose list holds < k or > cap entries (NaN rows, huge tie groups, a
// gallery the strided sample misrepresents) are redone by an exact kernel that recomputes their distance row with the same
// FMA chain on the vector ALU (bit-identical to the MFMA chain) and runs the radix select on it.
// Small problems (n < 16384, or k/N too large for the sample to resolve) keep the distance-slab path below.
// ------------------------------------------------------------------------------------------------

namespace se {

constexpr int FB_GRID = 256;                    // workgroups (and scratch rows) of the exact fallback kernel
constexpr double FUSED_P_FAIL = 1e-6;           // per-query probability of needing it, by design

static double binom_sf(int G, double p, int j)  // P(Binomial(G, p) >= j)
{
    if (j <= 0) return 1.0;
    if (j > G) return 0.0;
    if (p <= 0.0) return 0.0;
    if (p >= 1.0) return 1.0;
    double sum = 0.0;
    const double lp = log(p), lq = log1p(-p);
    for (int x = j; x <= G; x++) sum += exp(lgamma(G + 1.0) - lgamma(x + 1.0) - lgamma(G - x + 1.0) + x * lp + (G - x) * lq);
    return sum < 1.0 ? sum : 1.0;
}

struct FusedPlan {
    bool ok;
    int S;          // sampled gallery rows (multiple of 128)
    int64_t step;   // sample i = gallery row i * step
    int G;          // group minima per query = S / 16
    int j;          // tau = j-th smallest group minimum (1-based)
    int cap;        // candidate list capacity per query
};

static FusedPlan fused_plan_compute(int64_t n, int64_t ldg, int k)
{
    FusedPlan p = {false, 0, 0, 0, 0, 0};
    int force = -1;                                                          // -DSE_TUNING build: SE_TOPK_FUSED=0 / 1 pins the path
    if (const char *e = tuning_env("SE_TOPK_FUSED")) force = atoi(e);
    if (force == 0) return p;
    if (n < (force == 1 ? 256 : 16384)) return p;
    int S = n >= 65536 ? 4096 : 2048;
    if ((int64_t)S > n / 2) S = (int)(n / 2 / 128 * 128);
    if (const char *e = tuning_env("SE_TOPK_SAMPLES")) S = atoi(e) / 128 * 128;
    if (S < 128) return p;
    int64_t step = n / S;
    if (step * ldg >= pdist_max_ld()) step = (pdist_max_ld() - 1) / ldg;    // 32-bit tile offsets: sample a prefix of the gallery more densely
    if (step < 1) return p;
    const int G = S / FUSED_GROUP;
    if (G > 256) return p;                                                   // threshold kernel: 4 minima per lane
    const double q0 = (double)k / (double)n;
    const double pi0 = 1.0 - pow(1.0 - q0, (double)FUSED_GROUP);
    int j = 1;
    while (j <= G && binom_sf(G, pi0, j) > FUSED_P_FAIL) j++;
    const bool forced = force == 1;                                          // tests: small galleries through the fused kernels anyway
    if (j > G / 2) {                                                         // k / n too large for this sample to resolve
        if (!forced) return p;
        if (j > G) j = G;
    }
    // largest plausible fraction of the gallery below tau: the j-th group minimum lies above the q1 quantile with probability
    // P(Binomial(G, pi(q1)) < j); grow q1 until that is <= FUSED_P_FAIL
    double q1 = q0;
    for (int it = 0; it < 400 && q1 < 1.0; it++) {
        const double pi1 = 1.0 - pow(1.0 - q1, (double)FUSED_GROUP);
        if (1.0 - binom_sf(G, pi1, j) <= FUSED_P_FAIL) break;
        q1 *= 1.05;
    }
    if (q1 > 0.25 && !forced) return p;
    if (q1 > 1.0) q1 = 1.0;
    double capd = q1 * (double)n + 6.0 * sqrt(q1 * (double)n) + 64.0;
    if (capd < 2.0 * k) capd = 2.0 * k;
    if (capd > (double)n) capd = (double)n;
    int cap = (int)((capd + 255.0) / 256.0) * 256;
    if (forced && cap > TK_CAP) cap = TK_CAP;
    if (const char *e = tuning_env("SE_TOPK_J")) j = atoi(e);                // tests: a j of 1 sends every query to the exact fallback
    if (const char *e = tuning_env("SE_TOPK_CAP")) cap = atoi(e);
    if (cap > TK_CAP || j < 1 || j > G) return p;
    p = {true, S, step, G, j, cap};
    return p;
}

// The plan depends on (n, ldg, k) only; its binomial-tail loops cost ~0.8 ms of host time and a Python-level call used to run them
// three times before the first kernel was enqueued.  Product build: a small table under a mutex (the tuning build re-derives it every
// time -- its environment switches may change between calls).
static FusedPlan fused_plan(int64_t n, int64_t ldg, int k)
{
    if (kTuning) return fused_plan_compute(n, ldg, k);
    struct Entry { int64_t n, ldg; int k; FusedPlan p; };
    static std::mutex mu;
    static std::vector<Entry> *table = new std::vector<Entry>();      // guarded by mu (never destroyed: no static-destruction order to get wrong)
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const Entry &e : *table)
            if (e.n == n && e.ldg == ldg && e.k == k) return e.p;
    }
    const FusedPlan p = fused_plan_compute(n, ldg, k);
    std::lock_guard<std::mutex> lock(mu);
    if (table->size() >= 64) table->erase(table->begin());
    table->push_back({n, ldg, k, p});
    return p;
}

// Bitonic sort of 64 * PER values held PER per lane by ONE wave (value e = lane * PER + r): partner registers of the same lane,
// then lanes via shuffles -- no LDS, no barrier.
template <typename T, int PER>
__device__ __forceinline__ void wave_bitonic_sort(T (&v)[PER], int lane)
{
    constexpr int P = 64 * PER;
#pragma unroll 1
    for (int k = 2; k <= P; k <<= 1) {
#pragma unroll 1
        for (int jj = k >> 1; jj >= PER; jj >>= 1) {          // partner in another lane
            const int lm = jj / PER;
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const T o = shfl_xor_any(v[r], lm);
                const int e = lane * PER + r;
                const bool take_min = (((e & k) == 0) == ((e & jj) == 0));
                v[r] = take_min ? (o < v[r] ? o : v[r]) : (o > v[r] ? o : v[r]);
            }
        }
#pragma unroll
        for (int jj = PER >> 1; jj > 0; jj >>= 1) {            // partner register of the same lane (static indices)
            if (jj < k) {
#pragma unroll
                for (int r = 0; r < PER; r++) {
                    if ((r & jj) == 0) {
                        const bool up = (((lane * PER + r) & k) == 0);
                        const T a = v[r], b = v[r | jj];
                        const bool sw = (a > b) == up;
                        v[r] = sw ? b : a;
                        v[r | jj] = sw ? a : b;
                    }
                }
            }
        }
    }
}

// The last phase of that sort on its own: a BITONIC sequence of 64 * PER values (blocked layout) -> ascending order.  log2(64 * PER) stages.
template <typename T, int PER>
__device__ __forceinline__ void wave_bitonic_merge(T (&v)[PER], int lane)
{
#pragma unroll
    for (int jj = 32 * PER; jj >= PER; jj >>= 1) {            // partner in another lane
        const int lm = jj / PER;
        const bool take_min = (lane & lm) == 0;
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const T o = shfl_xor_any(v[r], lm);
            v[r] = take_min ? (o < v[r] ? o : v[r]) : (o > v[r] ? o : v[r]);
        }
    }
#pragma unroll
    for (int jj = PER >> 1; jj > 0; jj >>= 1) {                // partner register of the same lane (static indices)
#pragma unroll
        for (int r = 0; r < PER; r++) {
            if ((r & jj) == 0) {
                const T a = v[r], b = v[r | jj];
                const bool sw = a > b;
                v[r] = sw ? b : a;
                v[r | jj] = sw ? a : b;
            }
        }
    }
}

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src)
{
    return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src, 64);
}

// se_topk_merge for k <= 1024: ONE WAVE per query keeps the best 64 * PER >= k entries in registers and folds the parts in one at a
// time -- min(best[e], part[P - 1 - e]) of two ascending lists is a bitonic sequence holding the P smallest of their union, and
// log2(P) compare-exchange stages put it in order again.  No LDS, no barrier.  (The first version sorted all parts * k entries of a
// query in LDS: 66 barrier-separated stages for 8 x 251 -- 2.0 ms for 50,000 queries, 0.45 TB/s.)  Parts are expected ascending under
// the canonical (distance, index) order, as se_retrieve_topk / se_topk_rows write them; a part that is not is sorted first.
template <int PER>
__global__ __launch_bounds__(256) void topk_merge_wave_kernel(const float *__restrict__ d, const int32_t *__restrict__ idx, int64_t part_stride,
                                                              int parts, int64_t Q, int k, float *__restrict__ out_d, int32_t *__restrict__ out_i)
{
    const int lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < Q; row += (int64_t)gridDim.x * 4) {
        uint64_t v[PER];
        for (int p = 0; p < parts; p++) {
            const int64_t base = (int64_t)p * part_stride + row * k;
            uint64_t w[PER];
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const int e = lane * PER + r;
                w[r] = e < k ? (((uint64_t)canon_key(d[base + e]) << 32) | (uint32_t)idx[base + e]) : ~0ull;
            }
            bool ok = true;
#pragma unroll
            for (int r = 0; r + 1 < PER; r++) ok = ok && w[r] <= w[r + 1];
            const uint64_t nxt = shfl_u64(w[0], lane < 63 ? lane + 1 : 63);
            ok = ok && (lane == 63 || w[PER - 1] <= nxt);
            if (__ballot(!ok) != 0ull) wave_bitonic_sort<uint64_t, PER>(w, lane);      // (wave-uniform; not expected)
            if (p == 0) {
#pragma unroll
                for (int r = 0; r < PER; r++) v[r] = w[r];
                continue;
            }
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const uint64_t o = shfl_u64(w[PER - 1 - r], 63 - lane);                 // the part, descending
                v[r] = o < v[r] ? o : v[r];
            }
            wave_bitonic_merge<uint64_t, PER>(v, lane);
        }
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int e = lane * PER + r;
            if (e < k) {
                out_d[row * k + e] = key_to_float((uint32_t)(v[r] >> 32));
                out_i[row * k + e] = (int32_t)(uint32_t)v[r];
            }
        }
    }
}

// tau[q] = j-th smallest of the G group minima of query q: one wave per query, 4 keys per lane, bitonic sort in registers.
__global__ __launch_bounds__(256) void topk_tau_kernel(const float *__restrict__ gm, int64_t gm_ld, int64_t Q, int G, int j, float *__restrict__ tau)
{
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    uint32_t v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int e = lane * 4 + r;
        v[r] = e < G ? canon_key(gm[q * gm_ld + e]) : 0xFFFFFFFFu;
    }
    wave_bitonic_sort<uint32_t, 4>(v, lane);
    const int want = j - 1;
    if (lane == (want >> 2)) {
        const uint32_t key = (want & 3) == 0 ? v[0] : ((want & 3) == 1 ? v[1] : ((want & 3) == 2 ? v[2] : v[3]));
        tau[q] = key == 0xFFFFFFFFu ? __builtin_nanf("") : key_to_float(key);   // NaN: nothing passes, the query goes to the fallback
    }
}

// ------------------------------------------------------------------------------------------------
// Candidate lists -> top-k, one WAVE per query (k <= 256, lists of <= 2048 candidates: the sizes the fused path produces):
//   1. the candidates sit PER per lane in registers;
//   2. radix SELECT of the k-th smallest key: 4 passes of 8 bits, a 256-bin histogram in wave-private LDS (one LDS add per key
//      and pass), wave scan of the bins;
//   3. every candidate with key <= that key (k plus the ties of the k-th key: a few hundred of ~760) is compacted into LDS,
//   4. sorted on the canonical 64-bit (key, index) composite by an in-register wave bitonic sort, and the first k are written.
// ~1.5k wave instructions per query instead of the ~13k of a full sort of the list.  Queries it cannot finish (k > 256, huge tie
// groups) are marked TK_WIDE for the workgroup-wide kernel below; lists outside [k, cap] are marked TK_REDO for the exact kernel.
constexpr int32_t TK_WIDE = -2;
constexpr int TL_WAVES = 4;
constexpr int TL_SEL = 512;             // compacted candidates per query the wave sort takes

template <int PER>
__device__ __forceinline__ bool topk_wave_select(const uint2 *__restrict__ lst, int total, int k, int64_t col_offset, uint32_t *hist,
                                                 uint64_t *sel, float *__restrict__ od, int32_t *__restrict__ oi, int lane)
{
    uint32_t key[PER], idx[PER];
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const int e = r * 64 + lane;
        const uint2 c = lst[e < total ? e : 0];
        key[r] = e < total ? canon_key(__uint_as_float(c.x)) : 0xFFFFFFFFu;   // (list entries are never NaN: padding sorts behind everything)
        idx[r] = e < total ? c.y : 0xFFFFFFFFu;
    }
    // ---- k-th smallest key ----
    uint32_t prefix = 0, pmask = 0, remaining = (uint32_t)k, n_eq = 0;
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8) {
#pragma unroll
        for (int i = 0; i < 4; i++) hist[lane * 4 + i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int r = 0; r < PER; r++)
            if ((key[r] & pmask) == prefix && r * 64 + lane < total) atomicAdd(&hist[(key[r] >> shift) & 255u], 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t c[4], local = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { c[i] = hist[lane * 4 + i]; local += c[i]; }
        uint32_t incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        uint32_t run = incl - local, digit = 0, rem = 0, cnt = 0;
        const bool mine = remaining > run && remaining <= incl;             // exactly one lane
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool hit = mine && remaining > run && remaining <= run + c[i];
            digit = hit ? (uint32_t)(lane * 4 + i) : digit;
            rem = hit ? remaining - run : rem;
            cnt = hit ? c[i] : cnt;
            run += c[i];
        }
        const int src = __ffsll((long long)__ballot(mine)) - 1;
        digit = (uint32_t)__shfl((int)digit, src, 64);
        remaining = (uint32_t)__shfl((int)rem, src, 64);
        n_eq = (uint32_t)__shfl((int)cnt, src, 64);
        prefix |= digit << shift;
        pmask |= 255u << shift;
    }
    const uint32_t kth = prefix;
    const uint32_t n_le = (uint32_t)k - remaining + n_eq;                   // keys below the k-th key + ALL its ties
    if (n_le > (uint32_t)TL_SEL) return false;
    // ---- compact the candidates with key <= kth into LDS, sort them on (key, index), write the first k ----
    uint32_t base = 0;
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const bool take = key[r] <= kth && r * 64 + lane < total;
        const uint64_t m = __ballot(take);
        if (take) sel[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = ((uint64_t)key[r] << 32) | idx[r];
        base += (uint32_t)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#define TL_SORT_OUT(P2)                                                                        \
    {                                                                                          \
        uint64_t sv[P2];                                                                       \
        _Pragma("unroll") for (int r = 0; r < P2; r++) {                                       \
            const int e = lane * P2 + r;                                                       \
            sv[r] = e < (int)n_le ? sel[e] : ~0ull;                                            \
        }                                                                                      \
        wave_bitonic_sort<uint64_t, P2>(sv, lane);                                             \
        _Pragma("unroll") for (int r = 0; r < P2; r++) {                                       \
            const int e = lane * P2 + r;                                                       \
            if (e < k) {                                                                       \
                od[e] = key_to_float((uint32_t)(sv[r] >> 32));                                 \
                oi[e] = (int32_t)(col_offset + (int64_t)(uint32_t)sv[r]);                      \
            }                                                                                  \
        }                                                                                      \
    }
    if (n_le <= 64) TL_SORT_OUT(1)
    else if (n_le <= 128) TL_SORT_OUT(2)
    else if (n_le <= 256) TL_SORT_OUT(4)
    else TL_SORT_OUT(8)
#undef TL_SORT_OUT
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                  // sel / hist are reused by this wave's next query
    return true;
}

__global__ __launch_bounds__(TL_WAVES * 64) void topk_lists_wave_kernel(const uint2 *__restrict__ lists, const unsigned *__restrict__ rowcnt, int64_t cap,
                                                                       int64_t Q, int64_t col_offset, int k, float *__restrict__ out_d,
                                                                       int32_t *__restrict__ out_i, unsigned *__restrict__ nflag)
{
    __shared__ uint32_t hist_all[TL_WAVES][256];
    __shared__ uint64_t sel_all[TL_WAVES][TL_SEL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *hist = hist_all[wave];
    uint64_t *sel = sel_all[wave];
    for (int64_t row = (int64_t)blockIdx.x * TL_WAVES + wave; row < Q; row += (int64_t)gridDim.x * TL_WAVES) {
        const unsigned total = rowcnt[row];
        if (total < (unsigned)k || total > (unsigned)cap) {
            if (lane == 0) { out_i[row * k] = TK_REDO; atomicAdd(&nflag[1], 1u); }
            continue;
        }
        const uint2 *lst = lists + row * cap;
        bool done = false;
        if (k <= 256 && total <= 2048u) {
            if (total <= 512u) done = topk_wave_select<8>(lst, (int)total, k, col_offset, hist, sel, out_d + row * k, out_i + row * k, lane);
            else if (total <= 1024u) done = topk_wave_select<16>(lst, (int)total, k, col_offset, hist, sel, out_d + row * k, out_i + row * k, lane);
            else done = topk_wave_select<32>(lst, (int)total, k, col_offset, hist, sel, out_d + row * k, out_i + row * k, lane);
        }
        if (!done && lane == 0) { out_i[row * k] = TK_WIDE; atomicAdd(&nflag[0], 1u); }
    }
}

// candidates of every query -> canonical top-k; queries whose list missed [k, cap] are flagged for the exact kernel
__global__ __launch_bounds__(TK_THREADS) void topk_lists_kernel(const uint2 *__restrict__ lists, const unsigned *__restrict__ rowcnt, int64_t cap,
                                                                int64_t Q, int64_t col_offset, int k, float *__restrict__ out_d,
                                                                int32_t *__restrict__ out_i, int only_wide, unsigned *__restrict__ nflag)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t tl_lds64[];
    uint64_t *cand = tl_lds64;
    const int tid = threadIdx.x;
    if (only_wide && nflag[0] == 0) return;                        // nothing was handed over: the usual case
    for (int64_t row = blockIdx.x; row < Q; row += gridDim.x) {
        if (only_wide && out_i[row * k] != TK_WIDE) continue;      // behind topk_lists_wave_kernel: only the queries it handed over (uniform)
        const unsigned total = rowcnt[row];
        if (total < (unsigned)k || total > (unsigned)cap) {
            if (tid == 0) { out_i[row * k] = TK_REDO; atomicAdd(&nflag[1], 1u); }
            continue;
        }
        const uint2 *lst = lists + row * cap;
        wg_barrier();     // the previous row's output reads of cand are done
#define TK_SORT_LIST(PER)                                                                     \
    {                                                                                         \
        uint64_t cv[PER];                                                                     \
        _Pragma("unroll") for (int r = 0; r < PER; r++) {                                     \
            const int e = tid * PER + r;                                                      \
            cv[r] = ~0ull;                                                                    \
            if (e < (int)total) { const uint2 c = lst[e]; cv[r] = ((uint64_t)canon_key(__uint_as_float(c.x)) << 32) | c.y; } \
        }                                                                                     \
        blocked_bitonic_sort<uint64_t, PER>(cv, cand);                                        \
        wg_barrier();                                                                      \
        _Pragma("unroll") for (int r = 0; r < PER; r++) if (tid * PER + r < k) cand[tid * PER + r] = cv[r]; \
        wg_barrier();                                                                      \
    }
        if (total <= TK_THREADS) TK_SORT_LIST(1)
        else if (total <= 2 * TK_THREADS) TK_SORT_LIST(2)
        else if (total <= 4 * TK_THREADS) TK_SORT_LIST(4)
        else if (total <= 8 * TK_THREADS) TK_SORT_LIST(8)
        else TK_SORT_LIST(16)
#undef TK_SORT_LIST
        for (int r = tid; r < k; r += TK_THREADS) {
            const uint64_t c = cand[r];
            out_d[row * k + r] = key_to_float((uint32_t)(c >> 32));
            out_i[row * k + r] = (int32_t)(col_offset + (int64_t)(uint32_t)c);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 pre-filter path of se_retrieve_topk (prefilter.hip holds the tile kernel).  Notation: d = the canonical fp32 distance (what
// the caller gets), d~ = the distance the bf16 matrix-core pass computes, eps(q) >= |d~ - d| for every gallery item (pf_thr_kernel).
//   1. sample pass (bf16) -> tau~(q) = j-th smallest group minimum of d~;  thr(q) = tau~ + 2 eps: every item with d <= tau~ + eps --
//      a superset of what the exact-arithmetic threshold of the same sample would admit -- has d~ <= thr.
//   2. filter pass (bf16): items with d~ <= thr(q) (or d~ NaN: irregular rows) -> candidate list (d~, gallery row).
//   3. pf_refine_kernel, one wave per query:
//        kth~ = k-th smallest d~ of the list;  B = min(kth~ + 2 eps, thr);  R = {d~ <= B} + {d~ NaN}      (|R| ~ k + a 2 eps window)
//        exact fp32 FMA chain (K-blocks included) for every item of R, sort on (key, index), d_k = k-th smallest exact distance
//        accept iff d_k <= B - eps: an item outside R has d~ > B, hence d > B - eps >= d_k -- it is not among the k nearest, ties
//        included.  (With B = kth~ + 2 eps this always holds: the k items with d~ <= kth~ have d <= kth~ + eps.)
//      Everything else -- short / overflowing lists, |R| > RF_MAX, B cut by thr -- is flagged and redone by topk_fallback_kernel.
// The output is the head of the canonical ranking, bit for bit, whatever the filter computed: it only decides where to look.
// ------------------------------------------------------------------------------------------------
constexpr int RF_MAX = 1024;            // exact recomputations per query the refinement takes
constexpr int RF_WAVES = 4;
constexpr int PF_K_MAX = RF_MAX / 2;
#ifndef SE_RF_LPC
#define SE_RF_LPC 4
#endif
constexpr int RF_STAGE_M = 384, RF_STAGE_D = 256;      // lists of up to RF_STAGE_M candidates with rows of RF_STAGE_D columns and more are staged through LDS:
constexpr int RF_LPC = SE_RF_LPC;                       //   lanes per candidate row and load instruction (each 16 bytes)
constexpr int RF_KC = RF_LPC * 4;                       //   columns per chunk
constexpr int RF_CPR = 1024 / RF_KC;                    //   candidates per round (a 4 KB block of rows per chunk)
constexpr int RF_NI = RF_CPR * RF_LPC / 64;             //   load instructions per chunk
constexpr int RF_PITCH = RF_KC + 4;                     //   LDS row pitch in words: conflict-free 16-byte reads by one lane per row
static_assert(RF_NI >= 1 && RF_CPR <= 64 && (RF_MAX - RF_STAGE_M) * 8 >= RF_CPR * RF_PITCH * 4, "the row buffer lives in comp above the results");
constexpr size_t FB_ROWBUF_BYTES = (size_t)TK_WAVES * RF_CPR * RF_PITCH * 4;      // topk_fallback_kernel: one such buffer per wave, behind the select's LDS

// thr / eps per query from the group minima of the sample pass.  hq, rq: norm of the query's bf16 image / of its rounding residual
// (upper bounds, NaN for irregular rows); gctl[0..1]: float bits of the maxima of the same two norms over the gallery.
//   |a.b - a~.b~| = |a~.rb + ra.b~ + ra.rb| <= hq RG + rq HG + rq RG                      (Cauchy-Schwarz on the ACTUAL residuals)
//   matrix-core accumulation: each v_mfma_f32_32x32x16_f16 is assumed to return C + sum of its 16 products with an absolute error
//     <= 2^-20 (|C| + sum |products|) -- IEEE rounding of the exact sum would give 2^-24; measured on gfx950 over adversarial operands
//     (tests/test_gpu_topk.py): 2^-24.4 worst case.  kp / 16 instructions, |C| and the products bounded by hq HG.   [assumption A1, tested on the GPU]
//   the canonical chain itself: |chain - a.b| <= (d + nkb + 2) 2^-24 sum |a_k b_k| <= ... (hq + rq)(HG + RG)
//   fp32 denormal outputs flushed by the matrix core: <= kp 2^-126 (hq + HG + 1) (denormal fp16 INPUTS do not exist: the images flush them)
//   Euclidean epilogue fl(fl(sa + sb) - 2 v): 2 eps_v + 2^-21 ((hq + rq) + (HG + RG))^2
__global__ __launch_bounds__(256) void pf_thr_kernel(const float *__restrict__ gm, int64_t gm_ld, int64_t Q, int G, int j, const float *__restrict__ qn,
                                                     const float *__restrict__ qr, const unsigned *__restrict__ gctl, int metric, int d, int kp,
                                                     int nkb, float *__restrict__ thr, float *__restrict__ eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    uint32_t v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int e = lane * 4 + r;
        v[r] = e < G ? canon_key(gm[q * gm_ld + e]) : 0xFFFFFFFFu;
    }
    wave_bitonic_sort<uint32_t, 4>(v, lane);
    const int want = j - 1;
    if (lane == (want >> 2)) {
        const uint32_t key = (want & 3) == 0 ? v[0] : ((want & 3) == 1 ? v[1] : ((want & 3) == 2 ? v[2] : v[3]));
        const float tau = key == 0xFFFFFFFFu ? __builtin_nanf("") : key_to_float(key);
        const float hq = qn[q], rq = qr[q], HG = __uint_as_float(gctl[0]), RG = __uint_as_float(gctl[1]);
        const float e_round = hq * RG + rq * HG + rq * RG;
        const float e_acc = (float)(kp / 16) * 9.5368e-7f * 1.01f * (hq * HG);
        const float e_chain = (float)(d + nkb + 2) * 5.9605e-8f * 1.01f * ((hq + rq) * (HG + RG));
        const float e_flush = (float)kp * 1.1755e-38f * (hq + HG + 1.0f);
        float e = (e_round + e_acc + e_chain + e_flush) * 1.01f;
        if (metric == SE_METRIC_EUCLID) {
            const float s = (hq + rq) + (HG + RG);
            e = (2.0f * e + 4.7684e-7f * 1.05f * s * s) * 1.01f;
        }
        eps[q] = e;                           // NaN for an irregular query: nothing passes, the query is redone exactly
        thr[q] = tau + 2.05f * e;
    }
}

// exact canonical dot product of gallery row `g` with the (wave-uniform) query row: fmaf chain over k ascending, restarted per K-block,
// block sums added in order -- what the fp32 MFMA tiles and topk_fallback_kernel compute, bit for bit
template <bool VEC>
__device__ __forceinline__ float rf_chain(const float *__restrict__ g, const float *__restrict__ qv, const KBlocks &kbs)
{
    float tot = 0.f;
    int beg = 0;
    for (int kb = 0; kb < kbs.n; kb++) {
        const int end = beg + kbs.len[kb];
        float acc = 0.f;
        int kk = beg;
        if (VEC) {
            for (; kk < end && (kk & 3); kk++) acc = __builtin_fmaf(g[kk], qv[kk], acc);
            for (; kk + 32 <= end; kk += 32) {
                float4 x[8];
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = *(const float4 *)(g + kk + 4 * i);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    acc = __builtin_fmaf(x[i].x, qv[kk + 4 * i], acc);
                    acc = __builtin_fmaf(x[i].y, qv[kk + 4 * i + 1], acc);
                    acc = __builtin_fmaf(x[i].z, qv[kk + 4 * i + 2], acc);
                    acc = __builtin_fmaf(x[i].w, qv[kk + 4 * i + 3], acc);
                }
            }
            for (; kk + 4 <= end; kk += 4) {
                const float4 x = *(const float4 *)(g + kk);
                acc = __builtin_fmaf(x.x, qv[kk], acc);
                acc = __builtin_fmaf(x.y, qv[kk + 1], acc);
                acc = __builtin_fmaf(x.z, qv[kk + 2], acc);
                acc = __builtin_fmaf(x.w, qv[kk + 3], acc);
            }
        }
        for (; kk < end; kk++) acc = __builtin_fmaf(g[kk], qv[kk], acc);
        tot = kb == 0 ? acc : tot + acc;
        beg = end;
    }
    return tot;
}

// The same chains for RF_CPR gallery rows at once, the rows passing through LDS: RF_LPC lanes fetch RF_KC * 4 contiguous bytes of ONE row
// per load instruction (src[i]: this lane's piece of the row that instruction i serves -- row i * 64 / RF_LPC + lane / RF_LPC of the round,
// columns 4 (lane % RF_LPC) ...), two chunks in flight, then lane r < RF_CPR runs row r's chain out of `rowbuf` ([RF_CPR][RF_PITCH] floats,
// private to the wave).  g: the chain lane's own row (the D % 4 last columns).  Needs 16-byte aligned rows and K-block boundaries on
// multiples of 4 (the callers check).  Returns the chain lane's sum.
__device__ __forceinline__ float rf_chain_staged(const float *const (&src)[RF_NI], const float *__restrict__ g, const float *__restrict__ qv,
                                                 const KBlocks &kbs, float *rowbuf, int lane)
{
    float *wr = rowbuf + (lane / RF_LPC) * RF_PITCH + 4 * (lane % RF_LPC);
    const float *rd = rowbuf + (lane < RF_CPR ? lane : 0) * RF_PITCH;
    float tot = 0.f;
    int beg = 0;
    for (int kb = 0; kb < kbs.n; kb++) {
        const int end = beg + kbs.len[kb], end4 = beg + ((end - beg) & ~3);
        float acc = 0.f;
        float4 x0[RF_NI], x1[RF_NI];
        auto issue = [&](float4 (&x)[RF_NI], int kk) {
            const bool mine = kk + 4 * (lane % RF_LPC) < end4;
#pragma unroll
            for (int i = 0; i < RF_NI; i++) x[i] = mine ? *reinterpret_cast<const float4 *>(src[i] + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        auto consume = [&](float4 (&x)[RF_NI], int kk) {
#pragma unroll
            for (int i = 0; i < RF_NI; i++) *reinterpret_cast<float4 *>(wr + (64 / RF_LPC) * i * RF_PITCH) = x[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (kk + RF_KC <= end4) {
#pragma unroll
                for (int j = 0; j < RF_KC; j += 4) {
                    const float4 v4 = *reinterpret_cast<const float4 *>(rd + j);
                    acc = __builtin_fmaf(v4.x, qv[kk + j], acc);
                    acc = __builtin_fmaf(v4.y, qv[kk + j + 1], acc);
                    acc = __builtin_fmaf(v4.z, qv[kk + j + 2], acc);
                    acc = __builtin_fmaf(v4.w, qv[kk + j + 3], acc);
                }
            } else {
                for (int j = 0; kk + j < end4; j += 4) {
                    const float4 v4 = *reinterpret_cast<const float4 *>(rd + j);
                    acc = __builtin_fmaf(v4.x, qv[kk + j], acc);
                    acc = __builtin_fmaf(v4.y, qv[kk + j + 1], acc);
                    acc = __builtin_fmaf(v4.z, qv[kk + j + 2], acc);
                    acc = __builtin_fmaf(v4.w, qv[kk + j + 3], acc);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        };
        int kk = beg;
        if (kk < end4) issue(x0, kk);
        if (kk + RF_KC < end4) issue(x1, kk + RF_KC);
        while (kk < end4) {
            consume(x0, kk);
            if (kk + 2 * RF_KC < end4) issue(x0, kk + 2 * RF_KC);
            kk += RF_KC;
            if (kk >= end4) break;
            consume(x1, kk);
            if (kk + 2 * RF_KC < end4) issue(x1, kk + 2 * RF_KC);
            kk += RF_KC;
        }
        for (kk = end4; kk < end; kk++) acc = __builtin_fmaf(g[kk], qv[kk], acc);     // (last block only: D % 4 columns)
        tot = kb == 0 ? acc : tot + acc;
        beg = end;
    }
    return tot;
}

// Exact path for flagged queries: the query's whole distance row, recomputed with the canonical FMA chain on the vector
// ALU (fmaf over k ascending, restarted per K-block, block sums added in order: what the MFMA tiles compute), into this
// workgroup's scratch row; then the radix select of topk_rows_kernel on it.
template <int METRIC>
__global__ __launch_bounds__(TK_THREADS) void topk_fallback_kernel(const float *__restrict__ queries, int64_t ldq, const float *__restrict__ gallery,
                                                                   int64_t ldg, const float *__restrict__ sqq, const float *__restrict__ sqg,
                                                                   int64_t Q, int N, int D, KBlocks kbs, int64_t col_offset, int k, int P, int sel_bytes,
                                                                   float *__restrict__ scratch, float *__restrict__ out_d,
                                                                   int32_t *__restrict__ out_i, const unsigned *__restrict__ nflag)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t tk_lds64[];
    if (nflag[1] == 0) return;                                     // no query was flagged: the usual case
    float *drow = scratch + (int64_t)blockIdx.x * N;
    bool staged = (reinterpret_cast<uintptr_t>(gallery) & 15) == 0 && (ldg & 3) == 0;       // 16-byte rows, K-block boundaries on multiples of 4
    for (int kb = 0; kb + 1 < kbs.n; kb++) staged = staged && (kbs.len[kb] & 3) == 0;
    // this workgroup's rows blockIdx.x, blockIdx.x + gridDim.x, ...: the flags of 64 of them are read at once (every wave reads the same
    // 64 words: one round trip per 64 rows instead of one per row) and the flagged ones are redone one after the other
    for (int64_t r0 = blockIdx.x; r0 < Q; r0 += (int64_t)gridDim.x * 64) {
      const int64_t myrow = r0 + (int64_t)(threadIdx.x & 63) * gridDim.x;
      uint64_t todo = __ballot(myrow < Q && out_i[myrow * k] == TK_REDO);           // the same mask in every wave
      while (todo) {
        const int64_t row = r0 + (int64_t)(__ffsll((long long)todo) - 1) * gridDim.x;
        todo &= todo - 1;
        const float *qv = queries + row * ldq;
        const float sq_q = METRIC == SE_METRIC_EUCLID ? sqq[row] : 0.f;
        if (staged) {
            // the whole gallery against one query: 64 consecutive rows per wave and round through the wave's LDS buffer (one lane per
            // row fetching its own row 4 bytes at a time kept the CU's address path busy for 2 ms per query at 50,000 x 100)
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            float *rowbuf = reinterpret_cast<float *>(reinterpret_cast<char *>(tk_lds64) + sel_bytes) + wave * (RF_CPR * RF_PITCH);
            for (int c0 = wave * RF_CPR; c0 < N; c0 += TK_WAVES * RF_CPR) {
                const int c = c0 + lane;
                const int cc = (lane < RF_CPR && c < N) ? c : N - 1;
                const float *src[RF_NI];
#pragma unroll
                for (int i = 0; i < RF_NI; i++) {
                    const int cr = c0 + i * (64 / RF_LPC) + lane / RF_LPC;
                    src[i] = gallery + (int64_t)(cr < N ? cr : N - 1) * ldg + 4 * (lane % RF_LPC);
                }
                const float tot = rf_chain_staged(src, gallery + (int64_t)cc * ldg, qv, kbs, rowbuf, lane);
                float v;
                if (METRIC == SE_METRIC_COSINE) v = -tot;
                else v = (sqg[cc] + sq_q) - 2.0f * tot;
                if (lane < RF_CPR && c < N) drow[c] = v;
            }
        } else
        for (int c = threadIdx.x; c < N; c += TK_THREADS) {
            const float *g = gallery + (int64_t)c * ldg;
            float tot = 0.f;
            int beg = 0;
            for (int kb = 0; kb < kbs.n; kb++) {
                float acc = 0.f;
                for (int kk = beg; kk < beg + kbs.len[kb]; kk++) acc = __builtin_fmaf(g[kk], qv[kk], acc);
                tot = kb == 0 ? acc : tot + acc;
                beg += kbs.len[kb];
            }
            float v;
            if (METRIC == SE_METRIC_COSINE) v = -tot;
            else v = (sqg[c] + sq_q) - 2.0f * tot;
            drow[c] = v;
        }
        wg_barrier();     // (global writes of this workgroup are visible to it after the barrier)
        topk_select_row(drow, N, col_offset, k, P, out_d + row * k, out_i + row * k, tk_lds64);
        wg_barrier();
      }
    }
}


#ifdef SE_TUNING
__device__ unsigned long long rf_prof[8];      // tuning build, SE_TOPK_VERBOSE: cycles per phase of pf_refine_kernel, summed over every wave
#define RF_T(i) if (stats) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_acc[i] += now - t_last; t_last = now; }
#else
#define RF_T(i)
#endif

template <int METRIC, bool VEC, bool LONGROWS>
__global__ __launch_bounds__(RF_WAVES * 64, 3) void pf_refine_kernel(const uint2 *__restrict__ lists, const uint2 *__restrict__ spill_lists, int nsub, const unsigned *__restrict__ rowcnt, int64_t cap, int parts, int64_t Q,
                                                                  const float *__restrict__ thr, const float *__restrict__ eps,
                                                                  const float *__restrict__ queries, int64_t ldq, const float *__restrict__ gallery,
                                                                  int64_t ldg, const float *__restrict__ sqq, const float *__restrict__ sqg,
                                                                  KBlocks kbs, int64_t col_offset, int k, float *__restrict__ out_d,
                                                                  int32_t *__restrict__ out_i, unsigned *__restrict__ nflag, unsigned *__restrict__ stats)
{
    __shared__ uint32_t hist_all[RF_WAVES][256];
    __shared__ uint32_t sel_all[RF_WAVES][RF_MAX];
    __shared__ uint64_t comp_all[RF_WAVES][RF_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *hist = hist_all[wave], *sel = sel_all[wave];
    uint64_t *comp = comp_all[wave];
#ifdef SE_TUNING
    uint64_t t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = stats ? __builtin_amdgcn_s_memtime() : 0;
#endif
    for (int64_t row = (int64_t)blockIdx.x * RF_WAVES + wave; row < Q; row += (int64_t)gridDim.x * RF_WAVES) {
        const int64_t urow = __builtin_amdgcn_readfirstlane((int)row);          // (Q < 2^31: the launcher checks)
        RF_T(7)
        // the query's candidates: `parts` sub-lists (one per gallery range of the filter pass) of up to `cap` entries each
        // (sub-lists [0, nsub): the filter pass's own, [nsub, parts): the query's spill region cut into sub-list slots -- a separate array)
        const uint2 *lst_main = lists + urow * nsub * cap, *lst_spill = spill_lists + urow * (parts - nsub) * cap;
        auto sub_list = [&](int p) -> const uint2 * { return p < nsub ? lst_main + (int64_t)p * cap : lst_spill + (int64_t)(p - nsub) * cap; };
        const unsigned *cnts = rowcnt + urow * parts;
        unsigned total = 0;
        bool ok = true;
        for (int p = 0; p < parts; p++) {
            const unsigned c = cnts[p];
            ok = ok && c <= (unsigned)cap;
            total += c;
        }
        ok = ok && total >= (unsigned)k;
        RF_T(0)
        // ---- lists of up to RF_MAX entries are STAGED in LDS first, every load of the query in flight at once: the five passes below
        //      otherwise each walk the sub-lists with dependent global loads (~80 serialised round trips per query: 0.8 of the kernel's
        //      1.4 ms at 50k x 50k x 100).  Entry e of the concatenated list lives in sub-list p = the last one with pre[p] <= e. ----
        const bool staged = ok && total <= (unsigned)RF_MAX;
        uint2 *stg = (uint2 *)comp;
        if (staged) {
            uint32_t *pre = sel;                                     // (sel is free until the compaction below)
            uint32_t run = 0;
            for (int p0 = 0; p0 < parts; p0 += 64) {
                const uint32_t c = p0 + lane < parts ? cnts[p0 + lane] : 0u;
                uint32_t incl = c;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t t = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += t;
                }
                if (p0 + lane < parts) pre[p0 + lane + 1] = run + incl;
                run += (uint32_t)__shfl((int)incl, 63, 64);
            }
            if (lane == 0) pre[0] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (uint32_t base = 0; base < total; base += 512u) {        // 8 loads per lane in flight at a time (registers)
                uint2 val[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t e = base + (uint32_t)lane + 64u * i;
                    int lo = 0, hi2 = parts - 1;                         // largest p with pre[p] <= e
                    while (lo < hi2) {
                        const int mid = (lo + hi2 + 1) >> 1;
                        if (pre[mid] <= e) lo = mid; else hi2 = mid - 1;
                    }
                    const uint32_t ec = e < total ? e : 0u;
                    const int pc = e < total ? lo : 0;
                    val[i] = sub_list(pc)[ec - (e < total ? pre[pc] : 0u)];
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t e = base + (uint32_t)lane + 64u * i;
                    if (e < total) stg[e] = val[i];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        RF_T(1)
        uint32_t m = 0;
        float B = 0.f;
        const float e_q = eps[urow], thr_q = thr[urow];
        if (ok) {
            // ---- k-th smallest d~ key (NaN keys = 0xFFFFFFFF sort last): 4 passes of 8 bits over the list ----
            uint32_t prefix = 0, pmask = 0, remaining = (uint32_t)k;
#pragma unroll 1
            for (int shift = 24; shift >= 0; shift -= 8) {
#pragma unroll
                for (int i = 0; i < 4; i++) hist[lane * 4 + i] = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (staged) {
                    for (unsigned e = lane; e < total; e += 64) {
                        const uint32_t key = canon_key(__uint_as_float(stg[e].x));
                        if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
                    }
                } else {
                    for (int p = 0; p < parts; p++) {
                        const uint2 *lp = sub_list(p);
                        const unsigned cp = cnts[p];
                        for (unsigned e = lane; e < cp; e += 64) {
                            const uint32_t key = canon_key(__uint_as_float(lp[e].x));
                            if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                uint32_t c[4], local = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) { c[i] = hist[lane * 4 + i]; local += c[i]; }
                uint32_t incl = local;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t t = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += t;
                }
                uint32_t run = incl - local, digit = 0, rem = 0;
                const bool mine = remaining > run && remaining <= incl;             // exactly one lane
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const bool hit = mine && remaining > run && remaining <= run + c[i];
                    digit = hit ? (uint32_t)(lane * 4 + i) : digit;
                    rem = hit ? remaining - run : rem;
                    run += c[i];
                }
                const int src = __ffsll((long long)__ballot(mine)) - 1;
                digit = (uint32_t)__shfl((int)digit, src, 64);
                remaining = (uint32_t)__shfl((int)rem, src, 64);
                prefix |= digit << shift;
                pmask |= 255u << shift;
            }
            const uint32_t kth = prefix;
            RF_T(2)
            ok = kth != 0xFFFFFFFFu && e_q == e_q;                                  // fewer than k finite d~, or an irregular query
            if (ok) {
                const float b1 = key_to_float(kth) + 2.05f * e_q;
                B = b1 < thr_q ? b1 : thr_q;
                const uint32_t bkey = canon_key(B);
                // ---- R: entries with d~ <= B, plus every NaN d~ ----
                if (staged) {
                    // (sel doubled as the prefix array: every lane is past its last read of it -- the loop below writes it)
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    for (unsigned e0 = 0; e0 < total; e0 += 64) {
                        const unsigned e = e0 + lane;
                        const uint2 c = stg[e < total ? e : 0];
                        const uint32_t key = canon_key(__uint_as_float(c.x));
                        const bool take = e < total && (key <= bkey || key == 0xFFFFFFFFu);
                        const uint64_t bm = __ballot(take);
                        const uint32_t pos = m + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                        if (take && pos < (uint32_t)RF_MAX) sel[pos] = c.y;
                        m += (uint32_t)__popcll(bm);
                    }
                } else
                for (int p = 0; p < parts; p++) {
                    const uint2 *lp = sub_list(p);
                    const unsigned cp = cnts[p];
                    for (unsigned e0 = 0; e0 < cp; e0 += 64) {
                        const unsigned e = e0 + lane;
                        const uint2 c = lp[e < cp ? e : 0];
                        const uint32_t key = canon_key(__uint_as_float(c.x));
                        const bool take = e < cp && (key <= bkey || key == 0xFFFFFFFFu);
                        const uint64_t bm = __ballot(take);
                        const uint32_t pos = m + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                        if (take && pos < (uint32_t)RF_MAX) sel[pos] = c.y;
                        m += (uint32_t)__popcll(bm);
                    }
                }
                ok = m <= (uint32_t)RF_MAX;
            }
        }
        RF_T(3)
        if (ok) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // ---- exact distances of R: one lane per candidate, the canonical fmaf chain over its gallery row ----
            const float *qv = queries + urow * ldq;
            const float sq_q = METRIC == SE_METRIC_EUCLID ? sqq[urow] : 0.f;
            // Long rows (D >= RF_STAGE_D) reach the chains through LDS when the list fits beside the results (m <= RF_STAGE_M, 16-byte rows,
            // K-block boundaries on multiples of 4): RF_LPC lanes fetch RF_KC * 4 contiguous bytes of ONE candidate row per instruction
            // instead of every lane fetching 16 bytes of its own row -- a quarter of the cache lines per load instruction, and DRAM sees
            // 256-byte bursts of a row instead of 64-byte ones.  RF_CPR candidates per round (one chain lane each), two chunks in flight.
            // D = 1000 shard: 12.5 -> 10.7 ms with 4 lanes x 64 bytes (16 lanes x 256 bytes: 10.9 -- not a matter of burst length); short rows (D = 100: the 20 MB gallery sits in Infinity Cache) lose
            // 0.1 ms to it and keep the direct form.
            int dtot = 0;
            for (int kb = 0; kb < kbs.n; kb++) dtot += kbs.len[kb];
            bool rows_staged = LONGROWS && VEC && m <= (uint32_t)RF_STAGE_M && dtot >= RF_STAGE_D;     // (LONGROWS: its own instantiation -- the staged form's registers cost the short-row kernel 8 % when both lived in one)
            for (int kb = 0; kb + 1 < kbs.n; kb++) rows_staged = rows_staged && (kbs.len[kb] & 3) == 0;
            if (rows_staged) {
                float *rowbuf = reinterpret_cast<float *>(comp + RF_STAGE_M);          // [RF_CPR][RF_PITCH]: the part of comp above the results
                for (uint32_t e0 = 0; e0 < m; e0 += RF_CPR) {
                    const uint32_t e = e0 + lane;
                    const bool chain_lane = lane < RF_CPR && e < m;
                    const uint32_t gi = sel[chain_lane ? e : m - 1];
                    const float *src[RF_NI];
#pragma unroll
                    for (int i = 0; i < RF_NI; i++) {
                        const uint32_t ec = e0 + (uint32_t)(i * (64 / RF_LPC) + lane / RF_LPC);
                        src[i] = gallery + (int64_t)sel[ec < m ? ec : m - 1] * ldg + 4 * (lane % RF_LPC);
                    }
                    const float tot = rf_chain_staged(src, gallery + (int64_t)gi * ldg, qv, kbs, rowbuf, lane);
                    float v;
                    if (METRIC == SE_METRIC_COSINE) v = -tot;
                    else v = (sqg[gi] + sq_q) - 2.0f * tot;
                    if (chain_lane) comp[e] = ((uint64_t)canon_key(v) << 32) | gi;
                }
            } else
            for (uint32_t e0 = 0; e0 < m; e0 += 64) {
                const uint32_t e = e0 + lane;
                const uint32_t gi = sel[e < m ? e : m - 1];
                const float tot = rf_chain<VEC>(gallery + (int64_t)gi * ldg, qv, kbs);
                float v;
                if (METRIC == SE_METRIC_COSINE) v = -tot;
                else v = (sqg[gi] + sq_q) - 2.0f * tot;
                if (e < m) comp[e] = ((uint64_t)canon_key(v) << 32) | gi;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            RF_T(4)
            // ---- sort R on (key, index); accept iff the k-th exact distance is <= B - eps ----
            uint32_t kkey = 0;
#define RF_SORT_OUT(P2)                                                                        \
    {                                                                                          \
        uint64_t sv[P2];                                                                       \
        _Pragma("unroll") for (int r = 0; r < P2; r++) {                                       \
            const int e = lane * P2 + r;                                                       \
            sv[r] = e < (int)m ? comp[e] : ~0ull;                                              \
        }                                                                                      \
        wave_bitonic_sort<uint64_t, P2>(sv, lane);                                             \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                                 \
        _Pragma("unroll") for (int r = 0; r < P2; r++) {                                       \
            const int e = lane * P2 + r;                                                       \
            if (e < k) comp[e] = sv[r];                                                        \
        }                                                                                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                                 \
    }
            if (m <= 64) RF_SORT_OUT(1)
            else if (m <= 128) RF_SORT_OUT(2)
            else if (m <= 256) RF_SORT_OUT(4)
            else if (m <= 512) RF_SORT_OUT(8)
            else RF_SORT_OUT(16)
#undef RF_SORT_OUT
            kkey = (uint32_t)(comp[k - 1] >> 32);
            RF_T(5)
            const float dk = kkey == 0xFFFFFFFFu ? __builtin_nanf("") : key_to_float(kkey);
            ok = dk <= B - 1.01f * e_q;                     // false for NaN
            if (ok) {
                for (int r = lane; r < k; r += 64) {
                    const uint64_t c = comp[r];
                    out_d[urow * k + r] = key_to_float((uint32_t)(c >> 32));
                    out_i[urow * k + r] = (int32_t)(col_offset + (int64_t)(uint32_t)c);
                }
                if (stats && lane == 0) { atomicAdd(&stats[0], m); atomicAdd(&stats[1], total); }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        if (!ok && lane == 0) { out_i[urow * k] = TK_REDO; atomicAdd(&nflag[1], 1u); }
        RF_T(6)
    }
#ifdef SE_TUNING
    if (stats && lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&rf_prof[i], (unsigned long long)t_acc[i]);
#endif
}

}  // namespace se

using namespace se;

static int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

// slab path: query rows per [rows, n] distance slab of <= 2 GiB
static int64_t topk_qtile(int64_t q, int64_t n)
{
    int64_t rows = ((int64_t)2 << 30) / (n * 4);
    rows = rows / 128 * 128;
    if (rows < 128) rows = 128;
    if (rows > q) rows = q;
    return rows;
}

// fused path: query rows per pass (candidate lists + group minima of <= 4 GiB) and the workspace layout
struct FusedLayout {
    int64_t qt, off_tau, off_cnt, off_gm, off_lists, off_scratch, total;
    // pre-filter path only (parts: sub-lists per query, cap: entries per sub-list, geom: the filter pass's job geometry)
    int parts;
    PfGeom geom;
    int64_t cap, off_eps, off_ctl, off_gimg, off_gnrm, off_gres, off_qimg, off_qnrm, off_qres, off_spill;
    int kp;
    int nsub, spill;        // pre-filter: real sub-lists per query (filter geometry) + slots of the shared spill region; parts = nsub + spill
};
static bool prefilter_wanted(const FusedPlan &p, int k)
{
    if (!p.ok || k > PF_K_MAX) return false;
    if (const char *e = tuning_env("SE_TOPK_PREFILTER")) return atoi(e) != 0;      // -DSE_TUNING build: 0 pins the fp32 fused passes
    return true;
}
static FusedLayout fused_layout(int64_t q, int64_t n, int64_t d, const FusedPlan &p, bool pf)
{
    FusedLayout L = {};
    L.cap = p.cap;
    L.parts = 1;
    int64_t qt;
    if (pf) {
        // the half-precision thresholds admit a window of 4 eps more than the exact ones the plan was made for: 1.5x the planned total,
        // cut into the filter pass's gallery parts (+ room for the relative fluctuation of a small share)
        L.geom = pf_geometry(n, q, 0, pf_padded_dim(d));
        L.nsub = L.geom.parts * L.geom.gj;
        const int64_t total_cap = align256((int64_t)p.cap * 3 / 2);
        L.cap = (total_cap / L.nsub + 32 + 15) / 16 * 16;
        if (const char *e = tuning_env("SE_TOPK_CAP")) L.cap = atoi(e) / L.nsub > 0 ? atoi(e) / L.nsub : 1;
        // + a spill region of as many sub-list slots again, shared by the query's sub-lists (prefilter.hip: pf_spill_append): a gallery
        // sorted by class sends a query's candidates to two or three of them
        L.spill = L.nsub;
        if (const char *e = tuning_env("SE_TOPK_SPILL")) L.spill = atoi(e);
        L.parts = L.nsub + L.spill;
        const int64_t per_row = L.parts * (L.cap * 8 + 4) + (int64_t)p.G * 4 + 16;
        qt = ((int64_t)4 << 30) / per_row / 128 * 128;
    } else {
        const int64_t per_row = L.cap * 8 + (int64_t)p.G * 4 + 16;
        qt = ((int64_t)4 << 30) / per_row / 128 * 128;
    }
    if (qt < 128) qt = 128;
    if (qt > q) qt = q;
    L.qt = qt;
    L.off_tau = 0;
    L.off_cnt = align256(qt * 4);
    L.off_gm = L.off_cnt + align256(qt * L.parts * 4 + 16 + qt * 4);   // [qt (x parts)] candidate counts + 4 control words (flagged-query counters) + [qt] spill counts
    L.off_lists = L.off_gm + align256(qt * p.G * 4);
    L.off_spill = L.off_lists + align256(qt * (pf ? L.nsub : L.parts) * L.cap * 8);
    L.off_scratch = L.off_spill + align256(qt * (pf ? L.spill : 0) * L.cap * 8);
    L.total = L.off_scratch + align256((int64_t)FB_GRID * n * 4);
    if (pf) {
        L.kp = pf_padded_dim(d);
        L.off_eps = L.total;                                   L.total += align256(qt * 4);
        L.off_ctl = L.total;                                   L.total += 256;
        L.off_gnrm = L.total;                                  L.total += align256(n * 4);
        L.off_gres = L.total;                                  L.total += align256(n * 4);
        L.off_qnrm = L.total;                                  L.total += align256(qt * 4);
        L.off_qres = L.total;                                  L.total += align256(qt * 4);
        L.off_gimg = L.total;                                  L.total += align256(n * (int64_t)L.kp * 2);
        L.off_qimg = L.total;                                  L.total += align256(qt * (int64_t)L.kp * 2);
    }
    return L;
}

static int topk_merge_launch(const char *who, const float *d, const int32_t *idx, int64_t part_stride, int parts, int64_t q, int k, float *out_d,
                             int32_t *out_i, se_stream_t stream)
{
    if (parts < 1 || q < 0 || k < 1) return fail(SE_ERR_INVALID, "%s: bad shape", who);
    if ((int64_t)parts * k > (int64_t)SE_TOPK_MAX * 4) return fail(SE_ERR_UNSUPPORTED, "%s: parts*k = %lld exceeds %d", who, (long long)parts * k, SE_TOPK_MAX * 4);
    if (q == 0) return SE_OK;
    if (!d || !idx || !out_d || !out_i) return fail(SE_ERR_INVALID, "%s: null pointer", who);
    if (k <= 1024) {      // one wave per query, the running best list in registers (64 * PER >= k entries)
        const int64_t wgrid = (q + 3) / 4 < 8192 ? (q + 3) / 4 : 8192;
#define SE_MERGE_WAVE(PER) hipLaunchKernelGGL(topk_merge_wave_kernel<PER>, dim3((unsigned)wgrid), dim3(256), 0, (hipStream_t)stream, d, idx, part_stride, parts, q, k, out_d, out_i)
        if (k <= 64) SE_MERGE_WAVE(1);
        else if (k <= 128) SE_MERGE_WAVE(2);
        else if (k <= 256) SE_MERGE_WAVE(4);
        else if (k <= 512) SE_MERGE_WAVE(8);
        else SE_MERGE_WAVE(16);
#undef SE_MERGE_WAVE
        SE_LAUNCH_CHECK();
        return SE_OK;
    }
    const int P = next_pow2(parts * k);
    const size_t lds = (size_t)P * 8;
    const int64_t grid = q < 4096 ? q : 4096;
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, d, idx, part_stride, parts, q, k, P, out_d, out_i);
    SE_LAUNCH_CHECK();
    return SE_OK;
}

extern "C" int se_topk_merge(const float *d, const int32_t *idx, int parts, int64_t q, int k, float *out_d,
                             int32_t *out_i, se_stream_t stream)
{
    return topk_merge_launch("se_topk_merge", d, idx, q * k, parts, q, k, out_d, out_i, stream);
}

// packed lists: per part one contiguous block of 2 q k words -- [q, k] distances (f32) followed by [q, k] indices (i32) -- i.e.
// exactly what ONE all-gather of every rank's (dist | idx) block produces
extern "C" int se_topk_merge_packed(const void *packed, int parts, int64_t q, int k, float *out_d, int32_t *out_i, se_stream_t stream)
{
    const float *d = (const float *)packed;
    const int32_t *idx = packed ? (const int32_t *)packed + q * k : nullptr;
    return topk_merge_launch("se_topk_merge_packed", d, idx, 2 * q * k, parts, q, k, out_d, out_i, stream);
}

extern "C" int64_t se_retrieve_topk_workspace_bytes(int64_t q, int64_t n, int64_t d, int64_t ldg, int k)
{
    if (q <= 0 || n <= 0 || d <= 0) return 0;
    const FusedPlan p = fused_plan(n, ldg, k);
    if (p.ok) return fused_layout(q, n, d, p, prefilter_wanted(p, k)).total;
    return topk_qtile(q, n) * ((n + 3) / 4 * 4) * 4;          // slab rows on a 16-byte pitch (16-byte row stores of the distance kernel)
}

// the bf16 pre-filter path of one query tile (see the block comment above pf_thr_kernel)
static int retrieve_topk_prefilter(const float *queries, int64_t ldq, const float *gallery, int64_t ldg, const float *sqq, const float *sqg,
                                   int64_t q, int64_t n, int64_t d, int metric, const KBlocks &kbs, int64_t col_offset, int k, float *out_d,
                                   int32_t *out_i, char *ws, const FusedPlan &p, const FusedLayout &L, hipStream_t s)
{
    float *thr = (float *)(ws + L.off_tau), *eps = (float *)(ws + L.off_eps);
    unsigned *rowcnt = (unsigned *)(ws + L.off_cnt), *ctl = (unsigned *)(ws + L.off_ctl);
    float *gm = (float *)(ws + L.off_gm);
    uint2 *lists = (uint2 *)(ws + L.off_lists), *spill_lists = (uint2 *)(ws + L.off_spill);
    float *scratch = (float *)(ws + L.off_scratch);
    uint16_t *gimg = (uint16_t *)(ws + L.off_gimg), *qimg = (uint16_t *)(ws + L.off_qimg);
    float *gnrm = (float *)(ws + L.off_gnrm), *gres = (float *)(ws + L.off_gres), *qnrm = (float *)(ws + L.off_qnrm), *qres = (float *)(ws + L.off_qres);
    const int kp = L.kp;
    const int P = next_pow2(k);
    const size_t lds_sel = (size_t)P * 8 + (TK_NB + TK_WAVES + 1 + 4) * sizeof(uint32_t);
    const bool vec = (ldg % 4 == 0) && ((((uintptr_t)gallery) & 15) == 0);
    // ---- gallery image (once per call) ----
    phase_mark("start", s);
    SE_HIP_CHECK(hipMemsetAsync(ctl, 0, 256, s));
    if (const int rc = pf_convert(gallery, ldg, n, d, gimg, gnrm, gres, ctl, s)) return rc;
    phase_mark("convert", s);
    for (int64_t q0 = 0; q0 < q; q0 += L.qt) {
        const int64_t rows = (q - q0 < L.qt) ? (q - q0) : L.qt;
        const float *qs = queries + q0 * ldq;
        const float *sq = sqq ? sqq + q0 : nullptr;
        // all-pairs call: every item is query and gallery item -- one image serves both sides
        const bool sym = (qs == gallery) && (ldq == ldg) && (rows == n) && (metric != SE_METRIC_EUCLID || sq == sqg);
        const uint16_t *qi = gimg;
        const float *qn = gnrm, *qr = gres;
        const unsigned *qc = ctl;
        if (!sym) {
            SE_HIP_CHECK(hipMemsetAsync(ctl + 8, 0, 32, s));
            if (const int rc = pf_convert(qs, ldq, rows, d, qimg, qnrm, qres, ctl + 8, s)) return rc;
            qi = qimg; qn = qnrm; qr = qres; qc = ctl + 8;
            phase_mark("convert", s);
        }
        unsigned *nflag = rowcnt + rows * L.parts;                     // [1] queries handed to the exact kernel; [2..3] statistics (tuning)
        unsigned *spill_cnt = nflag + 4;                               // [rows] entries in every query's spill region
        SE_HIP_CHECK(hipMemsetAsync(rowcnt, 0, (size_t)rows * L.parts * 4 + 16 + (size_t)rows * 4, s));
        PfPassArgs pa = {gm, p.G, thr, rowcnt, lists, L.cap, L.spill, spill_lists, spill_cnt, p.step, nullptr, 0, (int)d};
        int rc = pf_pass(PF_EPI_GROUPMIN, nullptr, metric, gimg, p.step * (int64_t)kp, qi, kp, sqg, sq, p.S, rows, kp, ctl, qc, pa, s);
        if (rc != SE_OK) return rc;
        phase_mark("sample", s);
        hipLaunchKernelGGL(pf_thr_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, gm, (int64_t)p.G, rows, p.G, p.j, qn, qr, ctl, metric, (int)d, kp,
                           kbs.n, thr, eps);
        SE_LAUNCH_CHECK();
        pa.sqa_stride = 1;
        phase_mark("threshold", s);
        rc = pf_pass(PF_EPI_FILTER, &L.geom, metric, gimg, kp, qi, kp, sqg, sq, n, rows, kp, ctl, qc, pa, s);
        if (rc != SE_OK) return rc;
        if ((rc = pf_spill_counts(rowcnt, spill_cnt, rows, L.nsub, L.spill, L.cap, s)) != SE_OK) return rc;
        phase_mark("filter", s);
        const bool verbose = kTuning && tuning_env("SE_TOPK_VERBOSE");
        const bool count = verbose || phase_timing_on();               // [2..3] of nflag: entries recomputed exactly / candidates, summed over the queries

        const int64_t rgrid = (rows + RF_WAVES - 1) / RF_WAVES < 8192 ? (rows + RF_WAVES - 1) / RF_WAVES : 8192;
#define SE_RF_LAUNCH(M, V, LR) hipLaunchKernelGGL((pf_refine_kernel<M, V, LR>), dim3((unsigned)rgrid), dim3(RF_WAVES * 64), 0, s, lists, spill_lists, L.nsub, rowcnt, L.cap, L.parts, rows, thr, eps, qs, ldq, \
                                                  gallery, ldg, sq, sqg, kbs, col_offset, k, out_d + q0 * k, out_i + q0 * k, nflag, count ? nflag + 2 : nullptr)
        const bool longrows = vec && d >= RF_STAGE_D;                  // the build with the LDS-staged row gather
        if (metric == SE_METRIC_COSINE) { if (longrows) SE_RF_LAUNCH(SE_METRIC_COSINE, true, true); else if (vec) SE_RF_LAUNCH(SE_METRIC_COSINE, true, false); else SE_RF_LAUNCH(SE_METRIC_COSINE, false, false); }
        else { if (longrows) SE_RF_LAUNCH(SE_METRIC_EUCLID, true, true); else if (vec) SE_RF_LAUNCH(SE_METRIC_EUCLID, true, false); else SE_RF_LAUNCH(SE_METRIC_EUCLID, false, false); }
#undef SE_RF_LAUNCH
        SE_LAUNCH_CHECK();
        phase_mark("refine", s);
        if (verbose) {   // -DSE_TUNING build only: synchronises and reports how the lists came out
            SE_HIP_CHECK(hipStreamSynchronize(s));
#ifdef SE_TUNING
            {
                unsigned long long hp[8], zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                SE_HIP_CHECK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(rf_prof), sizeof(hp)));
                SE_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(rf_prof), zero, sizeof(zero)));
                static const char *names[8] = {"counts", "stage-lists", "select-kth", "compact", "chains", "sort", "accept+write", "loop"};
                double tot = 0;
                for (int i = 0; i < 8; i++) tot += (double)hp[i];
                fprintf(stderr, "[pf_refine_kernel profile]");
                for (int i = 0; i < 8; i++) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * (double)hp[i] / (tot > 0 ? tot : 1));
                fprintf(stderr, "  (%.0f cycles per query and wave)\n", tot / (double)rows);
            }
#endif
            unsigned h[4], hc[8];
            SE_HIP_CHECK(hipMemcpy(h, nflag, sizeof(h), hipMemcpyDeviceToHost));
            SE_HIP_CHECK(hipMemcpy(hc, ctl, sizeof(hc), hipMemcpyDeviceToHost));
            const double okq = (double)rows - (double)h[1];
            float gmaxn, gmaxr;
            memcpy(&gmaxn, &hc[0], 4); memcpy(&gmaxr, &hc[1], 4);
            fprintf(stderr, "[se_retrieve_topk] prefilter: n=%lld d=%lld kp=%d k=%d S=%d G=%d j=%d parts=%d cap=%lld rows=%lld sym=%d redo=%u mean_candidates=%.1f "
                            "mean_recomputed=%.1f gallery_max_norm=%.4g max_residual=%.4g irregular_rows=%u\n",
                    (long long)n, (long long)d, kp, k, p.S, p.G, p.j, L.parts, (long long)L.cap, (long long)rows, (int)sym, h[1],
                    okq > 0 ? (double)h[3] / okq : 0.0, okq > 0 ? (double)h[2] / okq : 0.0, (double)gmaxn, (double)gmaxr, hc[2]);
        }
        const int64_t fgrid = rows < FB_GRID ? rows : FB_GRID;
        if (metric == SE_METRIC_COSINE)
            hipLaunchKernelGGL(topk_fallback_kernel<SE_METRIC_COSINE>, dim3((unsigned)fgrid), dim3(TK_THREADS), ((lds_sel + 15) & ~(size_t)15) + FB_ROWBUF_BYTES, s, qs, ldq, gallery, ldg, sq, sqg,
                               rows, (int)n, (int)d, kbs, col_offset, k, P, (int)((lds_sel + 15) & ~(size_t)15), scratch, out_d + q0 * k, out_i + q0 * k, nflag);
        else
            hipLaunchKernelGGL(topk_fallback_kernel<SE_METRIC_EUCLID>, dim3((unsigned)fgrid), dim3(TK_THREADS), ((lds_sel + 15) & ~(size_t)15) + FB_ROWBUF_BYTES, s, qs, ldq, gallery, ldg, sq, sqg,
                               rows, (int)n, (int)d, kbs, col_offset, k, P, (int)((lds_sel + 15) & ~(size_t)15), scratch, out_d + q0 * k, out_i + q0 * k, nflag);
        SE_LAUNCH_CHECK();
        phase_mark("fallback", s);
        phase_note_counters(nflag, rows, s);      // final by now (refinement + exact fallback have been enqueued): copied on this stream
    }
    return SE_OK;
}

extern "C" int se_retrieve_topk(const float *queries, int64_t ldq, const float *gallery, int64_t ldg,
                                const float *sqq, const float *sqg, int64_t q, int64_t n, int64_t d,
                                int metric, const int32_t *kblocks, int nkb, int64_t col_offset, int k,
                                float *out_d, int32_t *out_i, void *workspace, int64_t workspace_bytes, se_stream_t stream)
{
    if (q < 0 || n <= 0 || d <= 0 || n > 0x7FFFFFFFll || q > 0x7FFFFFFFll) return fail(SE_ERR_INVALID, "se_retrieve_topk: bad shape");
    if (k < 1 || k > SE_TOPK_MAX || k > n) return fail(SE_ERR_INVALID, "se_retrieve_topk: need 1 <= k <= min(n, %d), got k=%d n=%lld", SE_TOPK_MAX, k, (long long)n);
    if (q == 0) return SE_OK;
    if (!queries || !gallery || !out_d || !out_i || ldq < d || ldg < d) return fail(SE_ERR_INVALID, "se_retrieve_topk: bad argument");
    if (metric != SE_METRIC_COSINE && metric != SE_METRIC_EUCLID) return fail(SE_ERR_INVALID, "se_retrieve_topk: metric must be SE_METRIC_COSINE or SE_METRIC_EUCLID");
    if (metric == SE_METRIC_EUCLID && (!sqq || !sqg)) return fail(SE_ERR_INVALID, "se_retrieve_topk: SE_METRIC_EUCLID needs sqq and sqg");
    KBlocks kbs;
    bool multi = false;
    if (const int rc = make_kblocks("se_retrieve_topk", kblocks, nkb, d, kbs, multi)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const FusedPlan p = fused_plan(n, ldg, k);
    const bool pf = prefilter_wanted(p, k);
    const FusedLayout L = p.ok ? fused_layout(q, n, d, p, pf) : FusedLayout{};
    const int64_t need = p.ok ? L.total : topk_qtile(q, n) * ((n + 3) / 4 * 4) * 4;
    if (!workspace || workspace_bytes < need) return fail(SE_ERR_WORKSPACE, "se_retrieve_topk: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    if (p.ok && (((uintptr_t)workspace) & 255) != 0) return fail(SE_ERR_INVALID, "se_retrieve_topk: workspace must be 256-byte aligned");
    if (pf) return retrieve_topk_prefilter(queries, ldq, gallery, ldg, sqq, sqg, q, n, d, metric, kbs, col_offset, k, out_d, out_i, (char *)workspace, p, L, s);

    if (!p.ok) {   // ---- small problems: [rows, n] distance slab -> se_topk_rows, query tile by query tile ----
        const int64_t qt = topk_qtile(q, n);
        float *slab = (float *)workspace;
        for (int64_t q0 = 0; q0 < q; q0 += qt) {
            const int64_t rows = (q - q0 < qt) ? (q - q0) : qt;
            const int64_t lds_ = (n + 3) / 4 * 4;
            int rc = se_pairwise_dist(queries + q0 * ldq, ldq, gallery, ldg, sqq ? sqq + q0 : nullptr, sqg, rows, n, d,
                                      metric, multi ? kblocks : nullptr, multi ? nkb : 0, slab, lds_, stream);
            if (rc != SE_OK) return rc;
            rc = se_topk_rows(slab, lds_, rows, n, col_offset, k, out_d + q0 * k, out_i + q0 * k, stream);
            if (rc != SE_OK) return rc;
        }
        return SE_OK;
    }

    // ---- fused path, fp32 passes (k > 512, or pinned by the tuning build) ----
    char *ws = (char *)workspace;
    float *tau = (float *)(ws + L.off_tau);
    unsigned *rowcnt = (unsigned *)(ws + L.off_cnt);
    float *gm = (float *)(ws + L.off_gm);
    uint2 *lists = (uint2 *)(ws + L.off_lists);
    float *scratch = (float *)(ws + L.off_scratch);
    int PER = 1;
    while (PER * TK_THREADS < p.cap) PER <<= 1;
    const size_t lds_lists = (size_t)PER * TK_THREADS * 8;
    const int P = next_pow2(k);
    const size_t lds_sel = (size_t)P * 8 + (TK_NB + TK_WAVES + 1 + 4) * sizeof(uint32_t);
    SE_HIP_CHECK(hipFuncSetAttribute((const void *)topk_lists_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lists));
    for (int64_t q0 = 0; q0 < q; q0 += L.qt) {
        const int64_t rows = (q - q0 < L.qt) ? (q - q0) : L.qt;
        const float *qs = queries + q0 * ldq;
        const float *sq = sqq ? sqq + q0 : nullptr;
        unsigned *nflag = rowcnt + rows;                               // [0] queries handed to the workgroup-wide sort, [1] to the exact kernel
        SE_HIP_CHECK(hipMemsetAsync(rowcnt, 0, (size_t)rows * 4 + 16, s));
        FusedArgs fa = {gm, p.G, tau, rowcnt, lists, p.cap, p.step};
        int rc = launch_fused_pass(EPI_GROUPMIN, gallery, p.step * ldg, qs, ldq, sqg, sq, p.S, rows, d, metric, kbs, multi, fa, s);
        if (rc != SE_OK) return rc;
        hipLaunchKernelGGL(topk_tau_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, gm, (int64_t)p.G, rows, p.G, p.j, tau);
        SE_LAUNCH_CHECK();
        fa.sqa_stride = 1;
        rc = launch_fused_pass(EPI_FILTER, gallery, ldg, qs, ldq, sqg, sq, n, rows, d, metric, kbs, multi, fa, s);
        if (rc != SE_OK) return rc;
        const int64_t grid = rows < 2048 ? rows : 2048;
        int only_wide = 0;
        if (k <= 256 && !tuning_env("SE_TOPK_NOWAVE")) {     // one wave per query (radix select + small sort); hands the rest on
            const int64_t wgrid = (rows + TL_WAVES - 1) / TL_WAVES < 4096 ? (rows + TL_WAVES - 1) / TL_WAVES : 4096;
            hipLaunchKernelGGL(topk_lists_wave_kernel, dim3((unsigned)wgrid), dim3(TL_WAVES * 64), 0, s, lists, rowcnt, (int64_t)p.cap, rows,
                               col_offset, k, out_d + q0 * k, out_i + q0 * k, nflag);
            SE_LAUNCH_CHECK();
            only_wide = 1;
        }
        hipLaunchKernelGGL(topk_lists_kernel, dim3((unsigned)grid), dim3(TK_THREADS), lds_lists, s, lists, rowcnt, (int64_t)p.cap, rows,
                           col_offset, k, out_d + q0 * k, out_i + q0 * k, only_wide, nflag);
        SE_LAUNCH_CHECK();
        if (kTuning && tuning_env("SE_TOPK_VERBOSE")) {   // -DSE_TUNING build only: synchronises and reports how the lists came out
            SE_HIP_CHECK(hipStreamSynchronize(s));
            unsigned *h = (unsigned *)malloc((size_t)rows * 4);
            SE_HIP_CHECK(hipMemcpy(h, rowcnt, (size_t)rows * 4, hipMemcpyDeviceToHost));
            int64_t flagged = 0;
            double sum = 0;
            for (int64_t i = 0; i < rows; i++) { flagged += (h[i] < (unsigned)k || h[i] > (unsigned)p.cap); sum += h[i]; }
            free(h);
            fprintf(stderr, "[se_retrieve_topk] fused: n=%lld k=%d S=%d step=%lld G=%d j=%d cap=%d rows=%lld flagged=%lld mean_candidates=%.1f\n",
                    (long long)n, k, p.S, (long long)p.step, p.G, p.j, p.cap, (long long)rows, (long long)flagged, sum / (double)rows);
        }
        const int64_t fgrid = rows < FB_GRID ? rows : FB_GRID;
        if (metric == SE_METRIC_COSINE)
            hipLaunchKernelGGL(topk_fallback_kernel<SE_METRIC_COSINE>, dim3((unsigned)fgrid), dim3(TK_THREADS), ((lds_sel + 15) & ~(size_t)15) + FB_ROWBUF_BYTES, s, qs, ldq, gallery, ldg, sq, sqg,
                               rows, (int)n, (int)d, kbs, col_offset, k, P, (int)((lds_sel + 15) & ~(size_t)15), scratch, out_d + q0 * k, out_i + q0 * k, nflag);
        else
            hipLaunchKernelGGL(topk_fallback_kernel<SE_METRIC_EUCLID>, dim3((unsigned)fgrid), dim3(TK_THREADS), ((lds_sel + 15) & ~(size_t)15) + FB_ROWBUF_BYTES, s, qs, ldq, gallery, ldg, sq, sqg,
                               rows, (int)n, (int)d, kbs, col_offset, k, P, (int)((lds_sel + 15) & ~(size_t)15), scratch, out_d + q0 * k, out_i + q0 * k, nflag);
        SE_LAUNCH_CHECK();
    }
    return SE_OK;
}

#ifdef SE_TUNING
// -DSE_TUNING build only (tests of the error bound of the bf16 pre-filter; not part of include/sehip.h): d~ of every (gallery row,
// query) pair as the pre-filter's tile loop computes it -> out_dt [n, ldo >= q], and eps(query) as pf_thr_kernel derives it -> out_eps [q].
//   workspace: 2 (n + q) pf_padded_dim(d) + 16 (n + q) + 4 q + 4096 bytes, 256-byte aligned.
extern "C" int se_tuning_prefilter_probe(const float *queries, int64_t ldq, const float *gallery, int64_t ldg, const float *sqq, const float *sqg,
                                         int64_t q, int64_t n, int64_t d, int metric, int nkb, float *out_dt, int64_t ldo, float *out_eps,
                                         void *workspace, int64_t workspace_bytes, se_stream_t stream)
{
    hipStream_t s = (hipStream_t)stream;
    const int kp = pf_padded_dim(d);
    char *w = (char *)workspace;
    unsigned *ctl = (unsigned *)w;                      w += 256;
    float *gnrm = (float *)w;                           w += align256(n * 4);
    float *gres = (float *)w;                           w += align256(n * 4);
    float *qnrm = (float *)w;                           w += align256(q * 4);
    float *qres = (float *)w;                           w += align256(q * 4);
    float *gm = (float *)w;                             w += align256(q * 4);      // one "group minimum" per query: +inf -> tau is irrelevant here
    float *thr = (float *)w;                            w += align256(q * 4);
    uint16_t *gimg = (uint16_t *)w;                     w += align256(n * (int64_t)kp * 2);
    uint16_t *qimg = (uint16_t *)w;                     w += align256(q * (int64_t)kp * 2);
    if (w - (char *)workspace > workspace_bytes) return fail(SE_ERR_WORKSPACE, "se_tuning_prefilter_probe: workspace too small");
    SE_HIP_CHECK(hipMemsetAsync(ctl, 0, 256, s));
    SE_HIP_CHECK(hipMemsetAsync(gm, 0, (size_t)q * 4, s));
    if (const int rc = pf_convert(gallery, ldg, n, d, gimg, gnrm, gres, ctl, s)) return rc;
    if (const int rc = pf_convert(queries, ldq, q, d, qimg, qnrm, qres, ctl + 8, s)) return rc;
    PfPassArgs pa = {nullptr, 0, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 1, out_dt, ldo, 0};
    if (const int rc = pf_pass(PF_EPI_STORE, nullptr, metric, gimg, kp, qimg, kp, sqg, sqq, n, q, kp, ctl, ctl + 8, pa, s)) return rc;
    hipLaunchKernelGGL(pf_thr_kernel, dim3((unsigned)((q + 3) / 4)), dim3(256), 0, s, gm, (int64_t)1, q, 1, 1, qnrm, qres, ctl, metric, (int)d, kp, nkb, thr, out_eps);
    SE_LAUNCH_CHECK();
    return SE_OK;
}
#endif
