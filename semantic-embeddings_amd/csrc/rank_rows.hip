// rank_rows.hip -- full ranking of every row of a distance matrix.
//
// Replaces `ranking = np.argsort(pdist, axis = -1)` (evaluate_retrieval.py:67) with the canonical
// tie rule: ascending (distance, index), NaN last, -0 == +0.
//
// One workgroup sorts one row (a "segment" of N keys) with a stable LSD radix sort on the
// order-preserving uint32 image of the float keys.  Stability + initial index order give the
// index tiebreak for free.  Each wave owns a contiguous chunk of the row:
//   1. per-wave digit histograms (LDS atomics),
//   2. one digit-major / wave-minor exclusive scan turns them into scatter cursors,
//   3. each wave walks its chunk in order, 64 keys per step; lanes holding equal digits find each
//      other with BITS ballots (wave-level multisplit), which yields the stable rank inside the
//      step without LDS traffic; one LDS read + one LDS write per digit group moves the cursor.
// Keys/indices ping-pong between two scratch rows in the caller's workspace (they stay in L2 /
// Infinity Cache: 2 x 8 B x N per resident workgroup); the last pass writes indices only.
// HBM-side algorithmic traffic: 4 N bytes in (distances) + 4 N bytes out (int32 ranks) per row.
#include "se_common.h"

namespace se {

constexpr int RK_THREADS = 512;
constexpr int RK_WAVES = RK_THREADS / WAVE;

template <int BITS>
__device__ __forceinline__ void rank_one_pass(
    int pass, int shift, int nbits_this, int N, int chunk,
    const float *__restrict__ drow,                 // pass 0 source (distances)
    const uint32_t *src_k, const uint32_t *src_i,   // later passes
    uint32_t *dst_k, uint32_t *dst_i,               // all but the last pass
    void *out_row, int idx64, bool last,
    uint32_t *cnt /* LDS [RK_WAVES][1 << BITS] */, uint32_t *scan_tmp /* LDS [64] */)
{
    constexpr int NB = 1 << BITS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t mask = (1u << nbits_this) - 1u;

    for (int i = tid; i < RK_WAVES * NB; i += RK_THREADS) cnt[i] = 0;
    __syncthreads();

    const int beg = wave * chunk;
    const int end = (beg + chunk < N) ? (beg + chunk) : N;

    // ---- 1. per-wave histogram ----
    uint32_t *mycnt = cnt + wave * NB;
    for (int i = beg + lane; i < end; i += WAVE) {
        const uint32_t key = (pass == 0) ? canon_key(drow[i]) : src_k[i];
        atomicAdd(&mycnt[(key >> shift) & mask], 1u);
    }
    __syncthreads();

    // ---- 2. exclusive scan, digit-major then wave-minor (done by wave 0) ----
    if (wave == 0) {
        constexpr int PER = NB / WAVE;  // digits per lane (NB >= 64)
        uint32_t local = 0;
        for (int dd = 0; dd < PER; dd++) {
            const int dgt = lane * PER + dd;
            for (int w = 0; w < RK_WAVES; w++) local += cnt[w * NB + dgt];
        }
        // exclusive wave scan of `local`
        uint32_t incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        uint32_t run = incl - local;
        for (int dd = 0; dd < PER; dd++) {
            const int dgt = lane * PER + dd;
            for (int w = 0; w < RK_WAVES; w++) {
                const uint32_t c = cnt[w * NB + dgt];
                cnt[w * NB + dgt] = run;
                run += c;
            }
        }
    }
    (void)scan_tmp;
    __syncthreads();

    // ---- 3. stable scatter, 64 keys per step ----
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    volatile uint32_t *cur = mycnt;
    for (int i0 = beg; i0 < end; i0 += WAVE) {
        const int i = i0 + lane;
        const bool valid = i < end;
        uint32_t key = 0, idx = 0;
        if (valid) {
            if (pass == 0) { key = canon_key(drow[i]); idx = (uint32_t)i; }
            else { key = src_k[i]; idx = src_i[i]; }
        }
        const uint32_t dgt = (key >> shift) & mask;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int bb = 0; bb < BITS; bb++) {
            const bool bit = (dgt >> bb) & 1u;
            const uint64_t m = __ballot(bit && valid);
            peers &= bit ? m : ~m;
        }
        const int rank = __popcll(peers & lt_mask);
        const int npeers = __popcll(peers);
        uint32_t pos = 0;
        if (valid) {
            const uint32_t base = cur[dgt];
            pos = base + (uint32_t)rank;
            if (rank == npeers - 1) cur[dgt] = base + (uint32_t)npeers;
        }
        if (valid) {
            if (last) {
                if (idx64) ((int64_t *)out_row)[pos] = (int64_t)idx;
                else ((int32_t *)out_row)[pos] = (int32_t)idx;
            } else {
                dst_k[pos] = key;
                dst_i[pos] = idx;
            }
        }
    }
    __syncthreads();
}

template <int BITS>
__global__ __launch_bounds__(RK_THREADS) void rank_rows_kernel(const float *__restrict__ pdist, int64_t ldp,
                                                               int64_t Q, int N, void *rank, int idx64,
                                                               int64_t ldr, uint32_t *ws, int64_t n_pad)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t rk_lds[];
    uint32_t *cnt = rk_lds;
    uint32_t *scan_tmp = rk_lds + RK_WAVES * (1 << BITS);
    // per-workgroup scratch: keysA, idxA, keysB, idxB
    uint32_t *kA = ws + (int64_t)blockIdx.x * 4 * n_pad;
    uint32_t *iA = kA + n_pad, *kB = iA + n_pad, *iB = kB + n_pad;
    constexpr int NPASS = (32 + BITS - 1) / BITS;
    int chunk = (N + RK_WAVES - 1) / RK_WAVES;
    chunk = (chunk + WAVE - 1) / WAVE * WAVE;

    for (int64_t row = blockIdx.x; row < Q; row += gridDim.x) {
        const float *drow = pdist + row * ldp;
        void *orow = idx64 ? (void *)((int64_t *)rank + row * ldr) : (void *)((int32_t *)rank + row * ldr);
        for (int p = 0; p < NPASS; p++) {
            const int shift = p * BITS;
            const int nb = (32 - shift < BITS) ? (32 - shift) : BITS;
            const uint32_t *sk = (p & 1) ? kA : kB;
            const uint32_t *si = (p & 1) ? iA : iB;
            uint32_t *dk = (p & 1) ? kB : kA;
            uint32_t *di = (p & 1) ? iB : iA;
            rank_one_pass<BITS>(p, shift, nb, N, chunk, drow, sk, si, dk, di, orow, idx64, p == NPASS - 1, cnt, scan_tmp);
        }
    }
}

}  // namespace se

using namespace se;

static int rank_grid(int64_t q)
{
    // resident workgroups: 256 CUs x up to 4 (512-thread, <= 16 KB LDS) -> 1024 scratch slots
    int64_t g = 1024;
    return (int)(q < g ? q : g);
}

static int64_t rank_npad(int64_t n) { return (n + 63) / 64 * 64; }

extern "C" int64_t se_rank_rows_workspace_bytes(int64_t q, int64_t n)
{
    if (q <= 0 || n <= 0) return 0;
    return (int64_t)rank_grid(q) * 4 * rank_npad(n) * (int64_t)sizeof(uint32_t);
}

extern "C" int se_rank_rows(const float *pdist, int64_t ldp, int64_t q, int64_t n, void *rank, int idx64,
                            int64_t ldr, void *workspace, int64_t workspace_bytes, se_stream_t stream)
{
    if (q < 0 || n < 0 || n > 0x7FFFFFFFll) return fail(SE_ERR_INVALID, "se_rank_rows: bad shape q=%lld n=%lld", (long long)q, (long long)n);
    if (q == 0 || n == 0) return SE_OK;
    if (!pdist || !rank || ldp < n || ldr < n) return fail(SE_ERR_INVALID, "se_rank_rows: bad argument");
    const int64_t need = se_rank_rows_workspace_bytes(q, n);
    if (!workspace || workspace_bytes < need) return fail(SE_ERR_WORKSPACE, "se_rank_rows: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    constexpr int BITS = 8;
    const size_t lds = (size_t)(RK_WAVES * (1 << BITS) + 64) * sizeof(uint32_t);
    hipLaunchKernelGGL(rank_rows_kernel<BITS>, dim3((unsigned)rank_grid(q)), dim3(RK_THREADS), lds, (hipStream_t)stream,
                       pdist, ldp, q, (int)n, rank, idx64, ldr, (uint32_t *)workspace, rank_npad(n));
    SE_LAUNCH_CHECK();
    return SE_OK;
}
