// rank_rows.hip -- full ranking of every row of a distance matrix.
//
// Replaces `ranking = np.argsort(pdist, axis = -1)` (evaluate_retrieval.py:67) with the canonical
// tie rule: ascending (distance, index), NaN last, -0 == +0.
//
// One workgroup sorts one row (a segment of N keys) with a stable LSD radix sort (8-bit digits, 4
// passes) on the order-preserving uint32 image of the float keys; stability + initial index order
// give the index tiebreak for free.  Structure per row:
//   H. one streaming read of the row builds the digit histograms of ALL four passes (LDS atomics);
//      their exclusive scans are the running output cursors gbase[pass][digit].
//   P. each pass walks the row in tiles of RK_TILE keys held in registers (wave w owns a contiguous
//      sub-chunk of the tile, 64 keys per step, coalesced loads):
//        1. per-wave digit counts of the tile (LDS atomics) -> digit-major / wave-minor scan;
//        2. stable rank inside the tile: lanes holding equal digits find each other with 8 ballots
//           (wave multisplit), one LDS cursor read+write per digit group; keys/indices are written
//           to their tile-sorted slot IN LDS;
//        3. the tile is read back in sorted order and written to global memory at
//           gbase[digit] + offset: consecutive lanes write consecutive addresses inside each digit
//           run, so HBM/L2 see coalesced runs (~RK_TILE/256 keys) instead of 4-byte scatters.
// Keys/indices ping-pong between two scratch rows per resident workgroup in the caller's workspace;
// the last pass writes the index matrix only.
// HBM-side algorithmic traffic: 4 N bytes in (distances) + 4 N bytes out (int32 ranks) per row.
#include "se_common.h"

namespace se {

constexpr int RK_THREADS = 512;
constexpr int RK_WAVES = RK_THREADS / WAVE;
constexpr int RK_ITEMS = 16;                       // keys per thread per tile
constexpr int RK_TILE = RK_THREADS * RK_ITEMS;     // 8192
constexpr int RK_WCHUNK = RK_TILE / RK_WAVES;      // 1024 keys per wave per tile
constexpr int RK_NB = 256;

struct RankLds {
    uint32_t gbase[4][RK_NB];        // running global cursors per pass
    uint32_t wcnt[RK_WAVES][RK_NB];  // per-wave cursors inside the current tile
    uint32_t tile_start[RK_NB];      // first tile-sorted slot of each digit
    uint32_t wave_tot[RK_WAVES];
    uint32_t tkeys[RK_TILE];
    uint32_t tidx[RK_TILE];
};

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total)
{
    const int lane = lane_id();
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    total = __shfl(incl, 63, 64);
    return incl - v;
}

template <bool IDX64>
__global__ __launch_bounds__(RK_THREADS) void rank_rows_kernel(const float *__restrict__ pdist, int64_t ldp,
                                                               int64_t Q, int N, void *rank, int64_t ldr,
                                                               uint32_t *ws, int64_t n_pad)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rk_raw[];
    RankLds &L = *reinterpret_cast<RankLds *>(rk_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    uint32_t *kA = ws + (int64_t)blockIdx.x * 4 * n_pad;
    uint32_t *iA = kA + n_pad, *kB = iA + n_pad, *iB = kB + n_pad;
    const int ntiles = (N + RK_TILE - 1) / RK_TILE;

    for (int64_t row = blockIdx.x; row < Q; row += gridDim.x) {
        const float *drow = pdist + row * ldp;

        // ---------------- H: histograms of all four digits, one read of the row ----------------
        __syncthreads();
        for (int i = tid; i < 4 * RK_NB; i += RK_THREADS) (&L.gbase[0][0])[i] = 0;
        __syncthreads();
        for (int i = tid; i < N; i += RK_THREADS) {
            const uint32_t key = canon_key(drow[i]);
            atomicAdd(&L.gbase[0][key & 0xFF], 1u);
            atomicAdd(&L.gbase[1][(key >> 8) & 0xFF], 1u);
            atomicAdd(&L.gbase[2][(key >> 16) & 0xFF], 1u);
            atomicAdd(&L.gbase[3][key >> 24], 1u);
        }
        __syncthreads();
        if (wave < 4) {  // wave p scans histogram p (4 digits per lane)
            uint32_t c[4], s = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { c[j] = L.gbase[wave][lane * 4 + j]; s += c[j]; }
            uint32_t tot;
            uint32_t run = wave_excl_scan(s, tot);
#pragma unroll
            for (int j = 0; j < 4; j++) { L.gbase[wave][lane * 4 + j] = run; run += c[j]; }
        }
        __syncthreads();

        // ---------------- P: four stable counting passes ----------------
#pragma unroll 1
        for (int p = 0; p < 4; p++) {
            const int shift = p * 8;
            const uint32_t *sk = (p & 1) ? kA : kB;
            const uint32_t *si = (p & 1) ? iA : iB;
            uint32_t *dk = (p & 1) ? kB : kA;
            uint32_t *di = (p & 1) ? iB : iA;
            const bool last = (p == 3);
#pragma unroll 1
            for (int t = 0; t < ntiles; t++) {
                const int tbase = t * RK_TILE;
                const int tcount = (N - tbase < RK_TILE) ? (N - tbase) : RK_TILE;
                const int wbeg = wave * RK_WCHUNK;   // tile-local first element of this wave

                // 1a. load the wave's sub-chunk (coalesced), zero own counters.  Loads are unconditional
                //     (clamped index + select): branching around each load would serialise them.
                uint32_t key[RK_ITEMS], idx[RK_ITEMS];
                if (p == 0) {
#pragma unroll
                    for (int s = 0; s < RK_ITEMS; s++) {
                        const int li = wbeg + s * WAVE + lane;
                        const int gi = tbase + li;
                        const float f = drow[gi < N ? gi : N - 1];
                        key[s] = (li < tcount) ? canon_key(f) : 0xFFFFFFFFu;
                        idx[s] = (uint32_t)gi;
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < RK_ITEMS; s++) {
                        const int li = wbeg + s * WAVE + lane;
                        const int gi = tbase + li;
                        const int gc = gi < N ? gi : N - 1;
                        const uint32_t k = sk[gc], i2 = si[gc];
                        key[s] = (li < tcount) ? k : 0xFFFFFFFFu;
                        idx[s] = i2;
                    }
                }
#pragma unroll
                for (int j = 0; j < RK_NB / WAVE; j++) L.wcnt[wave][j * WAVE + lane] = 0;
                // 1b. per-wave digit counts
#pragma unroll
                for (int s = 0; s < RK_ITEMS; s++) {
                    const int li = wbeg + s * WAVE + lane;
                    if (li < tcount) atomicAdd(&L.wcnt[wave][(key[s] >> shift) & 0xFF], 1u);
                }
                __syncthreads();
                // 1c. digit-major / wave-minor exclusive scan (threads 0..255 own one digit each)
                uint32_t my_total = 0;
                if (tid < RK_NB) {
                    uint32_t run = 0;
#pragma unroll
                    for (int w = 0; w < RK_WAVES; w++) {
                        const uint32_t c = L.wcnt[w][tid];
                        L.wcnt[w][tid] = run;
                        run += c;
                    }
                    my_total = run;
                    uint32_t wtot;
                    const uint32_t ex = wave_excl_scan(my_total, wtot);
                    if (lane == 63) L.wave_tot[wave] = wtot;
                    L.tile_start[tid] = ex;  // still missing the totals of the lower waves
                }
                __syncthreads();
                if (tid < RK_NB) {
                    uint32_t add = 0;
                    for (int w = 0; w < wave; w++) add += L.wave_tot[w];
                    const uint32_t st = L.tile_start[tid] + add;
                    L.tile_start[tid] = st;
#pragma unroll
                    for (int w = 0; w < RK_WAVES; w++) L.wcnt[w][tid] += st;
                }
                __syncthreads();

                // 2. stable rank inside the tile, place into LDS in tile-sorted order
                volatile uint32_t *cur = L.wcnt[wave];
#pragma unroll
                for (int s = 0; s < RK_ITEMS; s++) {
                    const int li = wbeg + s * WAVE + lane;
                    const bool valid = li < tcount;
                    const uint32_t dgt = (key[s] >> shift) & 0xFF;
                    uint64_t peers = __ballot(valid);
#pragma unroll
                    for (int bb = 0; bb < 8; bb++) {
                        const bool bit = (dgt >> bb) & 1u;
                        const uint64_t m = __ballot(bit && valid);
                        peers &= bit ? m : ~m;
                    }
                    if (valid) {
                        const int rnk = __popcll(peers & lt_mask);
                        const int npeers = __popcll(peers);
                        const uint32_t base = cur[dgt];
                        if (rnk == npeers - 1) cur[dgt] = base + (uint32_t)npeers;
                        L.tkeys[base + rnk] = key[s];
                        L.tidx[base + rnk] = idx[s];
                    }
                }
                __syncthreads();

                // 3. coalesced write-out of the tile-sorted keys
#pragma unroll
                for (int j = 0; j < RK_ITEMS; j++) {
                    const int i = j * RK_THREADS + tid;
                    if (i < tcount) {
                        const uint32_t k = L.tkeys[i];
                        const uint32_t id = L.tidx[i];
                        const uint32_t d = (k >> shift) & 0xFF;
                        const uint32_t gpos = L.gbase[p][d] + ((uint32_t)i - L.tile_start[d]);
                        if (last) {
                            if (IDX64) ((int64_t *)rank)[row * ldr + gpos] = (int64_t)id;
                            else ((int32_t *)rank)[row * ldr + gpos] = (int32_t)id;
                        } else {
                            dk[gpos] = k;
                            di[gpos] = id;
                        }
                    }
                }
                __syncthreads();
                if (tid < RK_NB) L.gbase[p][tid] += my_total;
                // (next tile's first barrier orders this update before its use)
            }
        }
    }
}

}  // namespace se

using namespace se;

static int rank_grid(int64_t q)
{
    // resident workgroups: 256 CUs x 2 (512 threads, ~78 KB LDS each)
    const int64_t g = 512;
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
    const size_t lds = sizeof(RankLds);
    hipStream_t s = (hipStream_t)stream;
    if (idx64) {
        SE_HIP_CHECK(hipFuncSetAttribute((const void *)rank_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(rank_rows_kernel<true>, dim3((unsigned)rank_grid(q)), dim3(RK_THREADS), lds, s, pdist, ldp, q, (int)n, rank, ldr, (uint32_t *)workspace, rank_npad(n));
    } else {
        SE_HIP_CHECK(hipFuncSetAttribute((const void *)rank_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(rank_rows_kernel<false>, dim3((unsigned)rank_grid(q)), dim3(RK_THREADS), lds, s, pdist, ldp, q, (int)n, rank, ldr, (uint32_t *)workspace, rank_npad(n));
    }
    SE_LAUNCH_CHECK();
    return SE_OK;
}
