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
#include <stdlib.h>
#include <atomic>
#include <type_traits>

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

// Exclusive prefix sum over the 64 lanes with data-parallel primitives (DPP): four row_shr steps scan each row of 16 lanes, row_bcast15 /
// row_bcast31 carry the row totals on -- six VALU instructions instead of six ds_bpermute round trips (__shfl_up).
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total)
{
    uint32_t incl = v;
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false);   // row_shr:1
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false);   // row_shr:2
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false);   // row_shr:4
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false);   // row_shr:8
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false);   // row_bcast15 -> rows 1 and 3
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false);   // row_bcast31 -> rows 2 and 3
    total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
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
        wg_barrier();
        for (int i = tid; i < 4 * RK_NB; i += RK_THREADS) (&L.gbase[0][0])[i] = 0;
        wg_barrier();
        for (int i = tid; i < N; i += RK_THREADS) {
            const uint32_t key = canon_key(drow[i]);
            atomicAdd(&L.gbase[0][key & 0xFF], 1u);
            atomicAdd(&L.gbase[1][(key >> 8) & 0xFF], 1u);
            atomicAdd(&L.gbase[2][(key >> 16) & 0xFF], 1u);
            atomicAdd(&L.gbase[3][key >> 24], 1u);
        }
        wg_barrier();
        if (wave < 4) {  // wave p scans histogram p (4 digits per lane)
            uint32_t c[4], s = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { c[j] = L.gbase[wave][lane * 4 + j]; s += c[j]; }
            uint32_t tot;
            uint32_t run = wave_excl_scan(s, tot);
#pragma unroll
            for (int j = 0; j < 4; j++) { L.gbase[wave][lane * 4 + j] = run; run += c[j]; }
        }
        wg_barrier();

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
                wg_barrier();
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
                wg_barrier();
                if (tid < RK_NB) {
                    uint32_t add = 0;
                    for (int w = 0; w < wave; w++) add += L.wave_tot[w];
                    const uint32_t st = L.tile_start[tid] + add;
                    L.tile_start[tid] = st;
#pragma unroll
                    for (int w = 0; w < RK_WAVES; w++) L.wcnt[w][tid] += st;
                }
                wg_barrier();

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
                wg_barrier();

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
                wg_barrier();
                if (tid < RK_NB) L.gbase[p][tid] += my_total;
                // (next tile's first barrier orders this update before its use)
            }
        }
    }
}


// =================================================================================================
// Register-resident variant (the fast path, N <= RR_MAX_N): one 512-thread workgroup holds a whole
// row in registers (ITEMS keys per thread), so a row costs ONE HBM read of the distances and ONE HBM
// write of the ranks -- the algorithmic minimum -- and no global round trip sits between the passes.
//   * position p of the current arrangement lives in wave p / (64 ITEMS), step (p / 64) % ITEMS,
//     lane p % 64; positions >= N hold padding keys above every real key (rr_pad_key), so they end up
//     last and are never written out;
//   * per pass: (R) stable within-wave rank of every key on the wave's private LDS counters -- ONE returning add per key
//     (hardware-ordered build: 3 passes of 11 + 11 + 10 or 10 + 10 + 12 bits) or an 8-ballot wave multisplit + one add per
//     digit group (guaranteed-order build: 4 passes of 8 bits); (S) digit-major / wave-minor scan of the counters;
//     (X) in-place exchange through a 2-byte-per-key LDS buffer (160 KB of LDS cannot hold 4 B x 50k keys): the 16-bit
//     index, then the key halves that later passes still need -- 6 (8) two-byte exchanges per key instead of 9 (12);
//   * indices travel as 16 bits (N <= 65536 by construction) packed with the 16-bit destination;
//   * after the last pass the exchange buffer IS the ranking: it is streamed to HBM with 16-byte
//     stores.
#ifndef SE_RR_THREADS
#define SE_RR_THREADS 512   // 512 (8 waves, 2 per SIMD, <= 256 VGPRs) or 768 (12 waves, 3 per SIMD, <= 168 VGPRs)
#endif
constexpr int RR_THREADS = SE_RR_THREADS;
// instantiations (keys per thread) of the 512-thread build; -DSE_RR_DEV: a quick-to-compile subset for kernel work (NOT a product build)
#ifdef SE_RR_DEV98
#define SE_RR_CASES_512 SE_RR_CASE(98)
#elif defined(SE_RR_DEV)
#define SE_RR_CASES_512 SE_RR_CASE(8) SE_RR_CASE(72) SE_RR_CASE(98)
#else
#define SE_RR_CASES_512 SE_RR_CASE(2) SE_RR_CASE(8) SE_RR_CASE(12) SE_RR_CASE(20) SE_RR_CASE(30) SE_RR_CASE(40) SE_RR_CASE(46) SE_RR_CASE(52) SE_RR_CASE(58) SE_RR_CASE(64) SE_RR_CASE(72) SE_RR_CASE(80) SE_RR_CASE(88) SE_RR_CASE(98) SE_RR_CASE(104)
#endif
constexpr int RR_WAVES = RR_THREADS / WAVE;
constexpr int RR_SCAN_THREADS = 512;                  // threads that scan the packed counters (2 words each of the 1024 per wave)
constexpr int RR_SCAN_WAVES = RR_SCAN_THREADS / WAVE;
static_assert((53248 + RR_THREADS - 1) / RR_THREADS <= 104, "rows of up to 53,248 columns: 104 keys per thread (512 threads) / 70 (768 threads)");
constexpr int RR_G = 8;                               // steps ranked together (latency overlap vs live registers)
constexpr int RR_MAX_N = 53248;

// Wave multisplit on an 8-bit digit: bit mask of the lanes whose digit DIFFERS from this lane's, as
// OR_b (ballot_b ^ own_bit_b): per bit one v_bfe_i32, one v_cmp (the ballot) and one v_bitop3 per mask half.
__device__ __forceinline__ void differ_mask(uint32_t d, uint32_t &lo, uint32_t &hi)
{
    lo = 0;
    hi = 0;
#pragma unroll
    for (int bb = 0; bb < 8; bb += 2) {
        uint32_t s0 = (uint32_t)__builtin_amdgcn_sbfe((int)d, bb, 1);       // 0 or ~0
        uint32_t s1 = (uint32_t)__builtin_amdgcn_sbfe((int)d, bb + 1, 1);
        asm volatile("" : "+v"(s0), "+v"(s1));   // keep the ballot a plain compare of the extracted bit
        const uint64_t m0 = __ballot(s0 != 0), m1 = __ballot(s1 != 0);
        // acc | (ballot ^ own): v_bitop3_b32, truth table of a | (b ^ c) with a = 0xF0, b = 0xCC, c = 0xAA
        lo = __builtin_amdgcn_bitop3_b32(lo, (uint32_t)m0, s0, 0xF6);
        hi = __builtin_amdgcn_bitop3_b32(hi, (uint32_t)(m0 >> 32), s0, 0xF6);
        lo = __builtin_amdgcn_bitop3_b32(lo, (uint32_t)m1, s1, 0xF6);
        hi = __builtin_amdgcn_bitop3_b32(hi, (uint32_t)(m1 >> 32), s1, 0xF6);
    }
}

// Sort key of the register-resident kernel: the canonical order of canon_key() in 6 VALU instead of ~13 (the canonicalisation
// of a 50k row is 98 keys x 2 waves per SIMD: 6k of a row's 115k cycles with canon_key).  Negative values map to ~u + 1, so
// that -0.0 lands ON +0.0's key 0x80000000 without a compare (the keys never leave the kernel; order and ties are those of
// canon_key: ascending value, -0.0 == +0.0, every NaN on ONE key above +inf's 0xFF800000).
// Padding slots (positions >= the row's length) sort behind every real key, NaN included, but NOT on one key: with 0xFFFFFFFF for
// all of them every wave step of padding put its 64 returning adds on ONE counter in every pass -- 113 instead of 6.4 cycles per
// step (a 26,624-column row in the 32,768-slot instantiation took LONGER than a 32,768-column one; NABirds' 24,633 and CUB's 5,794
// test items sit deep inside their instantiations).  A lane's padding key carries the lane number in both lower digits (10 / 11-bit
// splits alike: bits 0-5 and 11-16) and one of the six most significant 12-bit values above NaN's: distinct counters in the lower
// passes, ~11 lanes per counter in the last one.  Their order among themselves does not matter: the output ends at the row's length
// -- but it decides which LDS banks their scatters hit: RR_CANON adds the step number (3 + 3 bits above the two lane fields), so that
// the padding keys of one wave step end up 8 slots apart instead of (padding steps) slots apart (a multiple of 16: 4 banks); the 98-
// and 104-key instantiations leave it out (one more instruction per key: +1 % at 50,000 columns, where there is next to no padding).
constexpr uint32_t RR_KEY_NAN = 0xFF900000u;
__device__ __forceinline__ uint32_t rr_pad_key(uint32_t lane) { return 0xFFA00000u + ((lane % 6u) << 20) + (lane << 11) + lane; }
__device__ __forceinline__ uint32_t rr_key(uint32_t u, bool pad, uint32_t pad_key)
{
    const uint32_t sx = (uint32_t)((int32_t)u >> 31);                                   // 0 or ~0
    uint32_t k = __builtin_amdgcn_bitop3_b32(u, sx, 0x80000000u, 0x1E);                 // u ^ (sx | 0x80000000): ~u or u | 0x80000000
    k -= sx;                                                                            // negative: + 1
    const float f = __uint_as_float(u);
    k = (f != f) ? RR_KEY_NAN : k;
    return pad ? pad_key : k;
}

__device__ __forceinline__ int32_t rr_clamp_i32(int32_t x, int32_t lo, int32_t hi) { return min(max(x, lo), hi); }   // (v_med3_i32)

template <typename T>
__device__ __forceinline__ void opaque(T &x) { asm volatile("" : "+v"(x)); }   // value barrier: no CSE / hoisting across it

__device__ __forceinline__ uint32_t lds_off(const void *p) { return (uint32_t)(uintptr_t)p; }   // LDS byte address of a __shared__ pointer
// The exchange is written with explicit DS instructions: the 16-bit loads land IN PLACE in one half of a
// live register (no temporaries, no merge VALU) and hipcc cannot cache 2 x ITEMS destination addresses.
// hipcc does not count these toward its own s_waitcnt bookkeeping, hence the explicit lds_wait().
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_st16_lo(uint32_t addr, uint32_t v) { asm volatile("ds_write_b16 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st16_hi(uint32_t addr, uint32_t v) { asm volatile("ds_write_b16_d16_hi %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// (gfx950 runs with SRAM-ECC: a d16 load ZEROES the other register half, so the 16-bit loads go to a
// small ring of temporaries and one v_perm_b32 merges each into the live register.)
template <int OFF>
__device__ __forceinline__ void lds_ld16(uint32_t &dst, uint32_t addr) { asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory"); }
// wait until at most K DS operations are outstanding; `landed` is the register the awaited load writes -- naming
// it as an in/out operand is what orders its consumers after the wait (the asm statements carry no other dependency)
template <int K>
__device__ __forceinline__ void lds_wait_le(uint32_t &landed) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(landed) : "n"(K) : "memory"); }

#ifndef SE_RR_RING
#define SE_RR_RING 12
#endif
constexpr int RR_RING = SE_RR_RING;   // 16-bit loads in flight per lane (lgkmcnt counts to 15)
// Software-pipelined read of this lane's ITEMS new 16-bit values (read slot of step s = addr + 128 s)
// into the HIGH (HI = true) or LOW half of a[s]: read i is issued RR_RING - 1 reads ahead of its merge.
template <int ITEMS, bool HI, int I = 0>
struct RRRead {
    static __device__ __forceinline__ void run(uint32_t (&a)[ITEMS], uint32_t (&t)[RR_RING], uint32_t addr)
    {
        constexpr int D = RR_RING - 1;
        if constexpr (I < ITEMS) lds_ld16<I * WAVE * 2>(t[I % RR_RING], addr);
        if constexpr (I >= D) {
            constexpr int J = I - D;
            constexpr int newest = (I < ITEMS ? I : ITEMS - 1);
            lds_wait_le<newest - J>(t[J % RR_RING]);
            // HI: (t << 16) | (a & 0xFFFF)      LO: (a & 0xFFFF0000) | (t & 0xFFFF)
            a[J] = __builtin_amdgcn_perm(t[J % RR_RING], a[J], HI ? 0x05040100u : 0x03020504u);
        }
        if constexpr (I + 1 < ITEMS + D) RRRead<ITEMS, HI, I + 1>::run(a, t, addr);
    }
};

// ---- R phase, one group of V <= RR_G consecutive steps starting at step S0 -------------------------------
// Per step: every lane READS its digit's counter (equal digits: one broadcast read), then the first lane of
// each digit group ADDS the group size (no return value).  DS operations of one wave execute in program
// order, so the read sees exactly the keys of earlier steps; nothing waits on the add, and the V reads of
// the group are in flight together.  Explicit DS instructions: a volatile / atomic C++ access to the
// counters would be emitted as a FLAT operation.
template <int ITEMS, int V, int S0, int I = 0>
struct RRRankFinish {   // read I of the group is followed by 2 (V-1-I) + 1 younger DS operations (every step issues both)
    static __device__ __forceinline__ void run(uint32_t (&ir)[ITEMS], uint32_t (&key)[ITEMS], uint32_t (&rnk)[V], uint32_t (&base)[V])
    {
        lds_wait_le<2 * (V - 1 - I) + 1>(base[I]);
        ir[S0 + I] = (ir[S0 + I] & 0xFFFF0000u) | (base[I] + rnk[I]);
        opaque(ir[S0 + I]);    // materialise now: nothing but key/ir stays live per key
        opaque(key[S0 + I]);   // (and no cached counter address either)
        if constexpr (I + 1 < V) RRRankFinish<ITEMS, V, S0, I + 1>::run(ir, key, rnk, base);
    }
};

template <int ITEMS, int S0 = 0>
struct RRRank {
    static __device__ __forceinline__ void run(uint32_t (&ir)[ITEMS], uint32_t (&key)[ITEMS], int shift, int lane, uint32_t cb)
    {
        constexpr int V = (ITEMS - S0 < RR_G) ? (ITEMS - S0) : RR_G;
        static_assert(2 * V - 1 <= 15, "lgkmcnt counts to 15");
        uint32_t rnk[V], base[V];
#pragma unroll
        for (int g = 0; g < V; g++) {
            const uint32_t d = (key[S0 + g] >> shift) & 0xFFu;
            uint32_t dlo, dhi;
            differ_mask(d, dlo, dhi);
            // lower lanes with the same digit = lower lanes - lower lanes that differ
            rnk[g] = (uint32_t)lane - __builtin_amdgcn_mbcnt_hi(dhi, __builtin_amdgcn_mbcnt_lo(dlo, 0u));
            const uint32_t ca = cb + (d << 2);
            asm volatile("ds_read_b32 %0, %1" : "=v"(base[g]) : "v"(ca) : "memory");
            if (rnk[g] == 0) {   // (lane 0 always leads a group, so the add is issued in every step)
                const uint32_t np = 64u - (uint32_t)(__popc(dlo) + __popc(dhi));
                asm volatile("ds_add_u32 %0, %1" ::"v"(ca), "v"(np) : "memory");
            }
        }
        RRRankFinish<ITEMS, V, S0>::run(ir, key, rnk, base);
        // the scheduling fence between groups keeps hipcc from hoisting all ITEMS steps' ballots at once
        // (hundreds of live SGPR pairs -> spills)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S0 + V < ITEMS) RRRank<ITEMS, S0 + V>::run(ir, key, shift, lane, cb);
    }
};

// ---- R phase, hardware-ordered variant -------------------------------------------------------------------
// One RETURNING add per key on the wave's digit counter: the returned value is the within-wave rank directly,
// PROVIDED the LDS serves the lanes of one ds_add_rtn that hit the same address in ascending lane order.
// gfx950 does (tools/probes/lds_atomic_order.hip: 84 M returns under every conflict pattern, 8 waves per
// workgroup hammering the LDS), but the ISA does not promise it, so this variant is only selected after the
// same property has been re-verified on the device at first use (se_rank_rows: probe kernel) and can be
// switched off with SE_RANK_SAFE=1.  ~5 VALU per key and pass instead of ~41.
#ifndef SE_RR_NT
#define SE_RR_NT 1   // 1: nontemporal rank stores (round 6: -2 % on the image path once the next row is loaded straight from HBM; neutral elsewhere)
#endif
typedef int rr_i32x4 __attribute__((ext_vector_type(4)));
typedef long long rr_i64x2 __attribute__((ext_vector_type(2)));
#ifndef SE_RR_PF
#define SE_RR_PF 1
#endif
#ifndef SE_RR_GH
#define SE_RR_GH 8
#endif
constexpr int RR_GH = SE_RR_GH;    // returning adds in flight per lane
constexpr int RR_HW_BITS = 11;   // digit width of the hardware-ordered variant: 3 passes (11 + 11 + 10 bits; rows of >= 32,768 columns: 10 + 10 + 12) instead of 4
constexpr int RR_WIDE_WORDS = 2048;   // counter words per wave of the 12-bit pass (4096 packed 16-bit counters)
#ifndef SE_RR_TWO
#define SE_RR_TWO 1                   // build parameter: 0 = never take the two-pass path below
#endif
#ifndef SE_RR_PROF_BARRIER
#define SE_RR_PROF_BARRIER 0          // profile build: 1 = a barrier in front of every timestamp (phase times include the skew between the waves)
#endif
#ifndef SE_RR_EARLY
#define SE_RR_EARLY 1                 // build parameter: image path -- next row's loads issued behind the last pass, waited for before the rank stores
#endif
#ifndef SE_RR_SCAN_DEPTH
#define SE_RR_SCAN_DEPTH 1            // build parameter: groups of the image path's tag scan whose random reads are in flight ahead of the arithmetic
#endif
#ifndef SE_RR_WO
#define SE_RR_WO 4                    // build parameter: steps of the int32 write-out loop whose LDS reads are in flight together
#endif
#ifndef SE_RR_IMG
#define SE_RR_IMG 1                   // build parameter: 0 = never take the image path (two passes on a 24-bit image + repair)
#endif
// Two-pass path of the long-row kernel: when all keys of a row but at most RR_TWO_OUT lie within RR_TWO_SPAN codes below its largest
// key (Euclidean-distance rows of one data set do: the outliers are the query's own distance and its near-duplicates; the reference's
// cosine rows -- -dot, both signs around zero, ~2^31 codes wide -- do not), the row is
// sorted on (key - lo + 1) << 8 -- 24 significant bits -- in TWO passes of 12 bits, and the few keys below the window, which that
// mapping sends to 0 (first places, index order), are put into their true order afterwards (rank by counting, one thread each).  7 instead of 12 random LDS
// operations per key.  Rows that do not qualify take the three passes.
constexpr uint32_t RR_TWO_SPAN = (1u << 24) - 3u;
constexpr int RR_TWO_OUT = 256;
// Image path (VAR 3; round 5): rows whose keys do NOT fit a 2^24-code window -- the reference's cosine rows, -dot of both signs -- are
// sorted on a monotone 24-bit IMAGE of the key, q = bits(fl(v + c)) - bits(c / 2) with c = the power of two that puts every |v| of
// the row below 0.75 c (fl(v + c) lies in [c / 4, 7 c / 4]: rounding is monotone non-decreasing, so the image order never contradicts
// the key order; below c / 2 the subtraction saturates at 0), again in two 12-bit passes -- and then REPAIRED: keys that share an
// image (~1.3 % of a 50,000-column cosine row: the image's ulp is c 2^-24 below c and c 2^-23 above it) sit next to each other in index
// order after the stable sort, and only their true (key, index) order can differ.  To SEE equal images at the final positions the low 8
// image bits are written as a one-byte TAG per COLUMN before the passes (tag plane, column order: consecutive lanes, consecutive
// bytes -- a linear write under the image arithmetic); after the last pass the tag of final position i is tagb[xbuf[i]]: a scan reads
// the indices in 16-byte groups, gathers their tags (one random byte read per key: 8 instead of 12 random LDS operations per key in
// all) and finds the neighbours with equal tags (every block of equal images is inside such a run; 1 / 256 of the other neighbours
// are false positives, which the repair leaves in place because their keys are in order); the pairs go to a worklist, and the owner
// of a run's first pair gathers the true keys from the row (L2 / Infinity Cache) and puts the run into canonical order.  (Tags
// scattered to the FINAL positions instead -- through a third exchange in round 5, from a bucket-start bitmap in round 6 -- were
// built twice and lost both times: tools/experiments/README.md.)  A row with a run above RR_IMG_RUN entries, a full worklist, or keys the image cannot take (NaN, infinities, all zero) is
// sorted again by the same workgroup one LEVEL further down: level 0 = the tight image (c half as large: the bulk of a cosine row
// spreads over twice as many most significant digits -- fewer lanes of a wave step on one counter, fewer 2-byte scatters into the
// same dword -- and half as many keys share an image; the keys beyond its range saturate at the ends: 8.0 -> 7.75 ms), level 1 = the
// image whose range holds every key of the row (keys below -c / 2 still saturate at image 0 and are repaired like any run: time, not
// correctness), level 2 = the three passes; the workgroup backs off from a level it had to give up for a while.
constexpr int RR_IMG_RUN = 8;          // longest run the repair sorts (entries)
constexpr int RR_IMG_WL = 3072;        // worklist entries (run start | length << 16)
constexpr int RR_IMG_MAX_ITEMS = 98;   // instantiations above this have no room for tags + worklist next to the exchange buffer
constexpr uint32_t RR_IMG_NAN_PADHI = 0xFF0u;   // most significant image digit of the padding slots (real images end at 0xE00)
// Counters are 16 bits wide, two per LDS word (8 waves x 2048 digits x 2 B = 32 KB next to the 100 KB exchange
// buffer): a wave holds at most 64 x 104 keys and a destination is < 53,248, so neither half can carry into the
// other.  The returning add is done on the word with the increment shifted into the digit's half.
// PEEL (used for the most significant digit): the lanes that share lane 0's digit are ranked by one ballot and ONE
// add of the group size issued by lane 0; only the other lanes issue their own returning add.  The top digit of
// real distance rows is heavily skewed -- all-positive Euclidean distances put every key of a wave step on one
// counter, i.e. a 64-way same-address conflict per instruction (16.9 ms instead of 11 ms on the CLI-default branch).
// Digit d of a pass = key bits [shift, shift + wlo + whi): its low wlo = 10 bits select the counter WORD, bit 10 (whi = 1;
// the 10-bit last pass has whi = 0) the 16-bit HALF -- both straight v_bfe_u32 of the key, and the S phase can scan the
// packed words without unpacking them (digits 0..1023 are the low halves, 1024..2047 the high halves).
// The adds are software-pipelined one by one -- "retire step S - RR_GH, issue step S" -- so every lane keeps RR_GH - 1 or
// RR_GH returning adds in flight from the first step to the last (tools/probes/lds_throughput.hip: the LDS sustains one
// random-word returning add per 6.4 cycles and CU; batches of 8 that drain before the next batch is issued reached 21).
template <int ITEMS, bool PEEL, int S = 0>
struct RRRankHW {
    // `peel` is wave-uniform (one code instance for all passes: two instances of this unrolled body make hipcc spill)
    static __device__ __forceinline__ void run(uint32_t (&ir)[ITEMS], uint32_t (&key)[ITEMS], uint32_t (&r)[RR_GH], uint32_t (&sh)[RR_GH],
                                               uint32_t (&grp)[RR_GH], uint32_t shift, uint32_t hshift, uint32_t wlo, uint32_t whi, uint32_t cb,
                                               int lane, bool peel)
    {
        constexpr int SLOT = S % RR_GH;
        uint32_t ca = 0, inc = 0, shv = 0, g = 0xFFFFFFFFu;
        uint64_t part = ~0ull;
        if constexpr (S < ITEMS) {
            const uint32_t w = __builtin_amdgcn_ubfe(key[S], shift, wlo);
            const uint32_t h = __builtin_amdgcn_ubfe(key[S], hshift, whi);   // width 0 -> 0
            ca = cb + (w << 2);
            if constexpr (PEEL) {
                shv = h << 4;
                inc = 1u << shv;
            } else {
                // no shift amounts: the increment is 1 + 0xFFFF h, and `shv` holds the v_perm_b32 selector that drops the returned
                // half straight into the low half of ir (bytes [ir3, ir2, r(1 + 2h), r(2h)]): 6 VALU per key instead of 7
                inc = __umul24(h, 0xFFFFu) + 1u;
                shv = __umul24(h, 0x0202u) + 0x07060100u;
            }
            if constexpr (PEEL) {
                if (peel) {                          // wave-uniform branch: the other passes skip the group bookkeeping
                    // lanes sharing lane 0's digit (most significant pass only) are ranked by one ballot and ONE add of the group size
                    const uint32_t d = w | (h << 11);   // (w < 2048)
                    const bool in = (d == (uint32_t)__builtin_amdgcn_readfirstlane((int)d));
                    const uint64_t m = __ballot(in);
                    if (in) g = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (lane == 0) inc = (uint32_t)__popcll(m) << shv;
                    part = ~m | 1ull;                // only lane 0 and the lanes outside its group issue the add
                }
            }
        }
        if constexpr (S >= RR_GH) {                  // retire step J: wait until only the adds issued after it are outstanding
            constexpr int J = S - RR_GH;
            constexpr int younger = ((S < ITEMS ? S : ITEMS) - 1) - J;
            lds_wait_le<younger>(r[SLOT]);
            if constexpr (PEEL) {
                const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)r[SLOT]);   // lane 0's counter word before its add
                const uint32_t own = (grp[SLOT] != 0xFFFFFFFFu) ? lead : r[SLOT];                 // group members use lane 0's word (same digit => same half)
                const uint32_t rank = __builtin_amdgcn_ubfe(own, sh[SLOT], 16u) + (grp[SLOT] != 0xFFFFFFFFu ? grp[SLOT] : 0u);
                ir[J] = (ir[J] & 0xFFFF0000u) | rank;
            } else {
                ir[J] = __builtin_amdgcn_perm(ir[J], r[SLOT], sh[SLOT]);
            }
            opaque(ir[J]);    // materialise now: nothing but key/ir stays live per key
            opaque(key[J]);   // (and no cached counter address either)
        }
        if constexpr (S < ITEMS) {
            sh[SLOT] = shv;
            grp[SLOT] = g;
            if constexpr (PEEL) {
                // The exec mask is narrowed INSIDE the asm statement: a C++ `if` around an asynchronous DS return lets hipcc copy the
                // (not yet landed) result register at the join.  Lane 0 always takes part: exactly one DS operation per step.
                uint64_t saved;
                r[SLOT] = 0;
                asm volatile("s_and_saveexec_b64 %0, %4\n\tds_add_rtn_u32 %1, %2, %3\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved), "+v"(r[SLOT])
                             : "v"(ca), "v"(inc), "s"(part)
                             : "memory");
            } else {
                asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(r[SLOT]) : "v"(ca), "v"(inc) : "memory");
            }
        }
        // the scheduling fence keeps hipcc from hoisting the address arithmetic of all ITEMS steps (live registers -> spills)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S + 1 < ITEMS + RR_GH) RRRankHW<ITEMS, PEEL, S + 1>::run(ir, key, r, sh, grp, shift, hshift, wlo, whi, cb, lane, peel);
    }
};

// LDS bytes in front of wave_tot / the exchange buffer: the dedicated per-wave counters, or (image path) the tag plane + worklist,
// whichever is larger (the two are never live together: the image path's passes both use the counters aliased onto the exchange buffer)
template <int ITEMS, bool HWORD, int VAR>
constexpr size_t rr_region0_bytes()
{
    const size_t cnt = (size_t)RR_WAVES * (HWORD ? (size_t)(1 << RR_HW_BITS) / 2 : (size_t)RK_NB) * sizeof(uint32_t);
    const size_t img = (size_t)RR_THREADS * ITEMS + 32 + (size_t)RR_IMG_WL * sizeof(uint32_t);   // tags (+ 32 B: the scan reads one group ahead), worklist
    return (VAR == 3 && img > cnt) ? img : cnt;
}

// SEG (rows of more than RR_MAX_N columns, see rank_runs below): the kernel sorts SEGMENTS of rows -- "row" v of the loop is segment
// v & (2^shift - 1) of matrix row v >> shift -- and instead of ranks it leaves each segment as a sorted RUN in three 16-bit planes
// (segment-local index, key bits 16-31, key bits 0-15; 512 x ITEMS entries each, the padding sorted last) for the merge kernel.
struct RankSeg {
    int shift;            // log2(segments per row)
    int seg_n;            // columns per segment (a row's last segment may be shorter)
    uint16_t *planes;     // [3][virtual rows][512 x ITEMS]
    int64_t plane_elems;  // virtual rows x 512 x ITEMS
};

template <int ITEMS, bool PROF, bool HWORD, int VAR, bool SEG = false>   // VAR: 0 plain, 1 group-peeling rank phase of the last pass, 2 two-pass path for rows that qualify, 3 two passes on a 24-bit image + repair
__global__ __launch_bounds__(RR_THREADS, RR_THREADS / 256) void rank_rows_reg_kernel(const float *__restrict__ pdist, int64_t ldp, int64_t Q,
                                                                    int N, void *rank, int64_t ldr, int idx64, int vec_ok,
                                                                    unsigned long long *prof, const uint32_t *skew_flag, const RankSeg seg)
{
    static_assert(!SEG || (HWORD && VAR == 0 && !PROF), "segment runs: plain hardware-ordered build only");
    // one launch per variant when the detector is used: the variants that do not match its flag leave at once
    if (skew_flag && *skew_flag != (uint32_t)VAR) return;
    constexpr bool PEEL = VAR == 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char rr_raw[];
    constexpr int BITS = HWORD ? RR_HW_BITS : 8;                        // digit width (narrow passes)
    constexpr int NB = 1 << BITS;
    constexpr int NPASS = (32 + BITS - 1) / BITS;                       // 3 or 4
    constexpr int CNT_WORDS = HWORD ? NB / 2 : NB;                      // LDS words per wave: packed 16-bit or 32-bit counters
    // WIDE: digits of 10 + 10 + 12 bits instead of 11 + 11 + 10.  The most significant digit of real distance rows is skewed (sign,
    // exponent, 1-2 mantissa bits: ~20 values, 8 lanes of a wave step on the same counter = 29 instead of 8 cycles per returning add);
    // two more mantissa bits quarter the multiplicity.  Its 8 x 4096 16-bit counters (64 KB) do not fit next to the exchange buffer, so
    // they ALIAS it: the buffer is idle from the last exchange read of the previous pass to the first scatter of this one (two extra
    // barriers per row keep the other waves' reads / counter look-ups on the right side of that reuse).
    constexpr bool WIDE = HWORD && ((size_t)RR_THREADS * ITEMS * sizeof(uint16_t) >= (size_t)RR_WAVES * RR_WIDE_WORDS * sizeof(uint32_t));
    constexpr bool IMG = WIDE && VAR == 3;                              // image path (see RR_IMG_*)
    constexpr bool RAWKEYS = IMG || (WIDE && VAR == 2);                 // the row loop carries RAW keys: each row picks its sort key itself
    static_assert(VAR != 3 || (WIDE && ITEMS <= RR_IMG_MAX_ITEMS && !SEG), "image path: long-row instantiations with room for the tag plane");
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(rr_raw);              // [RR_WAVES][CNT_WORDS]
    uint32_t *wave_tot = reinterpret_cast<uint32_t *>(rr_raw + rr_region0_bytes<ITEMS, HWORD, VAR>());   // [8] (+pad)
    uint16_t *xbuf = reinterpret_cast<uint16_t *>(wave_tot + 32);       // [RR_THREADS * ITEMS]
    [[maybe_unused]] uint8_t *tagb = rr_raw;                            // image path: one tag byte per final position (aliases the idle dedicated counters)
    [[maybe_unused]] uint32_t *wlist = reinterpret_cast<uint32_t *>(rr_raw + (size_t)RR_THREADS * ITEMS + 32);   // image path: [RR_IMG_WL]
    [[maybe_unused]] uint32_t *ictl = wave_tot + 24;                    // image path: [0] worklist cursor, [1] row given up
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    [[maybe_unused]] const int wave_s = __builtin_amdgcn_readfirstlane(wave);   // the wave's number as a scalar
    const int wpos0 = wave * (ITEMS * WAVE) + lane;                     // position of (step s) = wpos0 + 64 s
    const uint32_t xb = lds_off(xbuf);                                  // exchange buffer, byte address
    const uint32_t rb = xb + 2u * (uint32_t)wpos0;                      // this lane's read slot of step 0
#define RR_DST(IR) (xb + (((IR) & 0xFFFFu) << 1))
    // first distance and length of loop row v (SEG: a segment of matrix row v >> shift)
    auto row_ptr = [&](int64_t v) -> const float * {
        if constexpr (SEG) return pdist + (v >> seg.shift) * ldp + (v & ((1 << seg.shift) - 1)) * (int64_t)seg.seg_n;
        else return pdist + v * ldp;
    };
    auto row_len = [&](int64_t v) -> int {
        if constexpr (SEG) {
            const int rest = N - (int)(v & ((1 << seg.shift) - 1)) * seg.seg_n;
            return rest < seg.seg_n ? rest : seg.seg_n;
        } else return N;
    };
    // tuning aid (SE_RR_PROFILE=1): shader-clock cycles per phase, summed over every workgroup's wave 0
    // (round 6: twelve 32-bit counters and no barrier of their own in front of the timestamps -- every phase but the rank phase and the
    // key read-back ends in one anyway.  The round-5 form -- 36 64-bit counters with a run-time index, a barrier per timestamp -- was
    // 74 KB of code with 208 B of scratch: above the 64 KB instruction cache, it overstated every phase.)
    uint32_t t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = PROF ? (uint32_t)__builtin_amdgcn_s_memtime() : 0;
#define RR_T(i) if constexpr (PROF) { if (SE_RR_PROF_BARRIER) { lds_wait(); wg_barrier(); } const uint32_t now = (uint32_t)__builtin_amdgcn_s_memtime(); t_acc[i] += now - t_last; t_last = now; } else { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

    // The row loop is software-pipelined over HBM: the NEXT row is prefetched into L2 during the last pass (one dword per 128-byte
    // line), loaded into the key registers right after it -- BEFORE this row's rank stores are issued, so the memory pipeline serves
    // the loads first -- and canonicalised after the write-out, which covers most of their latency.  (With the loads issued after
    // the stores, as in the first version, every row first waited for HBM to absorb its 200-400 KB: 29 % of the kernel.)
    uint32_t key[ITEMS], ir[ITEMS];   // ir = (index << 16) | (within-wave rank, then destination)
    __amdgpu_buffer_rsrc_t ld_rsrc;
    uint32_t ld_voff = 0;
    uint32_t ring[RR_RING];
    // One BUFFER load per key: the row is its own buffer resource (base = first distance, extent = its length), the lane's byte offset
    // is one register per 16 steps (4,096 bytes: what the instruction's immediate offset spans) and positions behind the row's end read
    // as ZERO by the hardware's range check (the padding is applied in RR_CANON / by the two-pass paths) -- 8 bytes of code per key.
    // (round 6: the global-load form -- clamped index, 64-bit address arithmetic, 32 bytes of code per key and load site -- put the
    // 98-key kernels 2-4 KB above the 64 KB instruction cache; an unlucky placement of the code object then cost 4-10 % of the kernel.
    // The range check covers voffset + immediate only, so nothing goes into the scalar offset.)
#define RR_LOAD_ONE(DROW, WPOS, S, NN)                                                                               \
    {                                                                                                                 \
        if ((S) % 16 == 0) {                                                                                          \
            if ((S) == 0) {                                                                                           \
                ld_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(DROW), 0, (NN) * 4, 0x00020000);     \
                ld_voff = (uint32_t)(WPOS) * 4u;                                                                      \
            } else ld_voff += 4096u;                                                                                  \
            opaque(ld_voff);                                                                                          \
        }                                                                                                             \
        key[S] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ld_rsrc, (int)ld_voff + ((S) % 16) * 256, 0, 0);      \
    }
#define RR_CANON(NN)                                                                                                  \
    {                                                                                                                 \
        int wpos_ = wpos0;                                                                                            \
        opaque(wpos_);                                                                                                \
        const uint32_t pk_ = rr_pad_key((uint32_t)wpos_ & 63u);                                                       \
        _Pragma("unroll") for (int s = 0; s < ITEMS; s++) {                                                           \
            const int pos = wpos_ + s * WAVE;                                                                         \
            key[s] = rr_key(key[s], pos >= (NN), pk_ + (ITEMS <= 88 ? (uint32_t)((((s >> 3) & 7) << 17) | ((s & 7) << 6)) : 0u)); /* pos >= N: padding */ \
        }                                                                                                             \
    }
#define RR_PREFETCH_NEXT_ROW() RR_PREFETCH_NEXT_ROW_IF(end >= 32 && more)
#define RR_PREFETCH_NEXT_ROW_IF(COND) \
    if (COND) { \
                const char *nrow = (const char *)row_ptr(row + gridDim.x); \
                const uint32_t row_bytes = (uint32_t)row_len(row + gridDim.x) * 4u; \
                for (uint32_t off = (uint32_t)tid * 128u; off < row_bytes; off += RR_THREADS * 128u) \
                    asm volatile("global_load_dword %0, %1, off" : "=v"(pf_sink) : "v"(nrow + off) : "memory"); \
            }
    // SEG: the exchange buffer is one 16-bit plane of the sorted run -- 16 bytes per lane and step, straight from LDS to the plane
#define RR_STREAM_PLANE(P, ROW)                                                                                       \
    {                                                                                                                 \
        uint16_t *pl = seg.planes + (P) * seg.plane_elems + (ROW) * (int64_t)(RR_THREADS * ITEMS);                    \
        int j = tid * 8;                                                                                              \
        opaque(j);                                                                                                    \
        _Pragma("unroll 2") for (; j < RR_THREADS * ITEMS; j += RR_THREADS * 8)                                       \
            *reinterpret_cast<uint4 *>(pl + j) = *reinterpret_cast<const uint4 *>(xbuf + j);                          \
    }
    // Padding fix-ups of the two-pass paths run on the wave(s) that hold slots behind the row's end -- and the whole workgroup waits
    // for them at the next barrier.  A row that fills its instantiation to within RR_PADS wave steps (50,000 columns in the 98-key one:
    // 3 steps) only has such slots in its last RR_PADS steps: those alone are visited (round 6: 98 x 2-3 VALU on one wave per phase).
    constexpr int RR_PADS = ITEMS < 8 ? ITEMS : 8;
#define RR_PAD_LOOP(NROW, BODY)                                                                                       \
    {                                                                                                                 \
        const int first_pad_ = ((NROW) - wave_s * (ITEMS * WAVE)) >> 6;   /* steps below it hold row positions only (negative: all padding) */ \
        if (first_pad_ >= ITEMS - RR_PADS) {                                                                          \
            _Pragma("unroll") for (int s = ITEMS - RR_PADS; s < ITEMS; s++) { BODY }                                  \
        } else {                                                                                                      \
            _Pragma("unroll") for (int s = 0; s < ITEMS; s++) { BODY }                                                \
        }                                                                                                             \
    }
    if ((int64_t)blockIdx.x < Q) {
        const float *drow = row_ptr(blockIdx.x);
        const int n0 = row_len(blockIdx.x);
        int wpos = wpos0;
        opaque(wpos);
#pragma unroll
        for (int s = 0; s < ITEMS; s++) RR_LOAD_ONE(drow, wpos, s, n0)
        if constexpr (!RAWKEYS) RR_CANON(n0)   // (image / window paths: the keys stay raw until the row has chosen between its two-pass key and the canonical key)
    }
    uint32_t pf_sink = 0;
    // image path, all wave-uniform.  A row is tried at LEVEL 0 = the tight image (c one power of two smaller: the row's bulk spreads over twice
    // as many most significant digits and half as many keys share an image; the few keys beyond its range -- a cosine row's own
    // distance of -1 -- saturate at the ends and are put right by the repair like any other run), given up -> the same row again at
    // level 1 = the image that holds every key, given up -> level 2 = the three passes.  After a give-up at a level the workgroup
    // starts its next 1, 2, 4 ... 32 rows one level further down before it tries that level again.
    [[maybe_unused]] int img_level = 0;                       // level of the row in hand
    [[maybe_unused]] bool img_again = false;                  // the row in hand is a retry (its level was set by the give-up)
    [[maybe_unused]] int img_skip0 = 0, img_pen0 = 0, img_skip1 = 0, img_pen1 = 0;   // per level: rows still to skip it, length of its last back-off
    for (int64_t row = blockIdx.x; row < Q;) {
        const bool more = row + gridDim.x < Q;
        [[maybe_unused]] const int n_next = row_len(more ? row + gridDim.x : row);
        // a wave whose slots are ALL padding (rows well below the instantiation's capacity) sits the passes out: its counters stay zero,
        // its slots of the exchange buffer are never read by anyone, and it only joins the barriers and the counter scan.  (The big
        // instantiations keep the unconditional code: a branch around their unrolled phases makes hipcc copy the key registers.)
        constexpr bool SKIP_WAVES = HWORD && ITEMS <= 88;
        const bool wave_live = !SKIP_WAVES || wave * (ITEMS * WAVE) < row_len(row);
        // ---- does the row qualify for the two-pass path?  (uniform per row; before the index registers exist: only the keys are live) ----
        bool two = false;
        [[maybe_unused]] int n_out = 0;
        if constexpr (IMG) {
            const int n_row = row_len(row);
            if (!img_again) {                                   // a new row: the first level nothing has been given up at lately
                if (img_skip1 > 0) { img_level = 2; img_skip1--; }
                else if (img_skip0 > 0) { img_level = 1; img_skip0--; }
                else img_level = 0;
            }
            bool attempt = img_level < 2;
            uint32_t cbits = 0, base = 0;
            if (attempt) {
                // largest magnitude of the row as an integer maximum (NaN and infinities come out on top and disqualify the row)
                uint32_t mb = 0;
#pragma unroll
                for (int s = 0; s + 1 < ITEMS; s += 2) mb = max(max(mb, key[s] & 0x7FFFFFFFu), key[s + 1] & 0x7FFFFFFFu);   // v_max3_u32
                if constexpr (ITEMS & 1) mb = max(mb, key[ITEMS - 1] & 0x7FFFFFFFu);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) mb = max(mb, (uint32_t)__shfl_xor((int)mb, off, 64));
                if (lane == 0) wave_tot[wave] = mb;
                wg_barrier();
                // (behind the barrier: every wave has read the previous row's worklist cursor / give-up flag by now; the next use of either
                // word lies behind the barriers of the passes)
                if (tid == 0) { ictl[0] = 0; ictl[1] = 0; }
#pragma unroll
                for (int w = 0; w < RR_WAVES; w++) mb = max(mb, wave_tot[w]);
                mb = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb);
                // c = 2^(e + 1) with 2^e * 1.5 >= max |v|: every |v| <= 0.75 c.  Finite, not tiny (2^-100 <= max |v| < 2^126).
                const uint32_t e = (mb >> 23) + ((mb & 0x7FFFFFu) > 0x400000u ? 1u : 0u) - (img_level == 0 ? 1u : 0u);
                two = mb >= 0x0D800000u && mb < 0x7E800000u;
                cbits = (e + 1u) << 23;
                base = e << 23;
            }
            if (two) {
                const float c = __uint_as_float(cbits);
                int wpos_ = wpos0;
                opaque(wpos_);
                // padding slots: above every real image, spread over 16 most significant digits and (pass 0) one counter per lane
                const uint32_t pk_ = ((RR_IMG_NAN_PADHI + ((uint32_t)wpos_ & 15u)) << 20) | (((uint32_t)wpos_ & 63u) << 14);
                const uint32_t tb = lds_off(tagb) + (uint32_t)wpos_;            // tag plane, this lane's slot of step 0
#pragma unroll
                for (int s = 0; s < ITEMS; s++) {
                    const uint32_t t = __float_as_uint(__uint_as_float(key[s]) + c);
                    // (signed arithmetic: at the tight level fl(v + c) can be negative or reach 2 c -- both ends saturate, the map stays monotone)
                    const uint32_t q = (uint32_t)rr_clamp_i32(__builtin_elementwise_sub_sat((int32_t)t, (int32_t)base), 0, 0xFEFFFF);
                    // tag of column (wpos + 64 s) = the low 8 image bits, written IN COLUMN ORDER: consecutive lanes, consecutive bytes
                    asm volatile("ds_write_b8 %0, %1 offset:%2" ::"v"(tb), "v"(q), "n"(s * WAVE) : "memory");
                    key[s] = q << 8;
                }
                if (wave_s * (ITEMS * WAVE) + ITEMS * WAVE > n_row) {   // wave-uniform: only the wave(s) that hold padding slots
                    RR_PAD_LOOP(n_row, key[s] = (wpos_ + s * WAVE >= n_row) ? pk_ : key[s];)
                }
            } else {
                RR_CANON(n_row)
            }
            RR_T(8)
        }
        if constexpr (WIDE && VAR == 2) {
            // (round 5: the keys are still RAW float bits here -- the row is canonicalised only if it does not qualify.  For the rows this
            // path is for -- non-negative distances with a handful of keys below the window -- the order of the raw bits as SIGNED integers
            // is the canonical order above zero: negative values (the query's own distance can round to -1e-7) are negative integers, i.e.
            // "below the window" like every other small key; a row that holds a NaN of either sign does not qualify (see `umx` below).
            // 1.5 + 2 + 3 VALU per key instead of the 6 of the canonicalisation + 9: ~7k of a row's ~97k cycles.)
            uint32_t *stat = wave_tot;                                  // [0, 8): per-wave maxima, [8, 16): per-wave counts below the window
            uint2 *outl = reinterpret_cast<uint2 *>(wcnt);              // (key, position) of the keys below the window (the dedicated counters are idle on this path)
            const int n_row = row_len(row);
            // largest key, NaN read as +inf (a window anchored at +inf holds nothing else: the row falls back), negatives as 0
            int32_t mxi = 0;
            uint32_t umx = 0;   // largest raw key as an UNSIGNED integer: above -inf's 0xFF800000 only for a NaN with the sign bit set
#pragma unroll
            for (int s = 0; s + 1 < ITEMS; s += 2) {
                mxi = max(max(mxi, rr_clamp_i32((int32_t)key[s], 0, 0x7F800000)), rr_clamp_i32((int32_t)key[s + 1], 0, 0x7F800000));
                umx = max(max(umx, key[s]), key[s + 1]);   // v_max3_u32
            }
            if constexpr (ITEMS & 1) { mxi = max(mxi, rr_clamp_i32((int32_t)key[ITEMS - 1], 0, 0x7F800000)); umx = max(umx, key[ITEMS - 1]); }
            // a negative NaN (0xFFC00000: what 0 / 0 gives on x86) is a negative integer: it would count as "below the window" and be ranked
            // FIRST.  Read it as +inf like the positive ones: the window then holds nothing else and the row takes the canonical three passes.
            mxi = umx > 0xFF800000u ? 0x7F800000 : mxi;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mxi = max(mxi, __shfl_xor(mxi, off, 64));
            if (lane == 0) stat[wave] = (uint32_t)mxi;
            if (tid == 0) stat[2 * RR_WAVES] = 0;                        // cursor of the list
            RR_T(9)
            wg_barrier();
            RR_T(10)
#pragma unroll
            for (int w = 0; w < RR_WAVES; w++) mxi = max(mxi, (int32_t)stat[w]);
            const int32_t kmax = __builtin_amdgcn_readfirstlane(mxi);
            // window [lo, kmax] in raw-bit space; lo >= 1, so lo - 1 exists.  A key is "below" when (int32) raw < lo.
            const int32_t lo = kmax > (int32_t)RR_TWO_SPAN ? kmax - (int32_t)RR_TWO_SPAN : 1, lo1 = lo - 1;
            uint32_t below_l = 0;                                        // this lane's keys below the window (padding slots excluded below)
            {
                int wpos = wpos0;
                opaque(wpos);
#pragma unroll
                for (int s = 0; s < ITEMS; s++) below_l += ((int32_t)key[s] < lo) ? 1u : 0u;        // v_cmp + v_addc
                // padding slots hold a copy of the row's last element: take them out again (only the wave(s) that have any)
                // (`lo_b`, `lo_c` below: opaque copies -- hipcc otherwise shares the 98 compare masks between the three loops and spills the
                // SGPR pairs to VGPR lanes, ~600 v_writelane / v_readlane per row)
                if (wave_s * (ITEMS * WAVE) + ITEMS * WAVE > n_row) {
                    int32_t lo_b = lo;
                    opaque(lo_b);
                    RR_PAD_LOOP(n_row, below_l -= (wpos + s * WAVE >= n_row && (int32_t)key[s] < lo_b) ? 1u : 0u;)
                }
            }
            uint32_t below_w = below_l;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) below_w += (uint32_t)__shfl_xor((int)below_w, off, 64);
            if (lane == 0) stat[RR_WAVES + wave] = below_w;
            RR_T(11)
            wg_barrier();
            uint32_t below = 0;
#pragma unroll
            for (int w = 0; w < RR_WAVES; w++) below += stat[RR_WAVES + w];
            two = kmax > 0 && below <= (uint32_t)RR_TWO_OUT;
            n_out = (int)below;
            if (two) {
                int wpos = wpos0;
                opaque(wpos);
                if (below_w != 0) {   // wave-uniform and rare: only the waves that hold such keys walk theirs; the list takes CANONICAL keys
                    int32_t lo_c = lo;
                    opaque(lo_c);
#pragma unroll
                    for (int s = 0; s < ITEMS; s++)
                        if ((int32_t)key[s] < lo_c && wpos + s * WAVE < n_row)
                            outl[atomicAdd(&stat[2 * RR_WAVES], 1u)] = make_uint2(rr_key(key[s], false, 0u), (uint32_t)(wpos + s * WAVE));
                }
                // key' = clamp(raw - (lo - 1), 0, 0xFFFFFF) << 8: 0 below the window, 1 .. 2^24 - 2 inside it, 0xFFFFFF for NaN (and the padding)
#pragma unroll
                for (int s = 0; s < ITEMS; s++)
                    key[s] = (uint32_t)rr_clamp_i32(__builtin_elementwise_sub_sat((int32_t)key[s], lo1), 0, 0xFFFFFF) << 8;
                if (wave_s * (ITEMS * WAVE) + ITEMS * WAVE > n_row) {
                    RR_PAD_LOOP(n_row, key[s] = (wpos + s * WAVE >= n_row) ? 0xFFFFFF00u : key[s];)
                }
            } else {
                RR_CANON(n_row)
            }
            // (the first barrier of pass 0 orders these LDS accesses before anything that follows)
            RR_T(8)
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            int wpos = wpos0;
            opaque(wpos);
#pragma unroll
            for (int s = 0; s < ITEMS; s++) ir[s] = (uint32_t)(wpos + s * WAVE) << 16;
        }
#pragma unroll 1
        for (int p = 0; p < NPASS; p++) {
            const int shift = two ? 8 + 12 * p : (WIDE ? p * 10 : p * BITS);
            const int end = two ? (p == 1 ? 32 : 20)                        // bits [0, end) are sorted after this pass
                                : (WIDE ? (p == 2 ? 32 : shift + 10) : ((shift + BITS < 32) ? shift + BITS : 32));
            const bool wide = WIDE && (two || p == 2);                      // 12-bit digit, counters aliased onto the exchange buffer
            // ---- R: stable rank inside the wave ----
            // counters of this pass: the wave's slice of the dedicated region, or (wide pass) of the idle exchange buffer
            uint32_t *pcnt = wcnt;              // [RR_WAVES][pcw]
            int pcw = CNT_WORDS;
            if (wide) {
                // every wave has finished reading the exchange buffer (previous pass's key exchange).  The FIRST pass of a two-pass row
                // needs no barrier of its own: the row's qualification (image path: the maximum search; window path: two reductions) put
                // one between the previous row's write-out -- the last reader of the buffer -- and this point.
                if (!(RAWKEYS && two && p == 0)) wg_barrier();
                pcnt = reinterpret_cast<uint32_t *>(xbuf);
                pcw = RR_WIDE_WORDS;
            }
            uint32_t *mycnt = pcnt + wave * pcw;
            const uint32_t cb = lds_off(mycnt);                             // this wave's digit counters, byte address
#pragma unroll
            for (int j = 0; j < CNT_WORDS / WAVE; j++) mycnt[j * WAVE + lane] = 0;
            if (wide) {
#pragma unroll
                for (int j = CNT_WORDS / WAVE; j < RR_WIDE_WORDS / WAVE; j++) mycnt[j * WAVE + lane] = 0;
            }
            // HWORD digit split: low wlo bits = counter word, the rest (0 or 1 bit) = half
            [[maybe_unused]] const uint32_t wlo = wide ? 11u : 10u, whi = (uint32_t)(end - shift) - wlo, hshift = (uint32_t)(shift + (int)wlo) & 31u;
            if constexpr (HWORD) {
                if (wave_live) {
                    uint32_t hr[RR_GH], hs[RR_GH], hg[RR_GH];
                    RRRankHW<ITEMS, PEEL>::run(ir, key, hr, hs, hg, (uint32_t)shift, hshift, wlo, whi, cb, lane, end >= 32 && !two);
                }
            }
            else RRRank<ITEMS>::run(ir, key, shift, lane, cb);
            lds_wait();
            RR_T(1)
            wg_barrier();
            if (SE_RR_PF == 3) { RR_PREFETCH_NEXT_ROW() }
            // ---- S: counters -> first destination of every (wave, digit) ----
            if constexpr (!HWORD) {
            uint32_t ex = 0;
            if (tid < RK_NB) {
                uint32_t run = 0;
#pragma unroll
                for (int w = 0; w < RR_WAVES; w++) run += wcnt[w * RK_NB + tid];
                uint32_t wtot;
                ex = wave_excl_scan(run, wtot);
                if (lane == 63) wave_tot[wave] = wtot;
            }
            wg_barrier();
            if (tid < RK_NB) {   // (the per-wave counts are re-read rather than kept in 8 registers across the barrier)
                for (int w = 0; w < wave; w++) ex += wave_tot[w];
#pragma unroll
                for (int w = 0; w < RR_WAVES; w++) {
                    const uint32_t c = wcnt[w * RK_NB + tid];
                    wcnt[w * RK_NB + tid] = ex;
                    ex += c;
                }
            }
            } else {
                // 8 waves x pcw words of 16-bit counts, two per word (low half: digit w, high half: digit pcw + w).  Thread t owns the
                // words 2t and 2t+1 of every wave (one 8-byte LDS access each; the 12-bit pass: also 1024 + 2t and 1025 + 2t) and all
                // arithmetic stays PACKED: a half never exceeds the 53,248 keys of a row, so the low halves cannot carry into the high ones.
                const bool scanner = tid < RR_SCAN_THREADS;   // (with 768 threads the last 4 waves only take part in the barriers)
                const int st = scanner ? tid : 0;
                uint32_t T0 = 0, T1 = 0, U0 = 0, U1 = 0;   // per-word totals over the waves (U: second word group of the wide pass)
#pragma unroll
                for (int w = 0; w < RR_WAVES; w++) {
                    const uint2 v = *reinterpret_cast<const uint2 *>(pcnt + w * pcw + 2 * st);
                    T0 += v.x; T1 += v.y;
                }
                if (wide) {
                    __builtin_amdgcn_sched_barrier(0);   // one word group at a time: 2 x ITEMS registers are live across the scan
#pragma unroll
                    for (int w = 0; w < RR_WAVES; w++) {
                        const uint2 v = *reinterpret_cast<const uint2 *>(pcnt + w * pcw + CNT_WORDS + 2 * st);
                        U0 += v.x; U1 += v.y;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                uint32_t wtot, wtot2 = 0, ex2 = 0;
                uint32_t ex = wave_excl_scan(T0 + T1, wtot);   // both halves scanned at once
                if (wide) ex2 = wave_excl_scan(U0 + U1, wtot2);
                if (scanner && lane == 63) { wave_tot[wave] = wtot; wave_tot[RR_SCAN_WAVES + wave] = wtot2; }
                wg_barrier();
                uint32_t all = 0, all2 = 0;
#pragma unroll
                for (int w = 0; w < RR_SCAN_WAVES; w++) {
                    const uint32_t wt = wave_tot[w], wt2 = wave_tot[RR_SCAN_WAVES + w];
                    all += wt; all2 += wt2;
                    ex += (w < wave) ? wt : 0u;
                    ex2 += (w < wave) ? wt2 : 0u;
                }
                ex2 += all;                        // the second word group follows the whole first group
                const uint32_t hi = (all + all2) << 16;   // the high-half digits follow ALL low-half digits
                ex += hi; ex2 += hi;
                // digit-major / wave-minor: counts -> first destination of (wave, digit), written back in place
                if (scanner) {
                uint32_t s0 = ex, s1 = ex + T0;
#pragma unroll
                for (int w = 0; w < RR_WAVES; w++) {
                    uint2 *wp = reinterpret_cast<uint2 *>(pcnt + w * pcw + 2 * tid);
                    const uint2 v = *wp;
                    *wp = make_uint2(s0, s1);
                    s0 += v.x; s1 += v.y;
                }
                if (wide) {
                    __builtin_amdgcn_sched_barrier(0);
                    s0 = ex2; s1 = ex2 + U0;
#pragma unroll
                    for (int w = 0; w < RR_WAVES; w++) {
                        uint2 *wp = reinterpret_cast<uint2 *>(pcnt + w * pcw + CNT_WORDS + 2 * tid);
                        const uint2 v = *wp;
                        *wp = make_uint2(s0, s1);
                        s0 += v.x; s1 += v.y;
                    }
                }
                }
            }
            wg_barrier();
            RR_T(2)
            // L2 prefetch of this workgroup's NEXT row (one workgroup per CU: nothing else hides its 200 KB of HBM latency): one dword
            // per 128-byte line, all into one sink register that stays reserved until the loads after the pass loop have been waited for.
            // SE_RR_PF (build-time tuning aid): 0 = no prefetch, 1 = before the destination phase of the last pass (default), 2 = after it, 3 = before its scan
            // (image path: the row's scan + repair follow the last pass -- its prefetch is issued in front of the scan instead: lines brought
            // in this early were evicted again before the loads, 19 % of the row bytes fetched twice)
            if (SE_RR_PF == 1 && !(IMG && two)) { RR_PREFETCH_NEXT_ROW() }
            // ---- X: destinations, then the 2-byte exchanges ----
            if (wave_live) {
#pragma unroll
            for (int s0 = 0; s0 < ITEMS; s0 += 8) {
                uint32_t first[8];
#pragma unroll
                for (int g = 0; g < 8; g++)
                    if (s0 + g < ITEMS) {
                        if constexpr (HWORD) first[g] = reinterpret_cast<const uint16_t *>(mycnt)[(__builtin_amdgcn_ubfe(key[s0 + g], (uint32_t)shift, wlo) << 1) | __builtin_amdgcn_ubfe(key[s0 + g], hshift, whi)];
                        else first[g] = mycnt[(key[s0 + g] >> shift) & 0xFFu];
                    }
#pragma unroll
                for (int g = 0; g < 8; g++)
                    if (s0 + g < ITEMS) {
                        ir[s0 + g] += first[g];   // low half: rank -> destination (< 65536)
                        opaque(ir[s0 + g]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            if (SE_RR_PF == 2) { RR_PREFETCH_NEXT_ROW() }
            if (wide) wg_barrier();   // the scatter below overwrites the (aliased) counters other waves may still be looking up
            RR_T(3)
            if (wave_live) {
#pragma unroll
                for (int s = 0; s < ITEMS; s++) { opaque(ir[s]); lds_st16_hi(RR_DST(ir[s]), ir[s]); }       // index
            }
            lds_wait();
            wg_barrier();
            RR_T(4)
            const bool last = end >= 32;
            if (last) {
                if constexpr (!SEG) break;   // last pass: the index buffer is the ranking
                else {                       // run planes: the indices leave now, the two key halves follow through the same buffer
                    RR_STREAM_PLANE(0, row)
                    wg_barrier();
                }
            } else {
                if (wave_live) RRRead<ITEMS, true>::run(ir, ring, rb);
                lds_wait();
                wg_barrier();
            }
            RR_T(5)
            if (wave_live) {
#pragma unroll
                for (int s = 0; s < ITEMS; s++) { opaque(ir[s]); lds_st16_hi(RR_DST(ir[s]), key[s]); }      // key bits 16-31  (opaque: no cached addresses)
            }
            lds_wait();
            wg_barrier();
            if (SEG && last) {
                RR_STREAM_PLANE(1, row)
            } else {
                if (wave_live) RRRead<ITEMS, true>::run(key, ring, rb);
                lds_wait();
            }
            if (end < 16 || SEG) {                                                       // key bits 0-15: still needed by a later pass (SEG: by the run, so they travel through every pass)
                wg_barrier();
                if (wave_live) {
#pragma unroll
                    for (int s = 0; s < ITEMS; s++) { opaque(ir[s]); lds_st16_lo(RR_DST(ir[s]), key[s]); }  // (low half is still the old key's)
                }
                lds_wait();
                wg_barrier();
                if (SEG && last) break;      // the key registers are free: the next row is loaded before the last plane is streamed out
                if (wave_live) RRRead<ITEMS, false>::run(key, ring, rb);
                lds_wait();
            }
            RR_T(6)
            // (the next pass's barriers order these reads before its first exchange write)
        }
#undef RR_DST
        if constexpr (WIDE && VAR == 2) {
            if (two && n_out > 1) {   // the keys below the window hold the first n_out places in index order: order them by (key, index)
                const uint2 *outl = reinterpret_cast<const uint2 *>(wcnt);
                if (tid < n_out) {
                    const uint2 me = outl[tid];
                    int r = 0;
                    for (int j = 0; j < n_out; j++) {
                        const uint2 o = outl[j];
                        r += (o.x < me.x || (o.x == me.x && o.y < me.y)) ? 1 : 0;
                    }
                    xbuf[r] = (uint16_t)me.y;
                }
                wg_barrier();
            }
        }
        // ---- image path: the exchange buffer is sorted by (image, index); put the runs of equal tags into (key, index) order ----
        [[maybe_unused]] bool img_fail = false;
        if constexpr (IMG && SE_RR_EARLY) {
            // The NEXT row's loads are issued here, straight from HBM (no L2 prefetch): the key registers are dead behind the last pass, and
            // the scan + repair (~20k cycles) cover the latency.  They are waited for BEFORE this row's rank stores are issued, so that
            // nothing has to wait for the stores to drain: the next row's maximum / image phase (VALU only) runs under them.
            const float *drow = row_ptr(more ? row + gridDim.x : row);
            int wpos = wpos0;
            opaque(wpos);
#pragma unroll
            for (int s = 0; s < ITEMS; s++) RR_LOAD_ONE(drow, wpos, s, n_next)
        }
        if constexpr (IMG) {
            if (two) {
                // (per-row opaque copies: otherwise hipcc hoists every row-invariant mask / offset of this block out of the row loop and
                // keeps ~40 of them in scratch -- the reloads cost more than the whole scan)
                int n_row = row_len(row);
                int tsc = wave_s * WAVE + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // (tid itself lives in scratch by now)
                opaque(n_row);
                opaque(tsc);
                const float *drow_cur = row_ptr(row);
                // (1) scan: the tag of final position i is tagb[xbuf[i]] (the tags were written in column order).  Groups of 8 positions, group
                // G = tid + RR_THREADS g: one conflict-free 16-byte read of the indices, 9 random tag reads (the 9th: the next group's first
                // position), E = "tag(i) == tag(i + 1)" by byte-parallel arithmetic on the packed tags.  Every pair with equal tags goes to
                // the worklist (bit 8 b + 4 h of a group's word: position 4 h + b); ONE returning add per thread reserves its slots and
                // only the (rare) set bits are walked -- nothing in this phase waits for LDS inside a divergent loop.  Pairs that reach
                // into the padding become empty entries.
                if (SE_RR_PF == 1 && !SE_RR_EARLY) { RR_PREFETCH_NEXT_ROW_IF(more) }
                constexpr int NS = RR_THREADS * ITEMS;
                constexpr int NG = (NS / 8 + RR_THREADS - 1) / RR_THREADS;   // groups per thread
                uint32_t S[NG];
                uint32_t nstart = 0;
                // software pipeline: indices of all groups first (NG 16-byte reads in flight), then the tag reads of group g + 1 are issued
                // before the arithmetic of group g
                uint4 xg[NG];
                uint32_t nxg[NG];
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    const int b0 = (tsc + RR_THREADS * g) * 8;
                    const int bs = b0 < n_row - 1 ? b0 : 0;
                    xg[g] = *reinterpret_cast<const uint4 *>(xbuf + bs);
                    nxg[g] = xbuf[bs + 8 < NS ? bs + 8 : bs];
                }
                constexpr int TD = SE_RR_SCAN_DEPTH;   // groups whose tag reads are in flight ahead of the arithmetic
                uint32_t t[TD + 1][9];
                auto tag_reads = [&](int g, uint32_t (&tt)[9]) {
                    const uint32_t xi[4] = {xg[g].x, xg[g].y, xg[g].z, xg[g].w};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        tt[2 * j] = tagb[xi[j] & 0xFFFFu];
                        tt[2 * j + 1] = tagb[xi[j] >> 16];
                    }
                    tt[8] = tagb[nxg[g]];
                };
#pragma unroll
                for (int g = 0; g < TD && g < NG; g++) tag_reads(g, t[g % (TD + 1)]);
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    if (g + TD < NG) tag_reads(g + TD, t[(g + TD) % (TD + 1)]);
                    const uint32_t (&tc)[9] = t[g % (TD + 1)];
                    const bool live = (tsc + RR_THREADS * g) * 8 < n_row - 1;
                    const uint32_t A = tc[0] | (tc[1] << 8) | (tc[2] << 16) | (tc[3] << 24), B = tc[4] | (tc[5] << 8) | (tc[6] << 16) | (tc[7] << 24);
                    const uint32_t zA = A ^ __builtin_amdgcn_alignbyte(B, A, 1), zB = B ^ __builtin_amdgcn_alignbyte(tc[8], B, 1);   // byte i: tag(i) ^ tag(i + 1)
                    const uint32_t uA = (zA & 0x7F7F7F7Fu) + 0x7F7F7F7Fu, uB = (zB & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;               // bit 7 of byte i: the low 7 bits of the byte are not all zero
                    const uint32_t eA = __builtin_amdgcn_bitop3_b32(uA, zA, 0x80808080u, 0x02), eB = __builtin_amdgcn_bitop3_b32(uB, zB, 0x80808080u, 0x02);   // ~u & ~z & mask
                    S[g] = live ? ((eA >> 7) | (eB >> 3)) : 0u;
                    nstart += (uint32_t)__popc(S[g]);
                }
                uint32_t slot = 0;
                {
                    // (plain returning add per lane: hipcc's atomic optimizer would turn atomicAdd into a 64-step scalar loop per wave)
                    const uint32_t ca = lds_off(&ictl[0]);
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(slot) : "v"(ca), "v"(nstart) : "memory");
                }
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    uint32_t st = S[g];
                    const int b0 = (tsc + RR_THREADS * g) * 8;
                    while (st) {
                        const int bit = __builtin_ctz(st);
                        st &= st - 1u;
                        const int pos = b0 + (bit & 4) + (bit >> 3);
                        if (slot < (uint32_t)RR_IMG_WL) wlist[slot] = pos < n_row - 1 ? (uint32_t)pos : 0xFFFFFFFFu;
                        slot++;
                    }
                }
                wg_barrier();
                const uint32_t nwork = ictl[0];
                img_fail = nwork > (uint32_t)RR_IMG_WL;
                RR_T(9)
                if (!img_fail) {
                    // (2) repair: worklist entries dealt round-robin over the threads, GB per thread in flight.  An entry is a pair of equal
                    // tags; the thread whose pair has no such pair in front of it owns the run: it finds the length from the tags behind the
                    // pair (2, 3 or "4 and more"), gathers the true keys from the row (L2 / Infinity Cache; all loads of all its runs in
                    // flight together) and puts runs of 2 and 3 -- nearly all -- into (key, index) order in registers; longer runs take a
                    // loop (up to RR_IMG_RUN entries; beyond that the row is given up).  Equal keys keep their places: the sort was stable,
                    // so equal images are already in index order.
                    constexpr int GB = 2;
#pragma unroll 1
                    for (uint32_t e0 = 0; e0 < nwork; e0 += RR_THREADS * GB) {
                        uint32_t pp[GB], im[GB], ia[GB], ib[GB], ic[GB], id[GB], tm[GB], t0[GB], t2[GB], t3[GB], ka[GB], kb[GB], kc[GB];
#pragma unroll
                        for (int k = 0; k < GB; k++) {
                            const uint32_t e = e0 + (uint32_t)(k * RR_THREADS + tsc);
                            pp[k] = e < nwork ? wlist[e] : 0xFFFFFFFFu;
                        }
#pragma unroll
                        for (int k = 0; k < GB; k++) {
                            const uint32_t p = pp[k] != 0xFFFFFFFFu ? pp[k] : 0u;
                            im[k] = xbuf[p ? p - 1u : 0u]; ia[k] = xbuf[p]; ib[k] = xbuf[p + 1];
                            ic[k] = xbuf[p + 2 < (uint32_t)NS ? p + 2 : p]; id[k] = xbuf[p + 3 < (uint32_t)NS ? p + 3 : p];
                        }
#pragma unroll
                        for (int k = 0; k < GB; k++) { tm[k] = tagb[im[k]]; t0[k] = tagb[ia[k]]; t2[k] = tagb[ic[k]]; t3[k] = tagb[id[k]]; }
                        // the gathers (64-byte fabric requests for 4 bytes each: only the ones the run needs -- nothing for a pair whose run
                        // another thread owns, the third key only for a run of three)
#pragma unroll
                        for (int k = 0; k < GB; k++) {
                            const bool mine = pp[k] != 0xFFFFFFFFu && !(pp[k] != 0u && tm[k] == t0[k]);
                            const bool three = mine && (int)pp[k] + 2 < n_row && t2[k] == t0[k];
                            ka[k] = kb[k] = kc[k] = 0u;
                            if (mine) {
                                ka[k] = __float_as_uint(drow_cur[ia[k]]);
                                kb[k] = __float_as_uint(drow_cur[ib[k]]);
                            }
                            if (three) kc[k] = __float_as_uint(drow_cur[ic[k]]);
                        }
#pragma unroll
                        for (int k = 0; k < GB; k++) {
                            if (pp[k] == 0xFFFFFFFFu) continue;
                            const uint32_t p = pp[k];
                            if (p != 0u && tm[k] == t0[k]) continue;                  // the pair in front has equal tags too: its owner takes this run
                            const bool three = (int)p + 2 < n_row && t2[k] == t0[k];
                            const bool more4 = three && (int)p + 3 < n_row && t3[k] == t0[k];
                            const uint32_t ca = rr_key(ka[k], false, 0u), cb = rr_key(kb[k], false, 0u), cc = rr_key(kc[k], false, 0u);
                            if (!three) {
                                if (ca > cb) { xbuf[p] = (uint16_t)ib[k]; xbuf[p + 1] = (uint16_t)ia[k]; }
                            } else if (!more4) {
                                // ranks by counting; ties keep the input order (a before b before c)
                                const uint32_t ra = (cb < ca) + (cc < ca), rb = (ca <= cb) + (cc < cb), rc = (ca <= cc) + (cb <= cc);
                                if (ra != 0u || rb != 1u) {
                                    xbuf[p + ra] = (uint16_t)ia[k]; xbuf[p + rb] = (uint16_t)ib[k]; xbuf[p + rc] = (uint16_t)ic[k];
                                }
                            } else {
                                // rare: 4 .. RR_IMG_RUN entries.  Everything in registers (unrolled, predicated); ranks by counting.
                                int len = 4;
                                while (len <= RR_IMG_RUN && (int)p + len < n_row && tagb[xbuf[p + len]] == t0[k]) len++;
                                if (len > RR_IMG_RUN) { ictl[1] = 1u; continue; }
                                uint32_t idx[RR_IMG_RUN], kk[RR_IMG_RUN];
#pragma unroll
                                for (int j = 0; j < RR_IMG_RUN; j++) idx[j] = xbuf[p + (uint32_t)(j < len ? j : 0)];
#pragma unroll
                                for (int j = 0; j < RR_IMG_RUN; j++) kk[j] = rr_key(__float_as_uint(drow_cur[idx[j]]), false, 0u);
#pragma unroll
                                for (int j = 0; j < RR_IMG_RUN; j++) {
                                    uint32_t r = 0;
#pragma unroll
                                    for (int i = 0; i < RR_IMG_RUN; i++)
                                        if (i != j) r += (i < len && (kk[i] < kk[j] || (kk[i] == kk[j] && i < j))) ? 1u : 0u;
                                    if (j < len) xbuf[p + r] = (uint16_t)idx[j];
                                }
                            }
                        }
                    }
                    wg_barrier();
                    img_fail = ictl[1] != 0u;
                }
                RR_T(10)
            }
        }
        // ---- the exchange buffer now holds the ranking: canonicalise the next row's keys (waits for its loads), then stream the ranks out ----
        // (the loads sit after the pass loop, not inside its last iteration: a re-definition of the key registers on the `break` path makes
        // hipcc copy all ITEMS index registers there, and a separate straight-line instance of the last pass -- measured, DESIGN.md 5.2 --
        // pushes the 98-key build into scratch; the row was prefetched into L2 during the last pass.  They are issued BEFORE the rank
        // stores and waited for after them: the memory pipeline serves them first and the write-out covers most of their latency.)
        if (!(IMG && SE_RR_EARLY) || img_fail) {
            const float *drow = row_ptr((IMG && img_fail) ? row : (more ? row + gridDim.x : row));   // (image path given up: the same row again)
            int wpos = wpos0;
            opaque(wpos);   // per-row opaque: otherwise hipcc hoists ITEMS row-invariant clamps out of the row loop and keeps them live
#pragma unroll
            for (int s = 0; s < ITEMS; s++) RR_LOAD_ONE(drow, wpos, s, n_next)
        }
        if constexpr (IMG && SE_RR_EARLY) {
            // the next row's keys have landed (and the prefetch dwords of a row that took the three passes): nothing is outstanding when the stores start
#pragma unroll
            for (int s = 0; s < ITEMS; s++) opaque(key[s]);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink) : : "memory");
        }
        int wt = tid;
        opaque(wt);   // per-row opaque: the write-out offsets are recomputed here instead of living (spilled) across the whole row loop
        const bool write_out = !IMG || !img_fail;
        if constexpr (SEG) {
            RR_STREAM_PLANE(2, row)
        } else if (!write_out) {
            // image path given up: nothing to write, the row is sorted again
        } else if (idx64 == 2) {
            // uint16 ranks (rows of at most 65,536 columns: what the exchange buffer holds already): 16 bytes = 8 ranks per lane and step,
            // straight from LDS -- half the bytes of the int32 form for the kernel that reads them (se_hierarchical_precision_r16)
            uint16_t *o = (uint16_t *)rank + row * ldr;
            _Pragma("unroll 2") for (int j = wt * 8; j < N; j += RR_THREADS * 8) {
                if (vec_ok && j + 7 < N) {
                    const rr_i32x4 v = *reinterpret_cast<const rr_i32x4 *>(xbuf + j);
                    if (SE_RR_NT) __builtin_nontemporal_store(v, reinterpret_cast<rr_i32x4 *>(o + j));
                    else *reinterpret_cast<rr_i32x4 *>(o + j) = v;
                } else {
                    for (int e = 0; e < 8 && j + e < N; e++) o[j + e] = xbuf[j + e];
                }
            }
        } else if (idx64) {
            int64_t *o = (int64_t *)rank + row * ldr;
            // (the next row's keys are live: no unrolling, the LDS read of the next step is issued before this step's stores instead)
            uint2 nv = *reinterpret_cast<const uint2 *>(xbuf + (wt * 4 < N ? wt * 4 : 0));
            _Pragma("unroll 1") for (int j = wt * 4; j < N; j += RR_THREADS * 4) {
                const uint2 v = nv;
                const int jn = j + RR_THREADS * 4;
                nv = *reinterpret_cast<const uint2 *>(xbuf + (jn < N ? jn : 0));
                const int64_t e0 = v.x & 0xFFFFu, e1 = v.x >> 16, e2 = v.y & 0xFFFFu, e3 = v.y >> 16;
                if (vec_ok && j + 3 < N) {
                    if (SE_RR_NT) {
                        __builtin_nontemporal_store((rr_i64x2){e0, e1}, reinterpret_cast<rr_i64x2 *>(o + j));
                        __builtin_nontemporal_store((rr_i64x2){e2, e3}, reinterpret_cast<rr_i64x2 *>(o + j + 2));
                    } else {
                        *reinterpret_cast<longlong2 *>(o + j) = make_longlong2(e0, e1);
                        *reinterpret_cast<longlong2 *>(o + j + 2) = make_longlong2(e2, e3);
                    }
                } else {
                    o[j] = e0;
                    if (j + 1 < N) o[j + 1] = e1;
                    if (j + 2 < N) o[j + 2] = e2;
                    if (j + 3 < N) o[j + 3] = e3;
                }
            }
        } else {
            int32_t *o = (int32_t *)rank + row * ldr;
            // RR_WO steps per trip: their LDS reads are in flight together, then the stores (with one step per trip and one read ahead, as the
            // int64 path does, the loop paid an LDS round trip per 16-byte store: ~12k of a row's ~100k cycles)
            constexpr int RR_WO = SE_RR_WO;
            _Pragma("unroll 1") for (int j0 = wt * 4; j0 < N; j0 += RR_THREADS * 4 * RR_WO) {
                uint2 v[RR_WO];
#pragma unroll
                for (int u = 0; u < RR_WO; u++) {
                    const int j = j0 + u * RR_THREADS * 4;
                    v[u] = *reinterpret_cast<const uint2 *>(xbuf + (j < N ? j : 0));
                }
#pragma unroll
                for (int u = 0; u < RR_WO; u++) {
                    const int j = j0 + u * RR_THREADS * 4;
                    if (j >= N) break;
                    const int e0 = v[u].x & 0xFFFFu, e1 = v[u].x >> 16, e2 = v[u].y & 0xFFFFu, e3 = v[u].y >> 16;
                    if (vec_ok && j + 3 < N) {
                        if (SE_RR_NT) __builtin_nontemporal_store((rr_i32x4){e0, e1, e2, e3}, reinterpret_cast<rr_i32x4 *>(o + j));
                        else *reinterpret_cast<int4 *>(o + j) = make_int4(e0, e1, e2, e3);
                    } else {
                        o[j] = e0;
                        if (j + 1 < N) o[j + 1] = e1;
                        if (j + 2 < N) o[j + 2] = e2;
                        if (j + 3 < N) o[j + 3] = e3;
                    }
                }
            }
        }
        RR_T(7)
        if constexpr (!RAWKEYS) RR_CANON(n_next)
        if constexpr (PROF) {   // keep the canonicalisation (and with it the wait for the loads) inside the 'load' interval of the phase profile
            _Pragma("unroll") for (int s = 0; s < ITEMS; s++) opaque(key[s]);
        }
        if constexpr (!(IMG && SE_RR_EARLY))
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink) : : "memory");   // prefetch dwords landed too: the sink register is free again, nothing is outstanding
        RR_T(0)
        // (the next row's pass-0 barriers order these reads before its first exchange write)
        if constexpr (IMG) {
            if (img_fail) {
                // (img_fail implies an image attempt: level 0 or 1)
                if (img_level == 0) { img_pen0 = img_pen0 ? (img_pen0 < 16 ? 2 * img_pen0 : 32) : 1; img_skip0 = img_pen0; }
                else { img_pen1 = img_pen1 ? (img_pen1 < 16 ? 2 * img_pen1 : 32) : 1; img_skip1 = img_pen1; }
                img_level++;
                img_again = true;
            } else {
                if (two) { if (img_level == 0) img_pen0 = 0; else img_pen1 = 0; }
                img_again = false;
                row += gridDim.x;
            }
        } else row += gridDim.x;
    }
    if (PROF && tid == 0)
    {
        for (int i = 0; i < 12; i++) atomicAdd(&prof[i], (unsigned long long)t_acc[i]);
    }
#undef RR_T
}

}  // namespace se

using namespace se;

// index width code of se_rank_rows / se_rank_rows_check (`idx64`): 0 = int32, 1 = int64, 2 = uint16 (register-resident rows only)
static size_t rank_idx_bytes(int idx) { return idx == 2 ? 2 : (idx ? 8 : 4); }

static int rank_grid(int64_t q)
{
    // resident workgroups of the tiled kernel: 256 CUs x 2 (512 threads, ~78 KB LDS each)
    const int64_t g = 512;
    return (int)(q < g ? q : g);
}

static int64_t rank_npad(int64_t n) { return (n + 63) / 64 * 64; }

static bool rank_use_tiled(int64_t n)
{
    static const bool force = tuning_env("SE_RANK_TILED") != nullptr;   // -DSE_TUNING build only: always take the general kernel
    return force || n > RR_MAX_N;
}

// Skew detector for the hardware-ordered variant: do the keys share their most significant 10-bit digit (e.g.
// all-positive Euclidean distances: every lane of a wave step would hit ONE or two counters)?  Histogram of that
// digit over 3 rows x 1024 evenly spaced columns; flag = 1 when one value holds >= 30 % of them.  The two kernel variants launched
// behind it read the flag and the one it does not select returns immediately -- no host round trip.
__global__ __launch_bounds__(256) void rank_skew_detect_kernel(const float *__restrict__ pdist, int64_t ldp, int64_t Q, int N,
                                                               int shift, int two_ok, uint32_t *__restrict__ flag)
{
    constexpr int NBIN = 4096;   // values of the most significant digit: 1024 (10 bits, shift 22) or 4096 (12 bits, shift 20)
    __shared__ uint32_t hist[NBIN], hash_hist[NBIN];
    __shared__ uint32_t best, row_max[3], row_below[3], repeats;
    for (int i = threadIdx.x; i < NBIN; i += 256) { hist[i] = 0; hash_hist[i] = 0; }
    if (threadIdx.x == 0) { best = 0; repeats = 0; }
    if (threadIdx.x < 3) { row_max[threadIdx.x] = 0; row_below[threadIdx.x] = 0; }
    wg_barrier();
    const int cols = N < 1024 ? N : 1024;
    // (round 6: the 12 samples of a thread -- 3 rows x up to 4 columns -- are requested together and kept in registers for both steps:
    // the kernel used to walk them twice, one global-load round trip after the other: 42 us in front of every ranking call)
    uint32_t kk[3][4];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int64_t row = (r == 0) ? 0 : (r == 1 ? Q / 2 : Q - 1);
        const float *drow = pdist + row * ldp;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = threadIdx.x + 256 * u;
            kk[r][u] = canon_key(drow[(int64_t)(i < cols ? i : 0) * N / cols]);
        }
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = threadIdx.x + 256 * u;
            if (i >= cols) continue;
            const uint32_t k = kk[r][u];
            atomicAdd(&hist[k >> shift], 1u);
            if (k != 0xFFFFFFFFu) atomicMax(&row_max[r], k);
            // repeated keys among the first 256 sampled columns of the row (hash buckets: 4096 for 768 keys -- ~70 chance hits): rows made of
            // a few distinct values would give up the image path row by row
            if (i < 256 && atomicAdd(&hash_hist[((k ^ ((uint32_t)r * 0x3C6EF372u)) * 2654435761u) >> 20], 1u) != 0u) atomicAdd(&repeats, 1u);   // (r: calls of one row sample it three times)
        }
    }
    wg_barrier();
    // two-pass candidates: (nearly) every sampled key of every sampled row within RR_TWO_SPAN codes of the row's largest one --
    // one sampled key below the window stands for ~N / 1024 in the row; each row checks itself again inside the kernel
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const uint32_t lo = row_max[r] > RR_TWO_SPAN ? row_max[r] - RR_TWO_SPAN : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (threadIdx.x + 256 * u < cols && kk[r][u] < lo) atomicAdd(&row_below[r], 1u);
    }
    uint32_t mine = 0;
    for (int i = threadIdx.x; i < NBIN; i += 256) mine = hist[i] > mine ? hist[i] : mine;
    atomicMax(&best, mine);
    wg_barrier();
    // peeling pays from roughly a 30 % share of one digit (two-valued Euclidean rows: ~50 %; mixed-sign cosine rows: ~10 %)
    if (threadIdx.x == 0) {
        const bool two = (two_ok & 1) && row_below[0] <= 4 && row_below[1] <= 4 && row_below[2] <= 4;   // ~N / 1024 keys of the row per sampled key
        // window fits: two lossless passes.  Otherwise rows without a dominant most significant digit (mixed-sign cosine rows, wide positive
        // rows) take the image path; every row of it checks itself and falls back to the three passes (plain rank phase) when it must.
        const bool skewed = 10u * best >= 3u * 3u * (uint32_t)cols;
        const bool distinct = repeats < 3u * 96u;                       // fewer than ~3 / 8 of the hashed keys met an occupied bucket
        // (rows of a few distinct values: neither two-pass form spreads them over the counters -- every wave step would queue up on two
        // or three of them in BOTH passes; the three passes with the group-peeling last pass, or plain, are the better choice)
        *flag = (two && distinct) ? 2u : (skewed ? 1u : (((two_ok & 2) && distinct) ? 3u : 0u));
    }
}

template <int ITEMS, bool HW, int VAR>
static int launch_rank_reg_variant(const float *pdist, int64_t ldp, int64_t q, int n, void *rank, int idx64, int64_t ldr,
                                   const uint32_t *skew_flag, hipStream_t s)
{
    // dedicated counters (packed 16-bit vs 32-bit; image path: tag plane + worklist) + wave_tot + exchange buffer
    const size_t lds = rr_region0_bytes<ITEMS, HW, VAR>() + 32 * sizeof(uint32_t) + (size_t)RR_THREADS * ITEMS * sizeof(uint16_t);
    static_assert(rr_region0_bytes<ITEMS, HW, VAR>() + 32 * sizeof(uint32_t) + (size_t)RR_THREADS * ITEMS * sizeof(uint16_t) <= 160 * 1024, "LDS of one CU");
    static const bool profile = tuning_env("SE_RR_PROFILE") != nullptr;   // -DSE_TUNING build only: allocates, synchronises, prints
    auto kern = (kTuning && profile) ? rank_rows_reg_kernel<ITEMS, kTuning && (ITEMS == 98), HW, VAR> : rank_rows_reg_kernel<ITEMS, false, HW, VAR>;
    // per instantiation, computed once (thread-safe static initialisation): resident workgroups = CUs x occupancy
    struct Resident { hipError_t err; int64_t grid; };
    static const Resident res = [&]() -> Resident {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0, occ = 0;
        hipDeviceProp_t prop;
        if (e == hipSuccess) e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kern, RR_THREADS, lds);
        if (e != hipSuccess) return {e, 0};
        return {hipSuccess, (int64_t)(prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256) * (occ > 0 ? occ : 1)};
    }();
    if (res.err != hipSuccess) return fail(SE_ERR_HIP, "se_rank_rows: kernel set-up failed: %s", hipGetErrorString(res.err));
    int64_t grid = res.grid;
    if (grid > q) grid = q;
    const size_t esz = rank_idx_bytes(idx64);
    const int vec_ok = ((((uintptr_t)rank) & 15) == 0) && ((ldr * esz) % 16 == 0);
    unsigned long long *prof = nullptr;
    if (profile && ITEMS == 98) {
        SE_HIP_CHECK(hipMalloc((void **)&prof, 36 * sizeof(unsigned long long)));
        SE_HIP_CHECK(hipMemsetAsync(prof, 0, 36 * sizeof(unsigned long long), s));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(RR_THREADS), lds, s, pdist, ldp, q, n, rank, ldr, idx64, vec_ok, prof, skew_flag, RankSeg{0, 0, nullptr, 0});
    SE_LAUNCH_CHECK();
    if (prof) {
        unsigned long long h[36];
        SE_HIP_CHECK(hipStreamSynchronize(s));
        SE_HIP_CHECK(hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost));
        SE_HIP_CHECK(hipFree(prof));
        static const char *names[12] = {"load", "rank", "scan", "dest", "idx-write", "idx-read", "key-exchange", "write-out", "image-map|window-map", "tag-scan|window-max", "repair|window-barrier-1", "window-count"};
        double tot = 0;
        for (int i = 0; i < 12; i++) tot += (double)h[i];
        if (tot > 0) {
            fprintf(stderr, "[se_rank_rows profile] ITEMS=%d hw=%d peel=%d grid=%lld:", ITEMS, (int)HW, VAR, (long long)grid);
            for (int i = 0; i < 12; i++) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * (double)h[i] / tot);
            fprintf(stderr, "  (%.0f cycles per row; wave 0 of every workgroup, timestamps behind the phases' own barriers)", tot / (double)q);
            fprintf(stderr, "\n");
        }
    }
    return SE_OK;
}

// hw: hardware-ordered variant allowed (capability probe passed).  scratch: >= 256 bytes of caller workspace or NULL;
// with scratch the skew detector picks between the plain and the group-peeling hardware-ordered kernels.
template <int ITEMS>
static int launch_rank_reg(const float *pdist, int64_t ldp, int64_t q, int n, void *rank, int idx64, int64_t ldr, bool hw, void *scratch,
                           hipStream_t s)
{
    if (!hw) return launch_rank_reg_variant<ITEMS, false, 0>(pdist, ldp, q, n, rank, idx64, ldr, nullptr, s);
    // long rows (12-bit last digit, counters aliased onto the exchange buffer: WIDE in the kernel) also have the two-pass variant
    constexpr bool wide = (size_t)RR_THREADS * ITEMS * sizeof(uint16_t) >= (size_t)RR_WAVES * RR_WIDE_WORDS * sizeof(uint32_t);
    constexpr bool two_ok = wide && SE_RR_TWO;
    constexpr bool img_ok = wide && SE_RR_IMG && ITEMS <= RR_IMG_MAX_ITEMS && RR_THREADS == 512;
    static const char *force = tuning_env("SE_RANK_PEEL");   // -DSE_TUNING build only: "0" / "1" / "2" / "3" pins the variant
    if (force || !scratch) {
        if (force && force[0] == '1') return launch_rank_reg_variant<ITEMS, true, 1>(pdist, ldp, q, n, rank, idx64, ldr, nullptr, s);
        if (force && force[0] == '2' && two_ok) return launch_rank_reg_variant<ITEMS, true, two_ok ? 2 : 0>(pdist, ldp, q, n, rank, idx64, ldr, nullptr, s);
        if (force && force[0] == '3' && img_ok) return launch_rank_reg_variant<ITEMS, true, img_ok ? 3 : 0>(pdist, ldp, q, n, rank, idx64, ldr, nullptr, s);
        return launch_rank_reg_variant<ITEMS, true, 0>(pdist, ldp, q, n, rank, idx64, ldr, nullptr, s);
    }
    uint32_t *flag = (uint32_t *)scratch + 16;   // (words 0-1 belong to the capability probe)
    // first bit of the most significant digit: 20 for the instantiations whose last pass is 12 bits wide, else 22
    const int top_shift = wide ? 20 : 2 * RR_HW_BITS;
    hipLaunchKernelGGL(rank_skew_detect_kernel, dim3(1), dim3(256), 0, s, pdist, ldp, q, n, top_shift, (two_ok ? 1 : 0) | (img_ok ? 2 : 0), flag);
    SE_LAUNCH_CHECK();
    int rc = launch_rank_reg_variant<ITEMS, true, 0>(pdist, ldp, q, n, rank, idx64, ldr, flag, s);
    if (rc != SE_OK) return rc;
    rc = launch_rank_reg_variant<ITEMS, true, 1>(pdist, ldp, q, n, rank, idx64, ldr, flag, s);
    if (rc != SE_OK) return rc;
    if (two_ok) rc = launch_rank_reg_variant<ITEMS, true, two_ok ? 2 : 0>(pdist, ldp, q, n, rank, idx64, ldr, flag, s);
    if (rc != SE_OK || !img_ok) return rc;
    return launch_rank_reg_variant<ITEMS, true, img_ok ? 3 : 0>(pdist, ldp, q, n, rank, idx64, ldr, flag, s);
}

// ---- rows of more than RR_MAX_N columns: sorted runs + merge tree -----------------------------------------------------------------
// A row's working set in the register-resident kernel is 8 bytes of registers + 2 bytes of LDS per key: ~60k keys per CU.  Longer rows
// are cut by POSITION into S = 2, 4 or 8 segments, each sorted by that kernel (SEG build: output = a run of (key, segment-local
// index) in three 16-bit planes), and the runs are merged pairwise, level by level: ties between two runs go to the first one, whose
// indices are all smaller -- the canonical (key, index) order.  The last level writes the ranks; the levels before it (S > 2) write
// merged runs as (key, index) dword arrays laid out along the row (a run made of segments j .. j' starts at entry j x seg_n).  Per
// chunk of rows and level: merge-path partition (one thread per output tile: where the tile starts in the pair's first run), tile
// merge.  Chunks keep the scratch (6 bytes per key for the planes, 8 per intermediate level buffer) bounded.
constexpr int RC_CAP = 1022;   // listed rows (workspace: 256 B of probe / detector words + 4 KB of list)
constexpr int MG_THREADS = 256;
constexpr int MG_VT = 16;                         // outputs per thread of the merge kernel (tuning build: SE_MG_VT = 8 / 12 / 24)
constexpr int RUNS_MAX_SEG = 8;

struct MergeLevel {
    // input runs of a row: run j covers entries [j * run_n, min(N, (j + 1) * run_n)); pair p merges runs 2p and 2p + 1
    int run_n, N, pairs, tiles, tile;      // tiles per pair (of `tile` outputs each)
    // planes input (first level): run j of chunk row r = virtual row r * (2 * pairs) + j, `cap` entries per plane row
    const uint16_t *planes; int64_t plane_elems; int cap;
    // dword input (later levels): key / index arrays, `ld` entries per row, runs at their row offsets
    const uint32_t *in_key, *in_idx; int64_t ld;
};

template <bool PLANES>
__device__ __forceinline__ uint32_t level_key(const MergeLevel &L, int64_t r, int run, int i)
{
    if constexpr (PLANES) {
        const uint16_t *khi = L.planes + L.plane_elems + (r * (2 * L.pairs) + run) * (int64_t)L.cap, *klo = khi + L.plane_elems;
        return ((uint32_t)khi[i] << 16) | klo[i];
    } else return L.in_key[r * L.ld + (int64_t)run * L.run_n + i];
}
__device__ __forceinline__ int level_run_len(const MergeLevel &L, int run)
{
    const int64_t rest = (int64_t)L.N - (int64_t)run * L.run_n;
    return rest <= 0 ? 0 : (rest < L.run_n ? (int)rest : L.run_n);
}

// splits[(r * pairs + p) * (tiles + 1) + t] = entries of the pair's first run among the first min(t * tile, pair length) merged entries
template <bool PLANES>
__global__ __launch_bounds__(256) void rank_merge_partition_kernel(const MergeLevel L, int64_t rows, int32_t *__restrict__ splits)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= rows * L.pairs * (L.tiles + 1)) return;
    const int64_t rp = g / (L.tiles + 1);
    const int t = (int)(g - rp * (L.tiles + 1));
    const int64_t r = rp / L.pairs;
    const int p = (int)(rp - r * L.pairs);
    const int na = level_run_len(L, 2 * p), nb = level_run_len(L, 2 * p + 1);
    const int64_t d64 = (int64_t)t * L.tile;
    const int d = (int)(d64 < na + nb ? d64 : na + nb);
    int lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        // A[mid] goes before B[d - 1 - mid] when its key is <= (ties: run A first)
        if (level_key<PLANES>(L, r, 2 * p, mid) <= level_key<PLANES>(L, r, 2 * p + 1, d - 1 - mid)) lo = mid + 1;
        else hi = mid;
    }
    splits[g] = lo;
}

// one part of a tile -> LDS: entries [e0, e0 + cnt) of a run to sK / sI [dst, dst + cnt)
template <bool PLANES, typename IT>
__device__ __forceinline__ void merge_stage(const MergeLevel &L, int64_t r, int run, int e0, int cnt, uint32_t *sK, IT *sI, int dst, int tid)
{
    if constexpr (PLANES) {     // 4 entries (8 bytes of each plane) per lane and step
        const uint16_t *ix = L.planes + (r * (2 * L.pairs) + run) * (int64_t)L.cap, *khi = ix + L.plane_elems, *klo = khi + L.plane_elems;
        const int e1 = e0 + cnt;
        for (int j = (e0 & ~3) + tid * 4; j < e1; j += MG_THREADS * 4) {
            const uint2 x = *reinterpret_cast<const uint2 *>(ix + j), h = *reinterpret_cast<const uint2 *>(khi + j), l = *reinterpret_cast<const uint2 *>(klo + j);
            const uint32_t k[4] = {(h.x << 16) | (l.x & 0xFFFFu), (h.x & 0xFFFF0000u) | (l.x >> 16), (h.y << 16) | (l.y & 0xFFFFu), (h.y & 0xFFFF0000u) | (l.y >> 16)};
            const uint32_t id[4] = {x.x & 0xFFFFu, x.x >> 16, x.y & 0xFFFFu, x.y >> 16};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int pos = j + e - e0;
                if (pos >= 0 && pos < cnt) { sK[dst + pos] = k[e]; sI[dst + pos] = (IT)id[e]; }
            }
        }
    } else {
        const uint32_t *k = L.in_key + r * L.ld + (int64_t)run * L.run_n + e0, *ix = L.in_idx + r * L.ld + (int64_t)run * L.run_n + e0;
        for (int j = tid; j < cnt; j += MG_THREADS) { sK[dst + j] = k[j]; sI[dst + j] = (IT)ix[j]; }
    }
}

// OUT_RANKS: the level's single pair is the whole row and its merged order is the ranking; otherwise the merged run goes to
// out_key / out_idx (dword arrays like the later levels' input).
template <bool PLANES, bool OUT_RANKS, bool IDX64, int VT>
__global__ __launch_bounds__(MG_THREADS) void rank_merge_kernel(const MergeLevel L, const int32_t *__restrict__ splits, void *__restrict__ rank, int64_t ldr,
                                                                int vec_ok, uint32_t *__restrict__ out_key, uint32_t *__restrict__ out_idx)
{
    constexpr int TILE = MG_THREADS * VT;
    using IT = typename std::conditional<PLANES, uint16_t, uint32_t>::type;
    __shared__ __attribute__((aligned(16))) uint32_t sK[TILE + 8];
    __shared__ __attribute__((aligned(16))) IT sI[TILE + 8];
    const int tid = threadIdx.x;
    const int p = blockIdx.x / L.tiles, t = blockIdx.x - p * L.tiles;
    const int64_t r = blockIdx.y;
    const int na_run = level_run_len(L, 2 * p), nb_run = level_run_len(L, 2 * p + 1);
    const int64_t o0 = (int64_t)t * TILE;
    if (o0 >= na_run + nb_run) return;               // (a row's last pair may be shorter than the others: whole workgroup)
    const int32_t *sp = splits + (r * L.pairs + p) * (int64_t)(L.tiles + 1) + t;
    const int a0 = sp[0], a1 = sp[1];
    const int tot = (int)((o0 + TILE < na_run + nb_run ? o0 + TILE : na_run + nb_run) - o0);
    const int b0 = (int)o0 - a0;
    const int na = a1 - a0, nb = tot - na;
    // ---- both parts of the tile -> LDS: [0, na) from run A, [na, tot) from run B ----
    merge_stage<PLANES>(L, r, 2 * p, a0, na, sK, sI, 0, tid);
    merge_stage<PLANES>(L, r, 2 * p + 1, b0, nb, sK, sI, na, tid);
    wg_barrier();
    // ---- this thread's VT outputs start at diagonal d of the tile: merge path through LDS ----
    const int d = tid * VT < tot ? tid * VT : tot;
    int lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sK[mid] <= sK[na + d - 1 - mid]) lo = mid + 1;
        else hi = mid;
    }
    int a = lo, b = d - lo;
    uint32_t ka = a < na ? sK[a] : 0u, kb = b < nb ? sK[na + b] : 0u;
    // planes carry segment-local indices: the segment's first column is added here (run A = segment 2p, run B = segment 2p + 1)
    const uint32_t base_a = PLANES ? (uint32_t)(2 * p) * (uint32_t)L.run_n : 0u, base_b = PLANES ? base_a + (uint32_t)L.run_n : 0u;
    uint32_t out[VT];
    [[maybe_unused]] uint32_t outk[OUT_RANKS ? 1 : VT];
#pragma unroll
    for (int i = 0; i < VT; i++) {
        const bool take_a = (b >= nb) || (a < na && ka <= kb);
        const int pp = take_a ? a : na + b;
        out[i] = (uint32_t)sI[pp < tot ? pp : 0] + (take_a ? base_a : base_b);     // (beyond the tile's end the value is not stored)
        if constexpr (!OUT_RANKS) outk[i] = take_a ? ka : kb;
        if (take_a) { a++; ka = sK[a < na ? a : 0]; }
        else { b++; kb = sK[na + (b < nb ? b : 0)]; }
    }
    wg_barrier();
    if constexpr (OUT_RANKS) {
        // ---- transpose through LDS: 16-byte stores of consecutive ranks ----
#pragma unroll
        for (int i = 0; i < VT; i++) sK[tid * VT + i] = out[i];
        wg_barrier();
        if constexpr (IDX64) {
            int64_t *o = (int64_t *)rank + r * ldr + o0;
            for (int j = tid * 2; j < tot; j += MG_THREADS * 2) {
                if (vec_ok && j + 1 < tot) *reinterpret_cast<longlong2 *>(o + j) = make_longlong2((int64_t)sK[j], (int64_t)sK[j + 1]);
                else { o[j] = sK[j]; if (j + 1 < tot) o[j + 1] = sK[j + 1]; }
            }
        } else {
            int32_t *o = (int32_t *)rank + r * ldr + o0;
            for (int j = tid * 4; j < tot; j += MG_THREADS * 4) {
                if (vec_ok && j + 3 < tot) *reinterpret_cast<uint4 *>(o + j) = *reinterpret_cast<const uint4 *>(sK + j);
                else
                    for (int e = 0; e < 4 && j + e < tot; e++) o[j + e] = (int32_t)sK[j + e];
            }
        }
    } else {
        // ---- merged run: keys and indices to the level buffers, coalesced through LDS (sK: keys; the indices follow through it too) ----
        const int64_t ob = r * L.ld + (int64_t)(2 * p) * L.run_n + o0;
#pragma unroll
        for (int i = 0; i < VT; i++) sK[tid * VT + i] = outk[i];
        wg_barrier();
        for (int j = tid; j < tot; j += MG_THREADS) out_key[ob + j] = sK[j];
        wg_barrier();
#pragma unroll
        for (int i = 0; i < VT; i++) sK[tid * VT + i] = out[i];
        wg_barrier();
        for (int j = tid; j < tot; j += MG_THREADS) out_idx[ob + j] = sK[j];
    }
}

static int rank_merge_vt()
{
    if (kTuning) { const char *e = tuning_env("SE_MG_VT"); if (e) { const int v = atoi(e); if (v == 8 || v == 12 || v == 24) return v; } }
    return MG_VT;
}

// segments per row: the smallest of 2 / 4 / 8 whose segments fit the register-resident kernel; 0 = none (tiled kernel)
static int rank_runs_segments(int64_t n)
{
    static const bool off = tuning_env("SE_RANK_NORUNS") != nullptr;     // -DSE_TUNING build only: long rows take the tiled kernel
    if (off || n <= RR_MAX_N) return 0;
    for (int sgm = 2; sgm <= RUNS_MAX_SEG; sgm *= 2)
        if (((n + sgm - 1) / sgm + 7) / 8 * 8 <= RR_MAX_N) return sgm;
    return 0;
}
static bool rank_runs_ok(int64_t n) { return rank_runs_segments(n) != 0; }
static int64_t rank_runs_seg_n(int64_t n) { const int sgm = rank_runs_segments(n); return ((n + sgm - 1) / sgm + 7) / 8 * 8; }
static int rank_runs_items(int64_t n)
{
    const int items = (int)((rank_runs_seg_n(n) + RR_THREADS - 1) / RR_THREADS);
    return items <= 64 ? 64 : items <= 72 ? 72 : items <= 80 ? 80 : items <= 88 ? 88 : items <= 98 ? 98 : 104;
}
constexpr int64_t RUNS_HEAD = 256 + 4 * (RC_CAP + 2);
struct RunsLayout { int64_t chunk, cap, tile, split_words, head, split_bytes, plane_bytes, level_bytes, total; int sgm; };
static RunsLayout rank_runs_layout(int64_t q, int64_t n)
{
    RunsLayout y;
    y.sgm = rank_runs_segments(n);
    y.cap = (int64_t)RR_THREADS * rank_runs_items(n);
    y.tile = (int64_t)MG_THREADS * rank_merge_vt();
    // rows per chunk: ~3 GB of planes + level buffers, at most 4,096 rows (tuning build: SE_RANK_CHUNK; chunks of 128 ... 4,096 rows
    // measured 10.3 ... 7.4 ps per key at 100,000 columns: fewer, longer launches of the persistent segment kernel)
    const int64_t per_row = 6 * y.sgm * y.cap + (y.sgm > 2 ? 8 * (n + 8) : 0) + (y.sgm > 4 ? 8 * (n + 8) : 0);
    int64_t c = (int64_t)3072 * 1024 * 1024 / per_row;
    c = c > 4096 ? 4096 : (c < 64 ? 64 : c);
    if (kTuning) { const char *e = tuning_env("SE_RANK_CHUNK"); if (e && atoll(e) > 0) c = atoll(e); }
    y.chunk = q < c ? q : c;
    // split table: the first level has the most entries (rows x pairs x (tiles + 1), tiles per pair of 2 seg_n entries)
    const int64_t seg_n = rank_runs_seg_n(n);
    int64_t most = 0;
    for (int64_t pairs = y.sgm / 2, run = seg_n; pairs >= 1; pairs /= 2, run *= 2) {
        const int64_t words = y.chunk * pairs * ((2 * run + y.tile - 1) / y.tile + 1);
        most = words > most ? words : most;
    }
    y.split_words = most;
    y.head = (RUNS_HEAD + 255) / 256 * 256;
    y.split_bytes = (most * 4 + 255) / 256 * 256;
    y.plane_bytes = (3 * y.sgm * y.chunk * y.cap * 2 + 255) / 256 * 256;
    y.level_bytes = y.sgm > 2 ? (y.chunk * (n + 8) * 4 + 255) / 256 * 256 : 0;      // one dword array of a level buffer
    y.total = y.head + y.split_bytes + y.plane_bytes + (y.sgm > 2 ? 2 : 0) * y.level_bytes + (y.sgm > 4 ? 2 : 0) * y.level_bytes;
    return y;
}
static int64_t rank_runs_bytes(int64_t q, int64_t n) { return rank_runs_layout(q, n).total; }

template <bool PLANES, bool OUT_RANKS>
static int launch_merge_level(const MergeLevel &L, int64_t rows, int32_t *splits, void *rout, int idx64, int64_t ldr, int vec_ok, uint32_t *out_key,
                              uint32_t *out_idx, int vt, hipStream_t s)
{
    const int64_t nsplit = rows * L.pairs * (L.tiles + 1);
    hipLaunchKernelGGL(rank_merge_partition_kernel<PLANES>, dim3((unsigned)((nsplit + 255) / 256)), dim3(256), 0, s, L, rows, splits);
    SE_LAUNCH_CHECK();
    const dim3 mgrid((unsigned)(L.pairs * L.tiles), (unsigned)rows);
#define SE_MG_LAUNCH(I64, V) hipLaunchKernelGGL((rank_merge_kernel<PLANES, OUT_RANKS, I64, V>), mgrid, dim3(MG_THREADS), 0, s, L, splits, rout, ldr, vec_ok, out_key, out_idx)
    if (kTuning && vt != MG_VT) {
        if (vt == 8) { if (OUT_RANKS && idx64) SE_MG_LAUNCH(true, 8); else SE_MG_LAUNCH(false, 8); }
        else if (vt == 12) { if (OUT_RANKS && idx64) SE_MG_LAUNCH(true, 12); else SE_MG_LAUNCH(false, 12); }
        else { if (OUT_RANKS && idx64) SE_MG_LAUNCH(true, 24); else SE_MG_LAUNCH(false, 24); }
    } else if (OUT_RANKS && idx64) SE_MG_LAUNCH(true, MG_VT);
    else SE_MG_LAUNCH(false, MG_VT);
#undef SE_MG_LAUNCH
    SE_LAUNCH_CHECK();
    return SE_OK;
}

template <int ITEMS>
static int launch_rank_runs(const float *pdist, int64_t ldp, int64_t q, int n, void *rank, int idx64, int64_t ldr, void *workspace, hipStream_t s)
{
    constexpr bool wide = (size_t)RR_THREADS * ITEMS * sizeof(uint16_t) >= (size_t)RR_WAVES * RR_WIDE_WORDS * sizeof(uint32_t);
    static_assert(wide, "segment runs use the long-row instantiations");
    const size_t cnt_words = (size_t)(1 << RR_HW_BITS) / 2;
    const size_t lds = (RR_WAVES * cnt_words + 32) * sizeof(uint32_t) + (size_t)RR_THREADS * ITEMS * sizeof(uint16_t);
    auto kern = rank_rows_reg_kernel<ITEMS, false, true, 0, true>;
    struct Resident { hipError_t err; int64_t grid; };
    static const Resident res = [&]() -> Resident {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0, occ = 0;
        hipDeviceProp_t prop;
        if (e == hipSuccess) e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kern, RR_THREADS, lds);
        if (e != hipSuccess) return {e, 0};
        return {hipSuccess, (int64_t)(prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256) * (occ > 0 ? occ : 1)};
    }();
    if (res.err != hipSuccess) return fail(SE_ERR_HIP, "se_rank_rows: kernel set-up failed: %s", hipGetErrorString(res.err));
    const RunsLayout y = rank_runs_layout(q, n);
    if (y.cap != (int64_t)RR_THREADS * ITEMS) return fail(SE_ERR_INVALID, "se_rank_rows: run layout mismatch");
    const int vt = (int)(y.tile / MG_THREADS), sgm = y.sgm;
    const int seg_n = (int)rank_runs_seg_n(n);
    int seg_shift = 0;
    while ((1 << seg_shift) < sgm) seg_shift++;
    char *w = (char *)workspace + y.head;
    int32_t *splits = (int32_t *)w;                         w += y.split_bytes;
    uint16_t *planes = (uint16_t *)w;                       w += y.plane_bytes;
    uint32_t *lvl[4] = {nullptr, nullptr, nullptr, nullptr};     // key / index arrays of the two intermediate level buffers
    for (int i = 0; i < (sgm > 4 ? 4 : (sgm > 2 ? 2 : 0)); i++) { lvl[i] = (uint32_t *)w; w += y.level_bytes; }
    const int64_t ld = (int64_t)n + 8;
    const size_t esz = idx64 ? 8 : 4;
    const int vec_ok = ((((uintptr_t)rank) & 15) == 0) && ((ldr * esz) % 16 == 0);
    for (int64_t r0 = 0; r0 < q; r0 += y.chunk) {
        const int64_t rows = q - r0 < y.chunk ? q - r0 : y.chunk, vrows = sgm * rows;
        const int64_t plane_elems = sgm * rows * y.cap;
        const int64_t grid = res.grid < vrows ? res.grid : vrows;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(RR_THREADS), lds, s, pdist + r0 * ldp, ldp, vrows, n, (void *)nullptr, (int64_t)0, 0, 0,
                           (unsigned long long *)nullptr, (const uint32_t *)nullptr, RankSeg{seg_shift, seg_n, planes, plane_elems});
        SE_LAUNCH_CHECK();
        void *rout = (char *)rank + (size_t)r0 * (size_t)ldr * esz;
        int level = 0;
        for (int pairs = sgm / 2, run = seg_n; pairs >= 1; pairs /= 2, run *= 2, level++) {
            MergeLevel L;
            L.run_n = run; L.N = n; L.pairs = pairs; L.tile = (int)y.tile; L.tiles = (int)((2 * (int64_t)run + y.tile - 1) / y.tile);
            L.planes = planes; L.plane_elems = plane_elems; L.cap = (int)y.cap;
            const int src = (level - 1) & 1, dst = level & 1;                     // level buffers alternate
            L.in_key = level ? lvl[2 * src] : nullptr; L.in_idx = level ? lvl[2 * src + 1] : nullptr; L.ld = ld;
            const bool last = pairs == 1;
            int rc;
            if (level == 0) rc = last ? launch_merge_level<true, true>(L, rows, splits, rout, idx64, ldr, vec_ok, nullptr, nullptr, vt, s)
                                      : launch_merge_level<true, false>(L, rows, splits, nullptr, 0, 0, 0, lvl[2 * dst], lvl[2 * dst + 1], vt, s);
            else rc = last ? launch_merge_level<false, true>(L, rows, splits, rout, idx64, ldr, vec_ok, nullptr, nullptr, vt, s)
                           : launch_merge_level<false, false>(L, rows, splits, nullptr, 0, 0, 0, lvl[2 * dst], lvl[2 * dst + 1], vt, s);
            if (rc != SE_OK) return rc;
        }
    }
    return SE_OK;
}

// ---- capability probe for the hardware-ordered ranking --------------------------------------------------
// Every wave of 64 workgroups issues returning LDS adds under four conflict patterns (one address, 4, 16,
// 256 addresses) while its 7 sibling waves do the same, and compares each returned value with the stable rank
// computed by the ballot multisplit -- first with 32-bit counters and one add at a time, then the way the production kernel
// uses them: packed 16-bit halves of a shared word and eight adds in flight per lane.  res[0] = mismatches, res[1] = waves that reported.
constexpr int RR_PROBE_BLOCKS = 64, RR_PROBE_STEPS = 48;
__global__ __launch_bounds__(RR_THREADS) void rank_order_probe_kernel(uint32_t *res)
{
    __shared__ uint32_t cnt[RR_WAVES][RK_NB], ref[RR_WAVES][RK_NB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < RK_NB; i += WAVE) { cnt[wave][i] = 0; ref[wave][i] = 0; }
    wg_barrier();
    const uint32_t cb = lds_off(&cnt[wave][0]);
    uint32_t bad = 0, seed = (blockIdx.x * RR_THREADS + threadIdx.x) * 2654435761u + 12345u;
    const int mode = blockIdx.x & 3;
    for (int s = 0; s < RR_PROBE_STEPS; s++) {
        seed = seed * 1664525u + 1013904223u;
        const uint32_t rnd = seed >> 24;
        const uint32_t d = mode == 0 ? 7u : mode == 1 ? (rnd & 3u) : mode == 2 ? (rnd & 15u) * 16u : rnd;
        uint32_t got;
        const uint32_t ca = cb + (d << 2), one = 1u;
        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(ca), "v"(one) : "memory");
        uint32_t dlo, dhi;
        differ_mask(d, dlo, dhi);
        const uint32_t rnk = (uint32_t)lane - __builtin_amdgcn_mbcnt_hi(dhi, __builtin_amdgcn_mbcnt_lo(dlo, 0u));
        const uint32_t want = ref[wave][d] + rnk;
        if (rnk == 0) ref[wave][d] += 64u - (uint32_t)(__popc(dlo) + __popc(dhi));   // one lane per digit group
        bad += (got != want);
    }
    // ---- the same property under PRODUCTION conditions: packed 16-bit counter halves (increment 1 or 1 << 16 on the shared word),
    // eight returning adds in flight per lane before the first result is looked at, all eight waves hammering their tables ----
    wg_barrier();
    for (int i = lane; i < RK_NB; i += WAVE) { cnt[wave][i] = 0; ref[wave][i] = 0; }   // ref: low half = expected count of (word, half 0), high half = of half 1
    wg_barrier();
    for (int batch = 0; batch < 6; batch++) {
        uint32_t got[8], dw[8], dh[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            seed = seed * 1664525u + 1013904223u;
            const uint32_t rnd = seed >> 23;
            dw[j] = mode == 0 ? 7u : mode == 1 ? (rnd & 3u) : mode == 2 ? (rnd & 15u) * 16u : (rnd & 255u);
            dh[j] = (rnd >> 8) & 1u;
            const uint32_t ca = cb + (dw[j] << 2), inc = dh[j] ? 0x10000u : 1u;
            asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(got[j]) : "v"(ca), "v"(inc) : "memory");   // no wait: 8 in flight
        }
        lds_wait_le<0>(got[0]);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            opaque(got[j]);
            uint32_t dlo, dhi;
            differ_mask(dw[j], dlo, dhi);
            const uint64_t hb = __ballot(dh[j] != 0);
            const uint64_t hdiff = dh[j] ? ~hb : hb;                                  // lanes whose half differs from mine
            dlo |= (uint32_t)hdiff; dhi |= (uint32_t)(hdiff >> 32);
            const uint32_t rnk = (uint32_t)lane - __builtin_amdgcn_mbcnt_hi(dhi, __builtin_amdgcn_mbcnt_lo(dlo, 0u));
            const uint32_t packed = ref[wave][dw[j]];
            const uint32_t want = (dh[j] ? (packed >> 16) : (packed & 0xFFFFu)) + rnk;
            const uint32_t mine = dh[j] ? (got[j] >> 16) : (got[j] & 0xFFFFu);
            bad += (mine != want);
            wg_barrier();   // (every lane has read ref before the group leaders update it; whole workgroup in lockstep)
            if (rnk == 0) atomicAdd(&ref[wave][dw[j]], (64u - (uint32_t)(__popc(dlo) + __popc(dhi))) << (dh[j] ? 16 : 0));
            wg_barrier();
        }
    }
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off, 64);
    if (lane == 0) { atomicAdd(&res[0], bad); atomicAdd(&res[1], 1u); }
}

// ---- order guard for the hardware-ordered ranking ---------------------------------------------------------
// The probe above is evidence, not a guarantee (the ISA does not promise the lane order of same-address returning adds).  The
// guard checks a finished ranking directly: along every row (key[rank[r]], rank[r]) < (key[rank[r + 1]], rank[r + 1]) under the
// canonical order, every index in range.  A ranking produced by a stable sort that lost its stability is still a permutation,
// so this one pass (one gather of the distances through the ranks) detects exactly the failure mode.  Offending rows are listed
// (bad[0] = count, bad[1 + i] = row); se_rank_rows re-ranks them with the ballot kernel and stops using the hardware-ordered
// kernel on that device.  The gather is random 4-byte reads: 50 ms for all of 50k x 50k (1 us per row of 50k), five times the
// ranking itself.  So: behind the FIRST hardware-ordered call of a process 512 evenly spaced rows are checked (0.5 ms; a broken
// lane order is systematic -- every wave step with two equal digits is affected -- not a property of single rows) and any
// violation makes the whole call fall back; under SE_RANK_CHECK=1 every row of every call is checked and only the offending
// rows are redone; se_rank_rows_check audits any ranking in full (bench.py reports its verdict outside the timed region).
constexpr int RC_THREADS = 512, RC_PER = 8;
template <int IDX64>   // index width code: 0 int32, 1 int64, 2 uint16
__global__ __launch_bounds__(RC_THREADS) void rank_check_kernel(const float *__restrict__ pdist, int64_t ldp, int64_t Q, int N, const void *rank,
                                                               int64_t ldr, uint32_t *__restrict__ bad, int cap, int64_t row_stride)
{
    __shared__ uint32_t wg_bad;
    for (int64_t row = (int64_t)blockIdx.x * row_stride; row < Q; row += (int64_t)gridDim.x * row_stride) {
        const float *drow = pdist + row * ldp;
        if (threadIdx.x == 0) wg_bad = 0;
        wg_barrier();
        uint32_t mine = 0;
        for (int r0 = threadIdx.x * RC_PER; r0 < N; r0 += RC_THREADS * RC_PER) {
            uint32_t pk = 0, pi = 0;
#pragma unroll
            for (int e = 0; e <= RC_PER; e++) {            // RC_PER consecutive ranks + the first of the next thread's
                const int r = r0 + e;
                if (r >= N) break;
                const int64_t iv = IDX64 == 2 ? (int64_t)((const uint16_t *)rank)[row * ldr + r]
                                              : (IDX64 ? ((const int64_t *)rank)[row * ldr + r] : (int64_t)((const int32_t *)rank)[row * ldr + r]);
                const bool in = iv >= 0 && iv < N;
                const uint32_t i = in ? (uint32_t)iv : 0u;
                const uint32_t k = canon_key(drow[i]);
                if (!in) mine = 1;
                if (e > 0 && !(pk < k || (pk == k && pi < i))) mine = 1;
                pk = k; pi = i;
            }
        }
        if (mine) atomicOr(&wg_bad, 1u);
        wg_barrier();
        if (threadIdx.x == 0 && wg_bad) {
            const uint32_t slot = atomicAdd(&bad[0], 1u);
            if ((int)slot < cap) bad[1 + slot] = (uint32_t)row;
        }
        wg_barrier();
    }
}

// -DSE_TUNING build, SE_RANK_INJECT=1: swap two adjacent ranks in every 7th row (what a lost stability would look like) so that
// the guard's detection and repair can be tested
__global__ void rank_inject_kernel(void *rank, int idx64, int64_t ldr, int64_t Q, int N)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 7 + 3;
    if (row >= Q || N < 4) return;
    const int r = (int)((row * 37) % (N - 1));
    if (idx64 == 2) { uint16_t *p = (uint16_t *)rank + row * ldr + r; const uint16_t t = p[0]; p[0] = p[1]; p[1] = t; }
    else if (idx64) { int64_t *p = (int64_t *)rank + row * ldr + r; const int64_t t = p[0]; p[0] = p[1]; p[1] = t; }
    else { int32_t *p = (int32_t *)rank + row * ldr + r; const int32_t t = p[0]; p[0] = p[1]; p[1] = t; }
}


static int rank_check_launch(const float *pdist, int64_t ldp, int64_t q, int n, const void *rank, int idx64, int64_t ldr, uint32_t *bad, int cap,
                             int64_t row_stride, hipStream_t s)
{
    SE_HIP_CHECK(hipMemsetAsync(bad, 0, 4, s));
    const int64_t nrows = (q + row_stride - 1) / row_stride;
    const int64_t grid = nrows < 2048 ? nrows : 2048;
    if (idx64 == 2) hipLaunchKernelGGL(rank_check_kernel<2>, dim3((unsigned)grid), dim3(RC_THREADS), 0, s, pdist, ldp, q, n, rank, ldr, bad, cap, row_stride);
    else if (idx64) hipLaunchKernelGGL(rank_check_kernel<1>, dim3((unsigned)grid), dim3(RC_THREADS), 0, s, pdist, ldp, q, n, rank, ldr, bad, cap, row_stride);
    else hipLaunchKernelGGL(rank_check_kernel<0>, dim3((unsigned)grid), dim3(RC_THREADS), 0, s, pdist, ldp, q, n, rank, ldr, bad, cap, row_stride);
    SE_LAUNCH_CHECK();
    return SE_OK;
}

static std::atomic<int> rr_hw_state[64];   // per device: 0 unknown, 1 verified, -1 refuted (zero-initialised; two threads racing on the
                                           // probe both run it on their own workspace and store the same verdict)
static std::atomic<int> rr_checked[64];    // per device: the guard has run once behind a hardware-ordered ranking

// 1 = the hardware-ordered kernel may be used on the current device, 0 = it may not.  The first call per device
// runs the probe (needs 256 bytes of caller workspace, synchronises the stream once); SE_RANK_SAFE=1 forces 0.
static int rank_hw_order_ok(void *workspace, int64_t workspace_bytes, hipStream_t s)
{
    static const bool forced_safe = getenv("SE_RANK_SAFE") != nullptr;
    if (forced_safe) return 0;
    std::atomic<int> *state = rr_hw_state;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (state[dev].load(std::memory_order_acquire) == 0) {
        if (!workspace || workspace_bytes < 256) return 0;   // cannot probe without scratch: stay on the safe kernel
        uint32_t *res = (uint32_t *)workspace, h[2] = {1u, 0u};
        if (hipMemsetAsync(res, 0, 8, s) != hipSuccess) return 0;
        hipLaunchKernelGGL(rank_order_probe_kernel, dim3(RR_PROBE_BLOCKS), dim3(RR_THREADS), 0, s, res);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess ||
            hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
            return 0;
        const int verdict = (h[0] == 0u && h[1] == (uint32_t)(RR_PROBE_BLOCKS * RR_WAVES)) ? 1 : -1;
        state[dev].store(verdict, std::memory_order_release);
        if (getenv("SE_RANK_VERBOSE"))
            fprintf(stderr, "[se_rank_rows] LDS returning-add order probe on device %d: %u mismatches, %u waves -> %s kernel\n", dev,
                    h[0], h[1], verdict == 1 ? "hardware-ordered" : "ballot");
    }
    return state[dev].load(std::memory_order_acquire) == 1;
}

// ---- se_rank_rows_init: the ONE synchronising entry point of the ranking --------------------------------------------------------
// Probe (rank_order_probe_kernel) + a self-test that ranks crafted rows through EVERY hardware-ordered kernel variant -- short
// instantiation (11 + 11 + 10-bit digits) plain / group-peeling, long instantiation (10 + 10 + 12 bits, aliased counters) plain /
// peeling / two-pass, and the segment-run build + merge -- and audits every row of every result with rank_check_kernel.  The rows
// are made of few distinct values (thousands of exact ties per row: stability is what is being tested), mixed signs for the
// three-pass variants and one binade for the two-pass one.  Afterwards the per-device verdict is cached and se_rank_rows neither
// probes, guards nor synchronises again (SE_RANK_CHECK=1 still audits every call).
constexpr int RI_ROWS = 4, RI_N_SHORT = 4000, RI_N_LONG = 36000, RI_N_SEG = 70000;

__global__ void rank_selftest_fill_kernel(float *pd, int64_t ld, int rows, int n, int mode)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)rows * n) return;
    const int r = (int)(g / n), c = (int)(g - (int64_t)r * n);
    uint32_t h = (uint32_t)c * 2654435761u + (uint32_t)(r + 1) * 40503u + (uint32_t)mode * 97u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    float v;
    if (mode == 2) v = 1.0f + (float)(h % 3001u) * (1.0f / 8192.0f);              // one binade: the two-pass window holds every key
    else if (mode == 3) {
        // image path: mixed signs on a 2^-24 grid around zero below one key of magnitude 1 (c = 2: 2 / 4 grid points share an image).  Row 0 / 1:
        // a few hundred / thousand colliding pairs (repaired in place); row 2: more runs than the worklist holds; row 3: runs above the cap -- both re-sorted
        const uint32_t span = 1u << (22 - 2 * (r & 3));
        v = ((float)(h % span) - (float)(span / 2)) * (1.0f / 16777216.0f);
        if (c == 17) v = -1.0f;
    }
    else v = ((float)(h % 2001u) - 1000.0f) * (1.0f / 1024.0f);                     // mixed signs, +-0 included (canon: -0 == +0)
    if (mode < 2 && (h >> 20) % 97u == 0) v = -0.0f;
    if (mode == 1 && c % 1013 == 5) v = __builtin_nanf("");                         // a few NaN keys: sorted last, index order
    pd[r * ld + c] = v;
}

extern "C" int64_t se_rank_rows_init_workspace_bytes(void)
{
    const int64_t mat = (int64_t)RI_ROWS * (RI_N_SEG + 8) * 4;
    return 4096 + 4 * (RC_CAP + 2) + 2 * mat + (rank_runs_ok(RI_N_SEG) ? rank_runs_bytes(RI_ROWS, RI_N_SEG) : 0) + 1024;
}

extern "C" int se_rank_rows_init(void *workspace, int64_t workspace_bytes, se_stream_t stream)
{
    if (!workspace || workspace_bytes < se_rank_rows_init_workspace_bytes())
        return fail(SE_ERR_WORKSPACE, "se_rank_rows_init: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)se_rank_rows_init_workspace_bytes());
    if ((((uintptr_t)workspace) & 255) != 0) return fail(SE_ERR_INVALID, "se_rank_rows_init: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(SE_ERR_HIP, "se_rank_rows_init: no current device");
    if (!rank_hw_order_ok(workspace, workspace_bytes, s)) {
        // SE_RANK_SAFE, or the probe refuted the lane order: ballot kernels, nothing to audit.  A probe that could NOT run (stream under
        // capture, a failed synchronise / copy) leaves the verdict open: the device must not count as audited then -- a later se_rank_rows
        // could probe successfully and would run the hardware-ordered kernels with neither this self-test nor its first-call guard.
        static const bool forced_safe = getenv("SE_RANK_SAFE") != nullptr;
        if (!forced_safe && rr_hw_state[dev].load(std::memory_order_acquire) == 0)
            return fail(SE_ERR_HIP, "se_rank_rows_init: the capability probe could not run on this stream (under capture?); the device stays unaudited");
        rr_checked[dev].store(1, std::memory_order_release);
        return SE_OK;
    }
    char *w = (char *)workspace + 4096;
    uint32_t *bad = (uint32_t *)w;                               w += (4 * (RC_CAP + 2) + 255) / 256 * 256;
    const int64_t ld = RI_N_SEG + 8;
    float *pd = (float *)w;                                      w += (RI_ROWS * ld * 4 + 255) / 256 * 256;
    int32_t *rk = (int32_t *)w;                                  w += (RI_ROWS * ld * 4 + 255) / 256 * 256;
    void *runs_ws = w;
    uint32_t total_bad = 0;
    const bool verbose = getenv("SE_RANK_VERBOSE") != nullptr;
    auto audit = [&](const char *what, int n, int rc) -> int {
        if (rc != SE_OK) return rc;
        if (const int rc2 = rank_check_launch(pd, ld, RI_ROWS, n, rk, 0, ld, bad, RC_CAP, 1, s)) return rc2;
        uint32_t h = 0;
        SE_HIP_CHECK(hipStreamSynchronize(s));
        SE_HIP_CHECK(hipMemcpy(&h, bad, sizeof(h), hipMemcpyDeviceToHost));
        if (verbose || h) fprintf(stderr, "[se_rank_rows_init] device %d, %s: %u of %d rows out of canonical order\n", dev, what, h, RI_ROWS);
        total_bad += h;
        return SE_OK;
    };
    auto fill = [&](int n, int mode) {
        hipLaunchKernelGGL(rank_selftest_fill_kernel, dim3((unsigned)(((int64_t)RI_ROWS * n + 255) / 256)), dim3(256), 0, s, pd, ld, RI_ROWS, n, mode);
    };
    constexpr int IS = (RI_N_SHORT + RR_THREADS - 1) / RR_THREADS <= 8 ? 8 : 12;          // short instantiation that holds RI_N_SHORT
#if SE_RR_THREADS == 512
    constexpr int IL = 72;                                                                // long instantiation that holds RI_N_LONG (and the 35,000-column segments)
#else
    constexpr int IL = 70;
#endif
    static_assert((int64_t)IL * RR_THREADS >= RI_N_LONG, "self-test row does not fit its instantiation");
    int rc;
    fill(RI_N_SHORT, 0);
    if ((rc = audit("short rows, plain", RI_N_SHORT, launch_rank_reg_variant<IS, true, 0>(pd, ld, RI_ROWS, RI_N_SHORT, rk, 0, ld, nullptr, s)))) return rc;
    fill(RI_N_SHORT, 1);
    if ((rc = audit("short rows, group-peeling", RI_N_SHORT, launch_rank_reg_variant<IS, true, 1>(pd, ld, RI_ROWS, RI_N_SHORT, rk, 0, ld, nullptr, s)))) return rc;
    fill(RI_N_LONG, 0);
    if ((rc = audit("long rows, plain", RI_N_LONG, launch_rank_reg_variant<IL, true, 0>(pd, ld, RI_ROWS, RI_N_LONG, rk, 0, ld, nullptr, s)))) return rc;
    fill(RI_N_LONG, 1);
    if ((rc = audit("long rows, group-peeling", RI_N_LONG, launch_rank_reg_variant<IL, true, 1>(pd, ld, RI_ROWS, RI_N_LONG, rk, 0, ld, nullptr, s)))) return rc;
#if SE_RR_TWO
    fill(RI_N_LONG, 2);
    if ((rc = audit("long rows, two-pass", RI_N_LONG, launch_rank_reg_variant<IL, true, 2>(pd, ld, RI_ROWS, RI_N_LONG, rk, 0, ld, nullptr, s)))) return rc;
#endif
#if SE_RR_IMG && SE_RR_THREADS == 512
    fill(RI_N_LONG, 3);
    if ((rc = audit("long rows, image path", RI_N_LONG, launch_rank_reg_variant<IL, true, 3>(pd, ld, RI_ROWS, RI_N_LONG, rk, 0, ld, nullptr, s)))) return rc;
#endif
#if SE_RR_THREADS == 512
    if (rank_runs_ok(RI_N_SEG) && rank_runs_items(RI_N_SEG) == IL) {
        fill(RI_N_SEG, 1);
        if ((rc = audit("segment runs + merge", RI_N_SEG, launch_rank_runs<IL>(pd, ld, RI_ROWS, RI_N_SEG, rk, 0, ld, runs_ws, s)))) return rc;
    }
#endif
    if (total_bad) {
        rr_hw_state[dev].store(-1, std::memory_order_release);
        fprintf(stderr, "[se_rank_rows_init] device %d leaves the hardware-ordered ranking kernels (ballot / tiled kernels from now on)\n", dev);
    }
    rr_checked[dev].store(1, std::memory_order_release);
    return SE_OK;
}

extern "C" int64_t se_rank_rows_workspace_bytes(int64_t q, int64_t n)
{
    if (q <= 0 || n <= 0) return 0;
    if (!rank_use_tiled(n)) return 256 + 4 * (RC_CAP + 2);   // register-resident kernel: capability probe / detector words + the order guard's row list
    const int64_t tiled = (int64_t)rank_grid(q) * 4 * rank_npad(n) * (int64_t)sizeof(uint32_t);
    if (!rank_runs_ok(n)) return tiled;
    const int64_t runs = rank_runs_bytes(q, n);              // sorted runs + merge; the tiled kernel stays the fallback (same buffer)
    return runs > tiled ? runs : tiled;
}

extern "C" int se_rank_rows(const float *pdist, int64_t ldp, int64_t q, int64_t n, void *rank, int idx64,
                            int64_t ldr, void *workspace, int64_t workspace_bytes, se_stream_t stream)
{
    if (q < 0 || n < 0 || n > 0x7FFFFFFFll) return fail(SE_ERR_INVALID, "se_rank_rows: bad shape q=%lld n=%lld", (long long)q, (long long)n);
    if (q == 0 || n == 0) return SE_OK;
    if (!pdist || !rank || ldp < n || ldr < n) return fail(SE_ERR_INVALID, "se_rank_rows: bad argument");
    if (idx64 < 0 || idx64 > 2) return fail(SE_ERR_INVALID, "se_rank_rows: index width code %d (0 int32, 1 int64, 2 uint16)", idx64);
    if (idx64 == 2 && (n > RR_MAX_N || rank_use_tiled(n)))
        return fail(SE_ERR_UNSUPPORTED, "se_rank_rows: uint16 ranks are written by the register-resident kernel only (rows of at most %d columns)", RR_MAX_N);
    hipStream_t s = (hipStream_t)stream;
    if (!rank_use_tiled(n)) {
        const int items = (int)((n + RR_THREADS - 1) / RR_THREADS);
        bool hw = rank_hw_order_ok(workspace, workspace_bytes, s) != 0;
        void *scratch = (workspace && workspace_bytes >= 256) ? workspace : nullptr;
        // order guard (see rank_check_kernel): behind the first hardware-ordered ranking of a process that did not call
        // se_rank_rows_init, and under SE_RANK_CHECK=1
        int dev = 0;
        const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
        static const bool check_always = getenv("SE_RANK_CHECK") != nullptr && getenv("SE_RANK_CHECK")[0] != '0';
        const bool audited = have_dev && rr_checked[dev].load(std::memory_order_acquire) != 0;
        const bool can_guard = have_dev && workspace && workspace_bytes >= 256 + 4 * (RC_CAP + 2);
        if (hw && !audited && !can_guard) hw = false;   // never run the hardware-ordered kernel unaudited: no room for the guard -> ballot kernel
        const bool guard = hw && can_guard && (check_always || !audited);
        int rc = SE_ERR_INVALID;
#define SE_RR_CASE(I) if (rc == SE_ERR_INVALID && items <= I) rc = launch_rank_reg<I>(pdist, ldp, q, (int)n, rank, idx64, ldr, hw, scratch, s);
#if SE_RR_THREADS == 512
        SE_RR_CASES_512
#else
        SE_RR_CASE(2) SE_RR_CASE(6) SE_RR_CASE(14) SE_RR_CASE(28) SE_RR_CASE(44) SE_RR_CASE(56) SE_RR_CASE(66) SE_RR_CASE(70)
#endif
#undef SE_RR_CASE
        if (rc != SE_OK || !guard) return rc;
        if (kTuning && tuning_env("SE_RANK_INJECT")) {
            hipLaunchKernelGGL(rank_inject_kernel, dim3((unsigned)((q / 7 + 256) / 256)), dim3(256), 0, s, rank, idx64, ldr, q, (int)n);
            SE_LAUNCH_CHECK();
        }
        uint32_t *bad = (uint32_t *)((char *)workspace + 256);
        const int64_t row_stride = check_always ? 1 : (q > 512 ? q / 512 : 1);      // first call: a sample of the rows
        if (const int rc2 = rank_check_launch(pdist, ldp, q, (int)n, rank, idx64, ldr, bad, RC_CAP, row_stride, s)) return rc2;
        uint32_t h[RC_CAP + 1];
        SE_HIP_CHECK(hipStreamSynchronize(s));
        SE_HIP_CHECK(hipMemcpy(h, bad, sizeof(uint32_t), hipMemcpyDeviceToHost));
        rr_checked[dev].store(1, std::memory_order_release);
        if (h[0] == 0) return SE_OK;
        // the hardware-ordered ranking broke its own invariant: never use it again on this device, repair with the ballot kernel
        rr_hw_state[dev].store(-1, std::memory_order_release);
        fprintf(stderr, "[se_rank_rows] order guard: %u of %lld rows out of canonical order behind the hardware-ordered kernel -- re-ranking them "
                        "with the ballot kernel; device %d uses the ballot kernel from now on\n", h[0], (long long)q, dev);
        const uint32_t nbad = h[0];
        if (nbad > (uint32_t)RC_CAP || row_stride > 1) {   // more than the list holds, or only a sample was looked at: redo the whole call
#define SE_RR_CASE(I) if (items <= I) return launch_rank_reg<I>(pdist, ldp, q, (int)n, rank, idx64, ldr, false, scratch, s);
#if SE_RR_THREADS == 512
            SE_RR_CASES_512
#else
            SE_RR_CASE(2) SE_RR_CASE(6) SE_RR_CASE(14) SE_RR_CASE(28) SE_RR_CASE(44) SE_RR_CASE(56) SE_RR_CASE(66) SE_RR_CASE(70)
#endif
#undef SE_RR_CASE
        }
        SE_HIP_CHECK(hipMemcpy(h + 1, bad + 1, nbad * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const size_t esz = rank_idx_bytes(idx64);
        for (uint32_t i = 0; i < nbad; i++) {
            const int64_t row = h[1 + i];
            void *rrow = (char *)rank + (size_t)row * (size_t)ldr * esz;
            int rc3 = SE_ERR_INVALID;
#define SE_RR_CASE(I) if (rc3 == SE_ERR_INVALID && items <= I) rc3 = launch_rank_reg<I>(pdist + row * ldp, ldp, 1, (int)n, rrow, idx64, ldr, false, nullptr, s);
#if SE_RR_THREADS == 512
            SE_RR_CASES_512
#else
            SE_RR_CASE(2) SE_RR_CASE(6) SE_RR_CASE(14) SE_RR_CASE(28) SE_RR_CASE(44) SE_RR_CASE(56) SE_RR_CASE(66) SE_RR_CASE(70)
#endif
#undef SE_RR_CASE
            if (rc3 != SE_OK) return rc3;
        }
        return SE_OK;
    }
    const int64_t need = se_rank_rows_workspace_bytes(q, n);
    if (!workspace || workspace_bytes < need) return fail(SE_ERR_WORKSPACE, "se_rank_rows: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    if ((((uintptr_t)workspace) & 15) != 0) return fail(SE_ERR_INVALID, "se_rank_rows: workspace must be 16-byte aligned");
    if (rank_runs_ok(n) && rank_hw_order_ok(workspace, workspace_bytes, s)) {
        // RR_MAX_N < n <= 8 RR_MAX_N: 2 / 4 / 8 sorted runs per row (hardware-ordered register-resident kernel on the segments) + merge tree
        const int items = rank_runs_items(n);
        int rc = SE_ERR_INVALID;
#ifdef SE_RR_DEV
        if (items != 98) return fail(SE_ERR_INVALID, "se_rank_rows: -DSE_RR_DEV build");
        rc = launch_rank_runs<98>(pdist, ldp, q, (int)n, rank, idx64, ldr, workspace, s);
#else
        if (items == 64) rc = launch_rank_runs<64>(pdist, ldp, q, (int)n, rank, idx64, ldr, workspace, s);
        else if (items == 72) rc = launch_rank_runs<72>(pdist, ldp, q, (int)n, rank, idx64, ldr, workspace, s);
        else if (items == 80) rc = launch_rank_runs<80>(pdist, ldp, q, (int)n, rank, idx64, ldr, workspace, s);
        else if (items == 88) rc = launch_rank_runs<88>(pdist, ldp, q, (int)n, rank, idx64, ldr, workspace, s);
        else if (items == 98) rc = launch_rank_runs<98>(pdist, ldp, q, (int)n, rank, idx64, ldr, workspace, s);
        else rc = launch_rank_runs<104>(pdist, ldp, q, (int)n, rank, idx64, ldr, workspace, s);
#endif
        if (rc != SE_OK) return rc;
        int dev = 0;
        const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
        static const bool check_always = getenv("SE_RANK_CHECK") != nullptr && getenv("SE_RANK_CHECK")[0] != '0';
        if (!have_dev || !(check_always || rr_checked[dev].load(std::memory_order_acquire) == 0)) return SE_OK;
        if (kTuning && tuning_env("SE_RANK_NOGUARD")) return SE_OK;   // -DSE_TUNING build only: look at the raw output of the run kernels
        if (kTuning && tuning_env("SE_RANK_INJECT")) {
            hipLaunchKernelGGL(rank_inject_kernel, dim3((unsigned)((q / 7 + 256) / 256)), dim3(256), 0, s, rank, idx64, ldr, q, (int)n);
            SE_LAUNCH_CHECK();
        }
        // order guard, as for the short rows: a sample of the rows behind the first hardware-ordered call of the process (every row
        // under SE_RANK_CHECK=1); on a violation the whole call is redone by the tiled kernel below and the device leaves the fast paths
        uint32_t *bad = (uint32_t *)((char *)workspace + 256);
        const int64_t row_stride = check_always ? 1 : (q > 512 ? q / 512 : 1);
        if (const int rc2 = rank_check_launch(pdist, ldp, q, (int)n, rank, idx64, ldr, bad, RC_CAP, row_stride, s)) return rc2;
        uint32_t nbad = 0;
        SE_HIP_CHECK(hipStreamSynchronize(s));
        SE_HIP_CHECK(hipMemcpy(&nbad, bad, sizeof(uint32_t), hipMemcpyDeviceToHost));
        rr_checked[dev].store(1, std::memory_order_release);
        if (nbad == 0) return SE_OK;
        rr_hw_state[dev].store(-1, std::memory_order_release);
        fprintf(stderr, "[se_rank_rows] order guard: %u of %lld rows out of canonical order behind the hardware-ordered run kernel -- re-ranking the "
                        "call with the tiled kernel; device %d leaves the hardware-ordered paths\n", nbad, (long long)q, dev);
    }
    const size_t lds = sizeof(RankLds);
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

extern "C" int64_t se_rank_rows_check_workspace_bytes(void) { return 4 * (RC_CAP + 2); }

extern "C" int se_rank_rows_check(const float *pdist, int64_t ldp, int64_t q, int64_t n, const void *rank, int idx64, int64_t ldr,
                                  void *workspace, int64_t workspace_bytes, int64_t *bad_rows_host, se_stream_t stream)
{
    if (q < 0 || n < 0 || n > 0x7FFFFFFFll) return fail(SE_ERR_INVALID, "se_rank_rows_check: bad shape");
    if (bad_rows_host) *bad_rows_host = 0;
    if (q == 0 || n == 0) return SE_OK;
    if (!pdist || !rank || ldp < n || ldr < n || !bad_rows_host) return fail(SE_ERR_INVALID, "se_rank_rows_check: bad argument");
    if (idx64 < 0 || idx64 > 2 || (idx64 == 2 && n > 65536)) return fail(SE_ERR_INVALID, "se_rank_rows_check: index width code %d for %lld columns", idx64, (long long)n);
    if (!workspace || workspace_bytes < se_rank_rows_check_workspace_bytes()) return fail(SE_ERR_WORKSPACE, "se_rank_rows_check: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    uint32_t *bad = (uint32_t *)workspace;
    if (const int rc = rank_check_launch(pdist, ldp, q, (int)n, rank, idx64, ldr, bad, RC_CAP, 1, s)) return rc;
    uint32_t h = 0;
    SE_HIP_CHECK(hipStreamSynchronize(s));
    SE_HIP_CHECK(hipMemcpy(&h, bad, sizeof(h), hipMemcpyDeviceToHost));
    *bad_rows_host = (int64_t)h;
    return SE_OK;
}
