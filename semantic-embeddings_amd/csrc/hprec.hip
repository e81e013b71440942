// hprec.hip -- hierarchical retrieval metrics on the device (the consumer of the rankings).
//
// Replaces the per-query loop of `ClassHierarchy.hierarchical_precision` (class_hierarchy.py:211-316): for every
// query, class similarities gathered along its ranking (Wu-Palmer and 1 - LCS height), hierarchical precision at
// the cut-offs `ks`, the area under that curve (AHP, whole list or clipped) and average precision.  The class x
// class similarity tables and the best-possible cumulative similarity per query class (class_hierarchy.py:262-276:
// the descending-sorted similarities of the whole gallery, cumulated) are computed once on the host and passed in;
// the 0.6 h the reference spends here at N = 50k is the gather + prefix sums over Q x N ranks, which is this kernel.
//
// Persistent 256-thread workgroups (two per CU) take one query at a time and walk its ranking in 4096-rank chunks (thread t owns
// 16 consecutive ranks): rank -> class (a byte / 16-bit copy of the gallery's classes in LDS, filled once per workgroup) ->
// similarity pair (LDS), float64 running sums by a DPP wave scan + ONE barrier per chunk (wave totals double-buffered in LDS,
// the carry replicated in every thread), per-thread accumulation of the trapezoid / AP terms (reduced once per query).  Chunks
// come in three forms: interior ones (every rank live, behind the query, inside the AHP window, no cut-off: 2 adds + 2 FMAs per
// rank and nothing else), the one the list ends in (masked, otherwise the same), and the general form (first chunk: the query's
// own position, the cut-offs -- sorted once per workgroup and merged with each thread's consecutive ranks -- and the end points).
// What the memory system sees per rank: 4 B of the ranking (HBM, requested a chunk ahead) and 16 B of the query class's best
// curve.  The curve enters as the pre-divided, chunk-transposed table of `se_hprec_reciprocal_curves` -- 1 / (best[i] - 1) for
// the ranks behind the query, 1 / best[i] for those ahead of it, laid out so that a wave's loads are contiguous -- which turns
// two float64 divisions per rank into two multiplications; and the queries are visited in class order (`order_ws`: a counting
// sort of `qcls`, one contiguous segment of that order per XCD, handed out by per-XCD atomic counters, the next draw in flight
// while a query is processed) so that the workgroups resident on an XCD stream the SAME 16 N-byte curve and it stays in that
// XCD's 4 MB L2 instead of coming from HBM once per query.  Measured bound (phase profile, -DSE_HP_PROFILE=1): the interior
// chunks wait on their 20 B per rank of L2 -> L1 traffic (~9 TB/s aggregate); DESIGN.md section 5.3b has the numbers.
// Bookkeeping kept from the reference: the query itself is dropped from its ranking (position q_pos), which shifts
// the best curve left there and subtracts its self-similarity 1.0 (class_hierarchy.py:280-290); AHP is numpy's
// trapz of cum / best with dx = 1 / length; AP is the mean over the relevant ranks of precision at that rank.
// float64 throughout; sums are associated differently from numpy's sequential cumsum and cum / best is cum * (1 / best)
// (last-bits differences; the tests hold the outputs to 1e-10 of the reference's).
#include "se_common.h"

#include <type_traits>

namespace se {

#ifndef SE_HP_THREADS           // geometry of a chunk: threads x consecutive ranks per thread, and the occupancy the registers are
#define SE_HP_THREADS 256       // bounded for (waves per SIMD); tools/experiments/README.md has the measured alternatives
#define SE_HP_ITEMS 8          // (round 6: 8 ranks per thread and chunk -- two queries of a class share a pass, their state doubles)
#define SE_HP_WAVES_PER_SIMD 2
#endif
#ifndef SE_HP_PAIR
#define SE_HP_PAIR 1            // two queries of one class per workgroup pass (0: one, the round-5 kernel)
#endif
#ifndef SE_HP_PF
#define SE_HP_PF 1              // chunks of rank look-ahead
#endif
#ifndef SE_HP_PROFILE
#define SE_HP_PROFILE 0         // experiment build: shader-clock cycles per phase of wave 0 of every workgroup, printed after each launch
#endif
#if SE_HP_PROFILE
__device__ unsigned long long hp_prof[16];
#define HP_T(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (tid == 0) hp_acc[i] += now_ - hp_last; hp_last = now_; }
#else
#define HP_T(i) {}
#endif
constexpr int HP_THREADS = SE_HP_THREADS;
constexpr int HP_WAVES = HP_THREADS / WAVE;
constexpr int HP_ITEMS = SE_HP_ITEMS;
constexpr int HP_PF = SE_HP_PF;
constexpr int HP_CHUNK = HP_THREADS * HP_ITEMS;
constexpr int HP_MAX_KS = 512;
constexpr int HP_WS_HEAD = 16;      // order_ws (ints): [0, 8) per-XCD cursors, then one int4 per query in class order: (query, its class, its own gallery index or -1, 0)
constexpr int HP_ORDER_THREADS = 1024;

// ---- DPP wave scans (no LDS traffic): Kogge-Stone inside the 16-lane rows, then the row totals by row_bcast ----
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xf, true); }

template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = dpp_i32<CTRL, ROWS>(__double2loint(v)), hi = dpp_i32<CTRL, ROWS>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_incl_scan_f64(double v)
{
    v += dpp_f64<0x111, 0xf>(v);   // row_shr:1
    v += dpp_f64<0x112, 0xf>(v);   // row_shr:2
    v += dpp_f64<0x114, 0xf>(v);   // row_shr:4
    v += dpp_f64<0x118, 0xf>(v);   // row_shr:8
    v += dpp_f64<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp_f64<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ int wave_incl_scan_i32(int v)
{
    v += dpp_i32<0x111, 0xf>(v);
    v += dpp_i32<0x112, 0xf>(v);
    v += dpp_i32<0x114, 0xf>(v);
    v += dpp_i32<0x118, 0xf>(v);
    v += dpp_i32<0x142, 0xa>(v);
    v += dpp_i32<0x143, 0xc>(v);
    return v;
}

// 1 / x for x = 1 .. 2^31 (AP: precision at a relevant rank): v_rcp_f64 + two Newton steps instead of the IEEE division sequence
__device__ __forceinline__ double fast_rcp_f64(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

// slot of original position i inside a class row of the reciprocal table: chunk k = i / 2048, thread t = (i % 2048) / 8,
// element e = i % 8  ->  (k * 8 + e) * 256 + t   (for a fixed e the 256 threads of a chunk read 256 consecutive double2)
__device__ __host__ __forceinline__ int64_t hp_slot(int64_t i)
{
    const int64_t k = i / HP_CHUNK, p = i % HP_CHUNK;
    return (k * HP_ITEMS + (p % HP_ITEMS)) * HP_THREADS + p / HP_ITEMS;
}

__global__ __launch_bounds__(HP_THREADS) void hprec_rcp_kernel(const double *__restrict__ bw, const double *__restrict__ bl, int64_t ldb,
                                                               int64_t len, int64_t lp, double2 *__restrict__ out)
{
    const int64_t c = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * HP_THREADS + threadIdx.x; i < lp; i += (int64_t)gridDim.x * HP_THREADS) {
        double2 behind = make_double2(0.0, 0.0), ahead = behind;
        if (i < len) {
            const double w = bw[c * ldb + i], l = bl[c * ldb + i];
            behind = make_double2(1.0 / (w - 1.0), 1.0 / (l - 1.0));     // ranks behind the query: curve shifted, self-similarity removed
            ahead = make_double2(1.0 / w, 1.0 / l);                      // ranks ahead of it: the curve as it is
        }
        out[c * 2 * lp + hp_slot(i)] = behind;
        out[(c * 2 + 1) * lp + hp_slot(i)] = ahead;
    }
}

// Counting sort of the queries by class (any order inside a class) + the per-XCD cursors of hprec_kernel.
template <typename RT>
__global__ __launch_bounds__(HP_ORDER_THREADS) void hprec_order_kernel(const int32_t *__restrict__ qcls, const int32_t *__restrict__ qidx, int64_t Q, int C,
                                                                       const RT *__restrict__ rank, int64_t ldr, int32_t *__restrict__ ws)
{
    extern __shared__ int hp_hist[];            // [C] class counts -> cursors, then [16] wave totals
    int *s_wt = hp_hist + C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += HP_ORDER_THREADS) hp_hist[c] = 0;
    if (tid < HP_WS_HEAD) ws[tid] = 0;
    wg_barrier();
    for (int64_t i = tid; i < Q; i += HP_ORDER_THREADS) atomicAdd(&hp_hist[qcls[i]], 1);
    wg_barrier();
    const int per = (C + HP_ORDER_THREADS - 1) / HP_ORDER_THREADS, lo = tid * per, hi = (lo + per < C) ? lo + per : C;
    int mine = 0;
    for (int c = lo; c < hi; c++) mine += hp_hist[c];
    const int incl = wave_incl_scan_i32(mine);
    if (lane == 63) s_wt[wave] = incl;
    wg_barrier();
    int run = incl - mine;
    for (int w = 0; w < wave; w++) run += s_wt[w];
    for (int c = lo; c < hi; c++) { const int t = hp_hist[c]; hp_hist[c] = run; run += t; }
    wg_barrier();
    int4 *ent = reinterpret_cast<int4 *>(ws + HP_WS_HEAD);      // everything a workgroup needs to start a query, in one 16-byte load
    for (int64_t i = tid; i < Q; i += HP_ORDER_THREADS) {
        const int c = qcls[i];
        // flag: 1 = the query is the first entry of its own ranking (normally: its own nearest neighbour), 2 = it is not a gallery item,
        // 0 = its position has to be looked for -- queries of one class with equal non-zero flags are processed two per pass
        const int self = qidx ? qidx[i] : -1;
        ent[atomicAdd(&hp_hist[c], 1)] = make_int4((int)i, c, self, self < 0 ? 2 : ((int)rank[i * ldr] == self ? 1 : 0));
    }
}

// out row layout: [P@k WUP x nk][P@k LCS x nk][AHP WUP][AHP LCS][AP]
// CLSW: where the class of a ranked gallery item comes from -- 1 / 2: a byte / 16-bit copy of `cls` in LDS, filled once per
// (persistent) workgroup; 0: gathered from global memory (galleries too large for LDS).  The random 4-byte gather costs a
// 128-byte L2 -> L1 line per rank, 32x the ranking itself, and was what bounded the kernel before the LDS copy.
// RT: element type of the rankings -- int32_t, or uint16_t (se_rank_rows with idx64 == 2: one 16-byte load brings a thread's 8 ranks)
template <int CLSW, typename RT>
__global__ __launch_bounds__(HP_THREADS, SE_HP_WAVES_PER_SIMD) void hprec_kernel(const RT *__restrict__ rank, int64_t ldr, int64_t Q, int64_t L,
                                                              const int32_t *__restrict__ cls, int64_t gallery,
                                                              const int32_t *__restrict__ qcls, const int32_t *__restrict__ qidx,
                                                              const double *__restrict__ wup, const double *__restrict__ lcs, int C,
                                                              const double2 *__restrict__ rcp, int64_t ldc,
                                                              const int32_t *__restrict__ ks, int nk, int64_t ahp_len, int want_ap,
                                                              double *__restrict__ out, int64_t ldo, int32_t *__restrict__ order_ws, int vec_ok)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char hp_raw[];
    double2 *s_sim = reinterpret_cast<double2 *>(hp_raw);          // [C] (wup, lcs) similarity of the query class to every class
    double *s_part = reinterpret_cast<double *>(s_sim + C);        // [2][2][HP_WAVES][3] wave totals of the scans (two queries), double-buffered
    double *s_fin = s_part + 2 * 2 * HP_WAVES * 3;                 // [2][HP_WAVES][3] end-of-query reduction, then [2][4] trapezoid end points
    double *s_ends = s_fin + 2 * HP_WAVES * 3;
    int *s_ks = reinterpret_cast<int *>(s_ends + 8);               // [nk] the cut-offs, ascending
    int *s_perm = s_ks + nk;                                       // [nk] their slots in the output row
    int *s_qpos = s_perm + nk;
    int *s_next = s_qpos + 1;                                      // [9]: number of queries drawn, then (query, class, own index, flag) x 2
    unsigned char *s_cls8 = reinterpret_cast<unsigned char *>(s_qpos + 12);   // [gallery] class of every gallery item (CLSW 1 / 2)
    unsigned short *s_cls16 = reinterpret_cast<unsigned short *>(s_cls8);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

#if SE_HP_PROFILE
    unsigned long long hp_acc[16] = {}, hp_last = __builtin_amdgcn_s_memtime();
#endif
    // ---- once per workgroup: the gallery's classes into LDS, the cut-offs sorted (rank by counting; ties keep their order) ----
    if (CLSW == 1) {
        unsigned *w = reinterpret_cast<unsigned *>(s_cls8);
        const int64_t quads = (reinterpret_cast<uintptr_t>(cls) % 16 == 0) ? gallery / 4 : 0;     // 16-byte loads, 8 in flight per thread
        const int4 *c4 = reinterpret_cast<const int4 *>(cls);
#pragma unroll 8
        for (int64_t g = tid; g < quads; g += HP_THREADS) {
            const int4 v = c4[g];
            w[g] = ((unsigned)v.x & 0xFFu) | (((unsigned)v.y & 0xFFu) << 8) | (((unsigned)v.z & 0xFFu) << 16) | (((unsigned)v.w & 0xFFu) << 24);
        }
        for (int64_t i = quads * 4 + (int64_t)tid * 4; i < gallery; i += HP_THREADS * 4) {
            unsigned v = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) v |= (i + e < gallery ? (unsigned)cls[i + e] & 0xFFu : 0u) << (8 * e);
            w[i >> 2] = v;
        }
    } else if (CLSW == 2) {
        unsigned *w = reinterpret_cast<unsigned *>(s_cls16);
#pragma unroll 8
        for (int64_t i = (int64_t)tid * 2; i < gallery; i += HP_THREADS * 2)
            w[i >> 1] = ((unsigned)cls[i] & 0xFFFFu) | ((i + 1 < gallery ? (unsigned)cls[i + 1] & 0xFFFFu : 0u) << 16);
    }
    for (int s = tid; s < nk; s += HP_THREADS) s_perm[s] = ks[s];       // raw cut-offs (LDS copy), ranked below
    wg_barrier();
    int my_k[(HP_MAX_KS + HP_THREADS - 1) / HP_THREADS], my_r[(HP_MAX_KS + HP_THREADS - 1) / HP_THREADS];
#pragma unroll
    for (int v = 0; v < (HP_MAX_KS + HP_THREADS - 1) / HP_THREADS; v++) {
        const int s = tid + v * HP_THREADS;
        my_k[v] = s < nk ? s_perm[s] : 0;
        int r = 0;
        if (s < nk)
            for (int u = 0; u < nk; u++) { const int ku = s_perm[u]; r += (ku < my_k[v] || (ku == my_k[v] && u < s)) ? 1 : 0; }
        my_r[v] = r;
    }
    wg_barrier();
#pragma unroll
    for (int v = 0; v < (HP_MAX_KS + HP_THREADS - 1) / HP_THREADS; v++) {
        const int s = tid + v * HP_THREADS;
        if (s < nk) { s_ks[my_r[v]] = my_k[v]; s_perm[my_r[v]] = s; }
    }
    wg_barrier();
    const int kmax = nk > 0 ? s_ks[nk - 1] : 0;                    // cut-offs live at original positions <= kmax
    // the cut-offs the CLI asks for are 1, 2, ..., K in that order (evaluate_retrieval.py: range(1, plot_max + 1)): rank j then IS
    // slot j -- no look-up at all.  One vote per workgroup.
    bool iota = true;
    for (int s = tid; s < nk; s += HP_THREADS) iota = iota && (s_ks[s] == s + 1) && (s_perm[s] == s);
    const bool ks_iota = __syncthreads_and(iota ? 1 : 0) != 0;
    HP_T(0)

    // ---- which queries this workgroup takes: class order, one segment of it per XCD (blockIdx round-robins the XCDs), stealing
    //      from the next segments once its own is empty; or every gridDim-th query when no workspace was given ----
    const int xcd = blockIdx.x & 7;
    const int64_t seg_len = (Q + 7) / 8;
    int seg = 0, ticket = 0;                                       // thread 0: the segment it draws from and its next draw
    if (order_ws && tid == 0) ticket = atomicAdd(&order_ws[xcd], 2);   // (two consecutive entries per draw)
    int64_t q_static = blockIdx.x;
    const int Li = (int)L;                                         // positions are int32 (list_len < 2^31 - 8192 is checked at the entry point)
    // upper bound of the positions a query can need, known before its own position in the list is: the first ranks are requested
    // together with everything else a query starts with (one memory latency instead of four in a row)
    const int lp_bound = (want_ap || ahp_len == 0) ? Li : min(Li, max(kmax, ahp_len > 0 ? (int)ahp_len : 0) + 1);
    for (;;) {
        // ---- draw: up to TWO queries -- consecutive entries of the class order, i.e. mostly of one class ----
        int nq_drawn = 1;
        int64_t qv[2] = {0, 0};
        int qcv[2] = {0, 0}, flagv[2] = {0, 0};
        int32_t selfv[2] = {-1, -1};
        if (order_ws) {
            if (tid == 0) {   // resolve the draw made while the previous queries were being processed
                int n_got = 0;
                for (;;) {
                    const int xs = (xcd + seg) & 7;
                    const int64_t lo = xs * seg_len, hi = (lo + seg_len < Q) ? lo + seg_len : Q;
                    if (lo + ticket < hi) {
                        const int4 *ent = reinterpret_cast<const int4 *>(order_ws + HP_WS_HEAD);
                        const int4 e = ent[lo + ticket];
                        s_next[1] = e.x; s_next[2] = e.y; s_next[3] = e.z; s_next[4] = e.w;
                        n_got = 1;
                        if (lo + ticket + 1 < hi) {
                            const int4 f = ent[lo + ticket + 1];
                            s_next[5] = f.x; s_next[6] = f.y; s_next[7] = f.z; s_next[8] = f.w;
                            n_got = 2;
                        }
                        break;
                    }
                    if (++seg == 8) break;
                    ticket = atomicAdd(&order_ws[(xcd + seg) & 7], 2);
                }
                s_next[0] = n_got;
            }
            wg_barrier();
            nq_drawn = s_next[0];
            if (nq_drawn == 0) break;
            if (tid == 0 && seg < 8) ticket = atomicAdd(&order_ws[(xcd + seg) & 7], 2);   // the next draw: in flight during these queries
#pragma unroll
            for (int u = 0; u < 2; u++) { qv[u] = s_next[1 + 4 * u]; qcv[u] = s_next[2 + 4 * u]; selfv[u] = s_next[3 + 4 * u]; flagv[u] = s_next[4 + 4 * u]; }
        } else {
            if (q_static >= Q) break;
            qv[0] = q_static; qcv[0] = qcls[q_static]; selfv[0] = qidx ? qidx[q_static] : -1;
            q_static += gridDim.x;
        }
        // One pass over the ranks of NQ queries of ONE class whose own positions in their rankings agree (normally: both are their own
        // nearest neighbour): the 16 bytes of best curve per rank -- 4x the ranking's own bytes, what the kernel waits for -- are loaded
        // once and used for both.  Everything per query (ranks, sums, accumulators, end points, output row) is an array over u.
        auto run = [&](auto nq_c, const int64_t (&qq)[2], const int32_t (&selfq)[2], const int qc, const int qpos_known) {
            constexpr int NQ = decltype(nq_c)::value;
            const RT *rrow[NQ];
            double *orow[NQ];
#pragma unroll
            for (int u = 0; u < NQ; u++) { rrow[u] = rank + qq[u] * ldr; orow[u] = out + qq[u] * ldo; }
            const double2 *rc = rcp + (int64_t)qc * ldc;
            // The ranks a thread owns in a chunk are requested HP_PF chunks ahead (HP_PF register sets, refilled right after the barrier of
            // the chunk that consumed them).
            int r[NQ][HP_PF][HP_ITEMS];
            auto load_ranks = [&](int (&dst)[HP_ITEMS], const RT *row, int at, int bound) {
                if (vec_ok && at + HP_ITEMS <= bound) {
                    if constexpr (sizeof(RT) == 2) {
                        static_assert(HP_ITEMS % 8 == 0, "16-byte loads of 16-bit ranks");
#pragma unroll
                        for (int v = 0; v < HP_ITEMS / 8; v++) {
                            const uint4 a = *reinterpret_cast<const uint4 *>(row + at + 8 * v);
                            dst[8 * v] = (int)(a.x & 0xFFFFu); dst[8 * v + 1] = (int)(a.x >> 16); dst[8 * v + 2] = (int)(a.y & 0xFFFFu); dst[8 * v + 3] = (int)(a.y >> 16);
                            dst[8 * v + 4] = (int)(a.z & 0xFFFFu); dst[8 * v + 5] = (int)(a.z >> 16); dst[8 * v + 6] = (int)(a.w & 0xFFFFu); dst[8 * v + 7] = (int)(a.w >> 16);
                        }
                    } else {
#pragma unroll
                    for (int v = 0; v < HP_ITEMS / 4; v++) {
                        const int4 a = *reinterpret_cast<const int4 *>(row + at + 4 * v);
                        dst[4 * v] = a.x; dst[4 * v + 1] = a.y; dst[4 * v + 2] = a.z; dst[4 * v + 3] = a.w;
                    }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < HP_ITEMS; e++) dst[e] = (at + e < bound) ? (int)row[at + e] : 0;
                }
            };
#pragma unroll
            for (int u = 0; u < NQ; u++)
#pragma unroll
                for (int p = 0; p < HP_PF; p++) load_ranks(r[u][p], rrow[u], tid * HP_ITEMS + p * HP_CHUNK, lp_bound);
            const int32_t self = selfq[0];
            int32_t first = 0;
            if (NQ == 1 && qpos_known < 0) first = (int32_t)rrow[0][0];
            for (int c = tid; c < C; c += HP_THREADS) s_sim[c] = make_double2(wup[(int64_t)qc * C + c], lcs[(int64_t)qc * C + c]);
            if (tid == 0) *s_qpos = (qpos_known >= 0) ? qpos_known : ((self >= 0 && first == self) ? 0 : 0x7FFFFFFF);
            if (tid < 4 * NQ) s_ends[tid] = 0.0;
            wg_barrier();
            // ---- position of the query in its own ranking (first hit; L if absent): normally it is its own nearest neighbour (rank 0) ----
            if (NQ == 1 && qpos_known < 0 && self >= 0 && first != self) {   // otherwise chunk by chunk with a uniform early exit
                for (int64_t b0 = 0; b0 < L; b0 += 4 * HP_THREADS) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int64_t i = b0 + e * HP_THREADS + tid;
                        if (i < L && (int32_t)rrow[0][i] == self) atomicMin(s_qpos, (int)i);
                    }
                    wg_barrier();
                    const int found = *s_qpos;      // read into a register BEFORE the second barrier: a fast wave must not start the next
                    wg_barrier();                // chunk's atomicMin while a slow wave has yet to read the flag (the waves would take
                    if (found != 0x7FFFFFFF) break; // different exits and pair their barriers out of order)
                }
            }
            const int qpos = (*s_qpos == 0x7FFFFFFF) ? Li : *s_qpos;
            const int eff_len = (qpos < Li) ? Li - 1 : Li;                          // len(wup) after `del wup[qid_ind]`
            const int alen = (ahp_len > 0) ? (ahp_len < eff_len ? (int)ahp_len : eff_len) : eff_len;   // AHP window (effective ranks)
            // effective rank j -> original position: j if j < qpos else j + 1.  AHP / AP need positions up to:
            int need = kmax;
            if (ahp_len >= 0) need = (alen > need) ? alen : need;
            if (want_ap) need = eff_len;
            const int last_pos = (need < Li) ? need + 1 : Li;                       // original positions [0, last_pos) cover `need` effective ranks

            const int64_t half = ldc / 2;
            // whole-list AHP: its last end point is (all similarities) / best at the last rank that is not the query -- the divisor is
            // requested here, by the thread that will need it in the finish step
            double2 t_last = make_double2(0.0, 0.0);
            if (tid == 0 && ahp_len == 0 && eff_len > 0) {
                const int i_last = (eff_len - 1 < qpos) ? eff_len - 1 : eff_len;
                t_last = rc[hp_slot(i_last) + (i_last < qpos ? half : 0)];
            }
            HP_T(1)
            double car_w[NQ], car_l[NQ];          // running similarity sums up to the current chunk (the same value in every thread)
            int car_r[NQ];                        // relevant items so far
            double acc_w[NQ], acc_l[NQ], acc_ap[NQ];                                // this thread's share of sum(cum / best) and of the AP terms
#pragma unroll
            for (int u = 0; u < NQ; u++) { car_w[u] = car_l[u] = 0.0; car_r[u] = 0; acc_w[u] = acc_l[u] = acc_ap[u] = 0.0; }
            int par = 0;
            const double2 *rct = rc;            // uniform: the chunk's table rows, indexed e * HP_THREADS + tid
            int base = 0;
            // 1 / (best - 1) of a thread's positions in a chunk: contiguous across the wave for every e (1 / best for the ranks ahead of
            // the query: the second half of the class row)
            auto load_curve = [&](auto behind_c, double2 (&dst)[HP_ITEMS], const double2 *rows, int at) {
                constexpr bool BEHIND = decltype(behind_c)::value;       // every position is known to lie behind the query
#pragma unroll
                for (int e = 0; e < HP_ITEMS; e++) dst[e] = rows[e * HP_THREADS + tid + ((!BEHIND && at + e < qpos) ? half : 0)];
            };
            auto step = [&](auto slot_c) {
                constexpr int SLOT = decltype(slot_c)::value;
                const int i0 = base + tid * HP_ITEMS;
                const bool cuts = (nk > 0) && (base <= kmax);               // uniform: a cut-off may fall into this chunk
                // One chunk, in one of three forms (uniform tests below).  FAST: an interior chunk -- every position is live, behind the
                // query, inside the AHP window and away from its end points and from the cut-offs: nothing to mask or test per rank.
                // TAIL: the same except that the list ends inside the chunk (whole-list AHP: its last end point is taken in the finish
                // step): positions are masked, nothing else.  Otherwise the general form.
                auto chunk = [&](auto mode_c) {
                    constexpr int MODE = decltype(mode_c)::value;
                    constexpr bool FAST = MODE == 1, TAIL = MODE == 2;
                    double2 t[HP_ITEMS];              // 1 / (best - 1) of these positions: contiguous across the wave for every e -- ONE load for all NQ queries
                    load_curve(std::integral_constant<bool, FAST || TAIL>{}, t, rct, i0);
                    double2 sv[NQ][HP_ITEMS];         // rank -> class -> similarity pair, summed up inside the thread
                    unsigned rel[NQ];                 // bit e: position e is of the query's class
                    double tw[NQ], tl[NQ];
                    int tr[NQ];
#pragma unroll
                    for (int u = 0; u < NQ; u++) {
                        rel[u] = 0; tw[u] = 0.0; tl[u] = 0.0;
#pragma unroll
                        for (int e = 0; e < HP_ITEMS; e++) {
                            const int i = i0 + e;
                            const bool live = FAST || ((i < last_pos) && (TAIL || i != qpos));
                            int c = 0;
                            if (CLSW == 1) c = s_cls8[r[u][SLOT][e]];
                            else if (CLSW == 2) c = s_cls16[r[u][SLOT][e]];
                            else if (live) c = cls[r[u][SLOT][e]];
                            double2 v = s_sim[c];
                            if (!FAST) { v.x = live ? v.x : 0.0; v.y = live ? v.y : 0.0; }
                            rel[u] |= (live && c == qc) ? (1u << e) : 0u;
                            tw[u] += v.x; tl[u] += v.y;
                            sv[u][e] = make_double2(tw[u], tl[u]);   // inclusive prefix inside the thread: the walk below adds the thread's base to each,
                        }                                            // independent additions instead of a second dependent chain
                        tr[u] = __popc(rel[u]);
                    }
                    // the rank registers are free again: the NEXT chunk's ranks are requested here, in front of the scans and the barrier
                    // (behind the barrier, as in round 5, they had half a chunk to arrive in and the look-ups waited for HBM)
#pragma unroll
                    for (int u = 0; u < NQ; u++) load_ranks(r[u][SLOT], rrow[u], i0 + HP_PF * HP_CHUNK, last_pos);
                    if (FAST) HP_T(2)
                    // ---- workgroup exclusive scan of the thread totals: DPP inside the wave, wave totals through LDS ----
                    double iw[NQ], il[NQ];
                    int ir[NQ];
                    double *part = s_part + par * (2 * HP_WAVES * 3);
#pragma unroll
                    for (int u = 0; u < NQ; u++) {
                        iw[u] = wave_incl_scan_f64(tw[u]); il[u] = wave_incl_scan_f64(tl[u]);
                        ir[u] = wave_incl_scan_i32(tr[u]);
                        if (lane == 63) { part[(u * HP_WAVES + wave) * 3 + 0] = iw[u]; part[(u * HP_WAVES + wave) * 3 + 1] = il[u]; part[(u * HP_WAVES + wave) * 3 + 2] = (double)ir[u]; }
                    }
                    if (FAST) HP_T(3)
                    wg_barrier();   // the only barrier of a chunk (of all NQ queries): the other half of s_part is written next time
                    if (FAST) HP_T(4)
                    double cw[NQ], cl[NQ];
                    int cr[NQ];
#pragma unroll
                    for (int u = 0; u < NQ; u++) {
                        cw[u] = car_w[u] + (iw[u] - tw[u]); cl[u] = car_l[u] + (il[u] - tl[u]);     // sums BEFORE this thread's first element
                        cr[u] = car_r[u] + (ir[u] - tr[u]);
#pragma unroll
                        for (int w = 0; w < HP_WAVES; w++) {
                            const double pw = part[(u * HP_WAVES + w) * 3 + 0], pl = part[(u * HP_WAVES + w) * 3 + 1];
                            const int pr = (int)part[(u * HP_WAVES + w) * 3 + 2];
                            if (w < wave) { cw[u] += pw; cl[u] += pl; cr[u] += pr; }
                            car_w[u] += pw; car_l[u] += pl; car_r[u] += pr;
                        }
                    }
                    if (FAST) HP_T(5)
                    // ---- walk the positions: cumulative sums, cum / best, trapezoid terms, the cut-offs ----
                    int kat0 = 0, knext0 = 0x7FFFFFFF;
                    if (MODE == 0 && cuts && !ks_iota) {   // lower bound of this thread's first effective rank + 1 among the sorted cut-offs
                        const int k0 = (i0 <= qpos) ? i0 + 1 : i0;
                        int hi = nk;
                        while (kat0 < hi) {
                            const int mid = (kat0 + hi) >> 1;
                            if (s_ks[mid] < k0) kat0 = mid + 1; else hi = mid;
                        }
                        if (kat0 < nk) knext0 = s_ks[kat0];
                    }
#pragma unroll
                    for (int u = 0; u < NQ; u++) {
                        int kat = kat0, knext = knext0;
                        double odd_w = 0.0, odd_l = 0.0;       // second accumulator pair of the interior path (shorter FMA chains)
#pragma unroll
                        for (int e = 0; e < HP_ITEMS; e++) {
                            const int i = i0 + e;
                            const double cwe = cw[u] + sv[u][e].x, cle = cl[u] + sv[u][e].y;      // cumulative similarity up to and including this rank
                            if (FAST) {
                                if (e & 1) { odd_w = fma(cwe, t[e].x, odd_w); odd_l = fma(cle, t[e].y, odd_l); }
                                else { acc_w[u] = fma(cwe, t[e].x, acc_w[u]); acc_l[u] = fma(cle, t[e].y, acc_l[u]); }
                            } else if (TAIL) {
                                if (i < last_pos) { acc_w[u] = fma(cwe, t[e].x, acc_w[u]); acc_l[u] = fma(cle, t[e].y, acc_l[u]); }
                            } else if (i < last_pos && i != qpos) {
                                const double yw = cwe * t[e].x, yl = cle * t[e].y;
                                const int j = (i < qpos) ? i : i - 1;                   // effective rank
                                if (ahp_len >= 0 && j < alen) {
                                    acc_w[u] += yw; acc_l[u] += yl;
                                    if (j == 0) { s_ends[4 * u + 0] = yw; s_ends[4 * u + 1] = yl; }
                                    if (ahp_len > 0 && j == alen - 1) { s_ends[4 * u + 2] = yw; s_ends[4 * u + 3] = yl; }   // whole list: taken in the finish step
                                }
                                if (cuts && ks_iota) {
                                    if (j < nk) { orow[u][j] = yw; orow[u][nk + j] = yl; }
                                } else if (cuts && j < kmax) {   // hierarchical precision at k = j + 1, if that is a cut-off: the thread's ranks are
                                    while (knext < j + 1) knext = (++kat < nk) ? s_ks[kat] : 0x7FFFFFFF;   // consecutive, so it merges them with
                                    while (knext == j + 1) {                                            // the sorted cut-offs from its lower bound
                                        orow[u][s_perm[kat]] = yw; orow[u][nk + s_perm[kat]] = yl;
                                        knext = (++kat < nk) ? s_ks[kat] : 0x7FFFFFFF;
                                    }
                                }
                            }
                        }
                        if (FAST) { acc_w[u] += odd_w; acc_l[u] += odd_l; }
                    }
                    if (FAST) HP_T(6)
                    // ---- AP: precision at the relevant ranks (about one in C ranks: a loop over the set bits, not a test per rank) ----
                    if (want_ap) {
#pragma unroll
                        for (int u = 0; u < NQ; u++)
                            for (unsigned m = rel[u]; m; m &= m - 1) {
                                const int e = __ffs(m) - 1, i = i0 + e;
                                const int j1 = (i < qpos) ? i + 1 : i;                  // effective rank + 1
                                acc_ap[u] += (double)(cr[u] + __popc(rel[u] & ((2u << e) - 1u))) * fast_rcp_f64((double)j1);
                            }
                    }
                };
                const bool interior = vec_ok && ahp_len >= 0 && !cuts && base > qpos && base + HP_CHUNK <= last_pos && base + HP_CHUNK - 2 < alen - 1;
                const bool tail = ahp_len == 0 && !cuts && base > qpos;
                if (interior) { chunk(std::integral_constant<int, 1>{}); HP_T(7) }
                else if (tail) { chunk(std::integral_constant<int, 2>{}); HP_T(8) }
                else { chunk(std::integral_constant<int, 0>{}); HP_T(8) }
                base += HP_CHUNK; par ^= 1; rct += HP_CHUNK;
            };
            while (base < last_pos) {
                static_assert(HP_PF >= 1 && HP_PF <= 2, "rank look-ahead: one or two chunks");
                if (base < last_pos) step(std::integral_constant<int, 0>{});
                if (HP_PF > 1 && base < last_pos) step(std::integral_constant<int, HP_PF - 1>{});
            }
            // ---- finish: trapezoid and AP (the end points were left in LDS by whichever thread owned ranks 0 and alen - 1) ----
            {
#pragma unroll
                for (int u = 0; u < NQ; u++) {
                    const double f0 = wave_incl_scan_f64(acc_w[u]), f1 = wave_incl_scan_f64(acc_l[u]), f2 = wave_incl_scan_f64(acc_ap[u]);   // lane 63: the wave's sums
                    if (lane == 63) { s_fin[(u * HP_WAVES + wave) * 3 + 0] = f0; s_fin[(u * HP_WAVES + wave) * 3 + 1] = f1; s_fin[(u * HP_WAVES + wave) * 3 + 2] = f2; }
                }
                wg_barrier();
                if (tid == 0) {
#pragma unroll
                    for (int u = 0; u < NQ; u++) {
                        double g0 = 0.0, g1 = 0.0, g2 = 0.0;
                        for (int w = 0; w < HP_WAVES; w++) { g0 += s_fin[(u * HP_WAVES + w) * 3 + 0]; g1 += s_fin[(u * HP_WAVES + w) * 3 + 1]; g2 += s_fin[(u * HP_WAVES + w) * 3 + 2]; }
                        if (ahp_len == 0 && eff_len > 0) { s_ends[4 * u + 2] = car_w[u] * t_last.x; s_ends[4 * u + 3] = car_l[u] * t_last.y; }   // whole list: y[-1]
                        if (ahp_len >= 0) {
                            // np.trapz(y, dx) = dx * (sum(y) - (y[0] + y[-1]) / 2), dx = 1 / len(wup) (whole list) or 1 / clip
                            const double dx = 1.0 / (double)((ahp_len > 0) ? ahp_len : eff_len);
                            orow[u][2 * nk] = dx * (g0 - 0.5 * (s_ends[4 * u + 0] + s_ends[4 * u + 2]));
                            orow[u][2 * nk + 1] = dx * (g1 - 0.5 * (s_ends[4 * u + 1] + s_ends[4 * u + 3]));
                        }
                        if (want_ap) orow[u][2 * nk + 2] = car_r[u] > 0 ? g2 / (double)car_r[u] : 0.0;
                    }
                }
            }
            HP_T(9)
        };
        // flags of the order kernel: 1 = the query is the first entry of its own ranking, 2 = it is not in the gallery, 0 = look for it
        const bool pair = SE_HP_PAIR && order_ws && nq_drawn == 2 && qcv[0] == qcv[1] && flagv[0] == flagv[1] && flagv[0] != 0;
        if (pair) {
            run(std::integral_constant<int, 2>{}, qv, selfv, qcv[0], flagv[0] == 1 ? 0 : 0x7FFFFFFF);
        } else {
            for (int u = 0; u < nq_drawn; u++) {
                const int64_t q1[2] = {qv[u], 0};
                const int32_t s1[2] = {selfv[u], -1};
                run(std::integral_constant<int, 1>{}, q1, s1, qcv[u], (order_ws && flagv[u] != 0) ? (flagv[u] == 1 ? 0 : 0x7FFFFFFF) : -1);
            }
        }
    }
#if SE_HP_PROFILE
    if (tid == 0)
        for (int i = 0; i < 16; i++) atomicAdd(&hp_prof[i], hp_acc[i]);
#endif
}

}  // namespace se

using namespace se;

extern "C" int64_t se_hprec_curve_len(int64_t list_len)
{
    return list_len <= 0 ? 0 : (list_len + HP_CHUNK - 1) / HP_CHUNK * HP_CHUNK;
}

extern "C" int se_hprec_reciprocal_curves(const double *best_wup, const double *best_lcs, int64_t ldb, int num_classes, int64_t list_len,
                                          double *rcp, se_stream_t stream)
{
    if (num_classes <= 0 || list_len <= 0 || ldb < list_len) return fail(SE_ERR_INVALID, "se_hprec_reciprocal_curves: bad shape classes=%d len=%lld ldb=%lld", num_classes, (long long)list_len, (long long)ldb);
    if (!best_wup || !best_lcs || !rcp) return fail(SE_ERR_INVALID, "se_hprec_reciprocal_curves: null pointer");
    const int64_t lp = se_hprec_curve_len(list_len);
    int64_t gx = lp / HP_THREADS;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(hprec_rcp_kernel, dim3((unsigned)gx, (unsigned)num_classes), dim3(HP_THREADS), 0, (hipStream_t)stream, best_wup, best_lcs, ldb,
                       list_len, lp, reinterpret_cast<double2 *>(rcp));
    SE_LAUNCH_CHECK();
    return SE_OK;
}

extern "C" int64_t se_hprec_order_workspace_bytes(int64_t q) { return q < 0 ? 0 : (int64_t)sizeof(int32_t) * (HP_WS_HEAD + 4 * q); }

namespace se {

struct HpLaunch { int clsw; size_t lds; int grid; };

template <int CLSW, typename RT>
static hipError_t hp_occupancy(size_t lds, int *blocks_per_cu)
{
    hipError_t e = hipFuncSetAttribute((const void *)hprec_kernel<CLSW, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, hprec_kernel<CLSW, RT>, HP_THREADS, lds);
}

}  // namespace se

template <typename RT>
static int hp_run(const RT *rank, int64_t ldr, int64_t q, int64_t list_len, const int32_t *cls, int64_t gallery,
                  const int32_t *qcls, const int32_t *qidx, const double *wup, const double *lcs,
                  int num_classes, const double *rcp, int64_t rcp_len, const int32_t *ks, int nk, int64_t ahp_len, int want_ap,
                  double *out, int64_t ldo, void *order_ws, se_stream_t stream)
{
    if (sizeof(RT) == 2 && gallery > 65536) return fail(SE_ERR_INVALID, "se_hierarchical_precision_r16: a gallery of %lld items does not fit 16-bit ranks", (long long)gallery);
    if (q < 0 || list_len <= 0 || gallery <= 0 || num_classes <= 0 || nk < 0 || nk > HP_MAX_KS || q > 0x7FFFFFFF || list_len > 0x7FFFFFFF - 2 * HP_CHUNK)
        return fail(SE_ERR_INVALID, "se_hierarchical_precision: bad shape q=%lld len=%lld gallery=%lld classes=%d nk=%d", (long long)q,
                    (long long)list_len, (long long)gallery, num_classes, nk);
    if (q == 0) return SE_OK;
    if (!rank || !cls || !qcls || !wup || !lcs || !rcp || !out || (nk > 0 && !ks))
        return fail(SE_ERR_INVALID, "se_hierarchical_precision: null pointer");
    if (ldr < list_len || ldo < 2 * nk + 3) return fail(SE_ERR_INVALID, "se_hierarchical_precision: leading dimension too small");
    if (rcp_len < list_len) return fail(SE_ERR_INVALID, "se_hierarchical_precision: the reciprocal curves cover %lld positions, the rankings have %lld", (long long)rcp_len, (long long)list_len);
    const size_t fixed = (size_t)num_classes * sizeof(double2) + (size_t)(2 * 2 * HP_WAVES * 3 + 2 * HP_WAVES * 3 + 8) * sizeof(double) + (size_t)(2 * nk + 12) * sizeof(int);
    const size_t cap = 160 * 1024;
    if (fixed > cap) return fail(SE_ERR_UNSUPPORTED, "se_hierarchical_precision: %d classes exceed the LDS similarity rows", num_classes);
    // the gallery's classes as bytes / shorts in LDS when they fit (two workgroups per CU preferred for the byte table), else gathered
    const size_t tab8 = ((size_t)gallery + 15) / 16 * 16, tab16 = ((size_t)gallery * 2 + 15) / 16 * 16;
    int clsw = 0;
    if (list_len * (q < 4096 ? q : 4096) < 8 * gallery) clsw = 0;      // short lists / few queries: filling the table would cost more than the gathers
    else if (num_classes <= 256 && fixed + tab8 <= cap) clsw = 1;
    else if (num_classes <= 65536 && fixed + tab16 <= cap) clsw = 2;
    const size_t lds = fixed + (clsw == 1 ? tab8 : clsw == 2 ? tab16 : 0);
    hipStream_t s = (hipStream_t)stream;
    int per_cu = 0, dev = 0, cus = 0;
    SE_HIP_CHECK(clsw == 1 ? (hp_occupancy<1, RT>(lds, &per_cu)) : clsw == 2 ? (hp_occupancy<2, RT>(lds, &per_cu)) : (hp_occupancy<0, RT>(lds, &per_cu)));
    SE_HIP_CHECK(hipGetDevice(&dev));
    SE_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (per_cu < 1) per_cu = 1;
    int32_t *ws = static_cast<int32_t *>(order_ws);
    const size_t order_lds = (size_t)(num_classes + HP_ORDER_THREADS / WAVE) * sizeof(int);
    if (ws) {
        SE_HIP_CHECK(hipFuncSetAttribute((const void *)hprec_order_kernel<RT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)order_lds));
        if (reinterpret_cast<uintptr_t>(ws) % 16 != 0) return fail(SE_ERR_INVALID, "se_hierarchical_precision: order_ws must be 16-byte aligned");
        hipLaunchKernelGGL(hprec_order_kernel<RT>, dim3(1), dim3(HP_ORDER_THREADS), order_lds, s, qcls, qidx, q, num_classes, rank, ldr, ws);
        SE_LAUNCH_CHECK();
    }
    // persistent workgroups (the class table is loaded once each): as many as are resident at once
    const int64_t resident = (int64_t)per_cu * cus;
    const int64_t grid = q < resident ? q : resident;
    const int vec_ok = ((ldr * (int64_t)sizeof(RT)) % 16 == 0) && (reinterpret_cast<uintptr_t>(rank) % 16 == 0);
    const double2 *rc = reinterpret_cast<const double2 *>(rcp);
    const int64_t ldc = 2 * se_hprec_curve_len(rcp_len);
#define SE_HP_LAUNCH(W)                                                                                                                  \
    hipLaunchKernelGGL((hprec_kernel<W, RT>), dim3((unsigned)grid), dim3(HP_THREADS), lds, s, rank, ldr, q, list_len, cls, gallery, qcls, qidx, \
                       wup, lcs, num_classes, rc, ldc, ks, nk, ahp_len, want_ap, out, ldo, ws, vec_ok)
    if (clsw == 1) SE_HP_LAUNCH(1);
    else if (clsw == 2) SE_HP_LAUNCH(2);
    else SE_HP_LAUNCH(0);
#undef SE_HP_LAUNCH
    SE_LAUNCH_CHECK();
#if SE_HP_PROFILE
    {
        unsigned long long h[16], z[16] = {};
        SE_HIP_CHECK(hipStreamSynchronize(s));
        SE_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(hp_prof), sizeof(h)));
        SE_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(hp_prof), z, sizeof(z)));
        static const char *names[10] = {"setup", "query-prologue", "lookups", "scans", "barrier", "parts", "walk", "ap+rest(fast)", "slow-chunk", "finish"};
        double tot = 0;
        for (int i = 0; i < 10; i++) tot += (double)h[i];
        fprintf(stderr, "[se_hierarchical_precision profile] grid=%lld q=%lld len=%lld: cycles per workgroup %.0f;", (long long)grid, (long long)q, (long long)list_len, tot / (double)grid);
        for (int i = 0; i < 10; i++) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * (double)h[i] / tot);
        fprintf(stderr, "\n");
    }
#endif
    return SE_OK;
}

extern "C" int se_hierarchical_precision(const int32_t *rank, int64_t ldr, int64_t q, int64_t list_len, const int32_t *cls, int64_t gallery,
                                         const int32_t *qcls, const int32_t *qidx, const double *wup, const double *lcs,
                                         int num_classes, const double *rcp, int64_t rcp_len, const int32_t *ks, int nk, int64_t ahp_len, int want_ap,
                                         double *out, int64_t ldo, void *order_ws, se_stream_t stream)
{
    return hp_run<int32_t>(rank, ldr, q, list_len, cls, gallery, qcls, qidx, wup, lcs, num_classes, rcp, rcp_len, ks, nk, ahp_len, want_ap, out, ldo, order_ws, stream);
}

extern "C" int se_hierarchical_precision_r16(const uint16_t *rank, int64_t ldr, int64_t q, int64_t list_len, const int32_t *cls, int64_t gallery,
                                             const int32_t *qcls, const int32_t *qidx, const double *wup, const double *lcs,
                                             int num_classes, const double *rcp, int64_t rcp_len, const int32_t *ks, int nk, int64_t ahp_len, int want_ap,
                                             double *out, int64_t ldo, void *order_ws, se_stream_t stream)
{
    return hp_run<uint16_t>(rank, ldr, q, list_len, cls, gallery, qcls, qidx, wup, lcs, num_classes, rcp, rcp_len, ks, nk, ahp_len, want_ap, out, ldo, order_ws, stream);
}
