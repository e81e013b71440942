// hprec.hip -- hierarchical retrieval metrics on the device (the consumer of the rankings).
//
// Replaces the per-query loop of `ClassHierarchy.hierarchical_precision` (class_hierarchy.py:211-316): for every
// query, class similarities gathered along its ranking (Wu-Palmer and 1 - LCS height), hierarchical precision at
// the cut-offs `ks`, the area under that curve (AHP, whole list or clipped) and average precision.  The class x
// class similarity tables and the best-possible cumulative similarity per query class (class_hierarchy.py:262-276:
// the descending-sorted similarities of the whole gallery, cumulated) are computed once on the host and passed in;
// the 0.6 h the reference spends here at N = 50k is the gather + prefix sums over Q x N ranks, which is this kernel.
//
// One 256-thread workgroup per query walks its ranking in blocked chunks (thread t owns 8 consecutive ranks):
// gather class -> similarity row (LDS), float64 running sums by a workgroup scan with a carry between chunks.
// Bookkeeping kept from the reference: the query itself is dropped from its ranking (position q_pos), which shifts
// the best curve left there and subtracts its self-similarity 1.0 (class_hierarchy.py:280-290); AHP is numpy's
// trapz of cum / best with dx = 1 / length; AP is the mean over the relevant ranks of precision at that rank.
// float64 throughout; sums are associated differently from numpy's sequential cumsum (last-bits differences).
#include "se_common.h"

namespace se {

constexpr int HP_THREADS = 256;
constexpr int HP_WAVES = HP_THREADS / WAVE;
constexpr int HP_ITEMS = 8;
constexpr int HP_CHUNK = HP_THREADS * HP_ITEMS;
constexpr int HP_MAX_KS = 512;

struct HpCarry {
    double cw, cl;       // running similarity sums (effective sequence, query skipped)
    double yw, yl;       // running sum of cum / best (for the trapezoid)
    double ap;           // running sum of precision at relevant ranks
    long long rel;       // relevant items so far
};

__device__ __forceinline__ double wave_incl_scan_f64(double v)
{
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

__device__ __forceinline__ long long wave_incl_scan_i64(long long v)
{
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const long long t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// out row layout: [P@k WUP x nk][P@k LCS x nk][AHP WUP][AHP LCS][AP]
__global__ __launch_bounds__(HP_THREADS) void hprec_kernel(const int32_t *__restrict__ rank, int64_t ldr, int64_t Q, int64_t L,
                                                           const int32_t *__restrict__ cls, const int32_t *__restrict__ qcls,
                                                           const int32_t *__restrict__ qidx, const double *__restrict__ wup,
                                                           const double *__restrict__ lcs, int C, const double *__restrict__ best_wup,
                                                           const double *__restrict__ best_lcs, int64_t ldb,
                                                           const int32_t *__restrict__ ks, int nk, int64_t ahp_len, int want_ap,
                                                           double *__restrict__ out, int64_t ldo)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char hp_raw[];
    double *s_wup = reinterpret_cast<double *>(hp_raw);   // [C] similarity row of the query class
    double *s_lcs = s_wup + C;                            // [C]
    double *s_part = s_lcs + C;                           // [HP_WAVES][6] wave totals
    int *s_qpos = reinterpret_cast<int *>(s_part + HP_WAVES * 6);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int64_t q = blockIdx.x; q < Q; q += gridDim.x) {
        const int32_t *rrow = rank + q * ldr;
        const int qc = qcls[q];
        const int32_t self = qidx ? qidx[q] : -1;
        const double *bw = best_wup + (int64_t)qc * ldb, *bl = best_lcs + (int64_t)qc * ldb;
        double *orow = out + q * ldo;
        __syncthreads();
        for (int c = tid; c < C; c += HP_THREADS) { s_wup[c] = wup[(int64_t)qc * C + c]; s_lcs[c] = lcs[(int64_t)qc * C + c]; }
        if (tid == 0) *s_qpos = 0x7FFFFFFF;
        __syncthreads();
        // ---- position of the query in its own ranking (first hit; L if absent) ----
        if (self >= 0) {   // chunk by chunk with a uniform early exit: the query is normally its own nearest neighbour
            for (int64_t b0 = 0; b0 < L; b0 += 4 * HP_THREADS) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int64_t i = b0 + e * HP_THREADS + tid;
                    if (i < L && rrow[i] == self) atomicMin(s_qpos, (int)i);
                }
                __syncthreads();
                const int found = *s_qpos;      // read into a register BEFORE the second barrier: a fast wave must not start the next
                __syncthreads();                // chunk's atomicMin while a slow wave has yet to read the flag (the waves would take
                if (found != 0x7FFFFFFF) break; // different exits and pair their barriers out of order)
            }
        }
        __syncthreads();
        const int64_t qpos = (*s_qpos == 0x7FFFFFFF) ? L : (int64_t)*s_qpos;
        const int64_t eff_len = (qpos < L) ? L - 1 : L;                         // len(wup) after `del wup[qid_ind]`
        const int64_t alen = (ahp_len > 0) ? (ahp_len < eff_len ? ahp_len : eff_len) : eff_len;   // AHP window (effective ranks)
        // effective rank j -> original position: j if j < qpos else j + 1.  AHP / AP need positions up to:
        int64_t need = 0;
        for (int i = 0; i < nk; i++) need = (ks[i] > need) ? ks[i] : need;
        const int64_t kmax = need;
        if (ahp_len >= 0) need = (alen > need) ? alen : need;
        if (want_ap) need = eff_len;
        int64_t last_pos = need + 1;                                            // original positions [0, last_pos) cover `need` effective ranks
        if (last_pos > L) last_pos = L;

        HpCarry carry = {0.0, 0.0, 0.0, 0.0, 0.0, 0};
        double y_first_w = 0.0, y_first_l = 0.0, y_last_w = 0.0, y_last_l = 0.0;   // trapezoid end points (valid on the owning thread)
        for (int64_t base = 0; base < last_pos; base += HP_CHUNK) {
            // ---- this thread's 8 consecutive positions ----
            double vw[HP_ITEMS], vl[HP_ITEMS];
            int rl[HP_ITEMS];
            double tw = 0.0, tl = 0.0;
            long long tr = 0;
#pragma unroll
            for (int e = 0; e < HP_ITEMS; e++) {
                const int64_t i = base + (int64_t)tid * HP_ITEMS + e;
                const bool live = (i < last_pos) && (i != qpos);
                int c = 0;
                if (live) c = cls[rrow[i]];
                vw[e] = live ? s_wup[c] : 0.0;
                vl[e] = live ? s_lcs[c] : 0.0;
                rl[e] = (live && c == qc) ? 1 : 0;
                tw += vw[e]; tl += vl[e]; tr += rl[e];
            }
            // ---- workgroup exclusive scan of the thread totals ----
            const double iw = wave_incl_scan_f64(tw), il = wave_incl_scan_f64(tl);
            const long long ir = wave_incl_scan_i64(tr);
            if (lane == 63) { s_part[wave * 6 + 0] = iw; s_part[wave * 6 + 1] = il; s_part[wave * 6 + 2] = (double)ir; }
            __syncthreads();
            double ow = carry.cw, ol = carry.cl;
            long long orl = carry.rel;
            for (int w = 0; w < wave; w++) { ow += s_part[w * 6 + 0]; ol += s_part[w * 6 + 1]; orl += (long long)s_part[w * 6 + 2]; }
            double cw = ow + (iw - tw), cl = ol + (il - tl);     // sums BEFORE this thread's first element
            long long cr = orl + (ir - tr);
            // ---- walk the 8 positions: cumulative sums, P@k, trapezoid terms, AP terms ----
            double syw = 0.0, syl = 0.0, sap = 0.0;
#pragma unroll
            for (int e = 0; e < HP_ITEMS; e++) {
                const int64_t i = base + (int64_t)tid * HP_ITEMS + e;
                cw += vw[e]; cl += vl[e]; cr += rl[e];
                if (i < last_pos && i != qpos) {
                    const int64_t j = (i < qpos) ? i : i - 1;                  // effective rank
                    const double sub = (i > qpos) ? 1.0 : 0.0;                 // best curve shifted left at the query, minus its self-similarity
                    const double yw = cw / (bw[i] - sub), yl = cl / (bl[i] - sub);
                    if (j < kmax)
                        for (int t = 0; t < nk; t++)
                            if ((int64_t)ks[t] - 1 == j) { orow[t] = yw; orow[nk + t] = yl; }
                    if (ahp_len >= 0 && j < alen) {
                        syw += yw; syl += yl;
                        if (j == 0) { y_first_w = yw; y_first_l = yl; }
                        if (j == alen - 1) { y_last_w = yw; y_last_l = yl; }
                    }
                    if (want_ap && rl[e]) sap += (double)cr / (double)(j + 1);
                }
            }
            syw = wave_sum_f64(syw); syl = wave_sum_f64(syl); sap = wave_sum_f64(sap);
            __syncthreads();   // everyone has read the wave totals of the scan
            if (lane == 0) { s_part[wave * 6 + 3] = syw; s_part[wave * 6 + 4] = syl; s_part[wave * 6 + 5] = sap; }
            __syncthreads();
            for (int w = 0; w < HP_WAVES; w++) {
                carry.cw += s_part[w * 6 + 0]; carry.cl += s_part[w * 6 + 1]; carry.rel += (long long)s_part[w * 6 + 2];
                carry.yw += s_part[w * 6 + 3]; carry.yl += s_part[w * 6 + 4]; carry.ap += s_part[w * 6 + 5];
            }
            __syncthreads();
        }
        // ---- finish: trapezoid and AP (end points live on whichever thread owned ranks 0 and alen - 1) ----
        {
            double e0 = wave_sum_f64(y_first_w) , e1 = wave_sum_f64(y_first_l), e2 = wave_sum_f64(y_last_w), e3 = wave_sum_f64(y_last_l);
            if (lane == 0) { s_part[wave * 6 + 0] = e0; s_part[wave * 6 + 1] = e1; s_part[wave * 6 + 2] = e2; s_part[wave * 6 + 3] = e3; }
            __syncthreads();
            if (tid == 0) {
                double f_w = 0, f_l = 0, l_w = 0, l_l = 0;
                for (int w = 0; w < HP_WAVES; w++) { f_w += s_part[w * 6 + 0]; f_l += s_part[w * 6 + 1]; l_w += s_part[w * 6 + 2]; l_l += s_part[w * 6 + 3]; }
                if (ahp_len >= 0) {
                    // np.trapz(y, dx) = dx * (sum(y) - (y[0] + y[-1]) / 2), dx = 1 / len(wup) (whole list) or 1 / clip
                    const double dx = 1.0 / (double)((ahp_len > 0) ? ahp_len : eff_len);
                    orow[2 * nk] = dx * (carry.yw - 0.5 * (f_w + l_w));
                    orow[2 * nk + 1] = dx * (carry.yl - 0.5 * (f_l + l_l));
                }
                if (want_ap) orow[2 * nk + 2] = carry.rel > 0 ? carry.ap / (double)carry.rel : 0.0;
            }
        }
    }
}

}  // namespace se

using namespace se;

extern "C" int se_hierarchical_precision(const int32_t *rank, int64_t ldr, int64_t q, int64_t list_len, const int32_t *cls,
                                         const int32_t *qcls, const int32_t *qidx, const double *wup, const double *lcs,
                                         int num_classes, const double *best_wup, const double *best_lcs, int64_t ldb,
                                         const int32_t *ks, int nk, int64_t ahp_len, int want_ap, double *out, int64_t ldo,
                                         se_stream_t stream)
{
    if (q < 0 || list_len <= 0 || num_classes <= 0 || nk < 0 || nk > HP_MAX_KS)
        return fail(SE_ERR_INVALID, "se_hierarchical_precision: bad shape q=%lld len=%lld classes=%d nk=%d", (long long)q, (long long)list_len, num_classes, nk);
    if (q == 0) return SE_OK;
    if (!rank || !cls || !qcls || !wup || !lcs || !best_wup || !best_lcs || !out || (nk > 0 && !ks))
        return fail(SE_ERR_INVALID, "se_hierarchical_precision: null pointer");
    if (ldr < list_len || ldb < list_len || ldo < 2 * nk + 3) return fail(SE_ERR_INVALID, "se_hierarchical_precision: leading dimension too small");
    const size_t lds = (size_t)(2 * num_classes + HP_WAVES * 6) * sizeof(double) + 16;
    if (lds > 160 * 1024) return fail(SE_ERR_UNSUPPORTED, "se_hierarchical_precision: %d classes exceed the LDS similarity rows", num_classes);
    hipStream_t s = (hipStream_t)stream;
    SE_HIP_CHECK(hipFuncSetAttribute((const void *)hprec_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t grid = q < 2048 ? q : 2048;
    hipLaunchKernelGGL(hprec_kernel, dim3((unsigned)grid), dim3(HP_THREADS), lds, s, rank, ldr, q, list_len, cls, qcls, qidx, wup, lcs,
                       num_classes, best_wup, best_lcs, ldb, ks, nk, ahp_len, want_ap, out, ldo);
    SE_LAUNCH_CHECK();
    return SE_OK;
}
