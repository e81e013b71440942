// prefilter.hip -- half-precision matrix-core PRE-FILTER of the fused distance + top-k (se_retrieve_topk; SURVEY.md section 7 hard
// part 2, section 8d "fused top-k").
//
// Replaces nothing of the reference by itself: it decides WHICH of the Q x N distances of evaluate_retrieval.py:57-63 are worth
// computing exactly.  Every distance that reaches a caller still comes from the canonical fp32 FMA chain (topk.hip recomputes each
// surviving candidate with it), so the output stays bit-identical -- the filter only has to BOUND distances.
//
//   d~(q, g)  = distance computed from fp16 images of the operands on v_mfma_f32_32x32x16_f16 (2.5 PFLOP/s dense, 16x the fp32 pipe)
//   |d~ - d| <= eps(q)        rigorous, from the operands' actual rounding residuals (pf_convert_kernel) -- DESIGN.md section 5.3
//
// Why fp16 and not bf16: the filter's window is 2 eps wide and every item inside it costs one exact recomputation = one 4 D-byte
// gallery row out of HBM.  fp16 keeps 11 significant bits against bf16's 8 (eps 8x smaller); its narrow exponent range is dealt
// with by ONE power-of-two scale per operand matrix (largest regular magnitude -> [2^13, 2^14): no overflow; exact to undo) and by
// flushing scaled values below 2^-14 to zero in the image ourselves, so the matrix core never sees a denormal input and the
// residual norm accounts for the flush exactly.  (First version, bf16: 554 exact recomputations per query at D = 1000, k = 251.)
//
// Kernels:
//   pf_maxabs_kernel    largest regular magnitude of a matrix -> its scale exponent.
//   pf_convert_kernel   fp32 rows -> fp16 rows (x 2^e, padded with zeros to a multiple of 128 columns) + per row: the norm of the image
//                       (unscaled units) and the norm of the residual x - image (both rounded up) + the maxima over all rows.  Rows that
//                       are not "regular" (a non-finite entry, or a magnitude >= 2^60) get an all-NaN image: every d~ with such a row is
//                       NaN, NaN always passes the filter, the exact path decides about them.
//   pf_tile_kernel      the tile loop: persistent 256-thread workgroups (2 x 2 waves, 64 x 64 outputs per wave = 2 x 2 MFMA blocks),
//                       128 x 128 tiles, K-chunks of 128 halves (256 bytes per row) staged through LDS with a software-pipelined
//                       global -> register prefetch (16 x 16 bytes per thread in flight: at 16x the fp32 MFMA rate a chunk's matrix
//                       work no longer covers a global round trip, the chunk has to be long); rows = gallery, columns = queries.
//                       Epilogues:
//                         PF_GROUPMIN  sample pass: minimum of each lane's 16 values per block -> gm[query, group]
//                         PF_FILTER    values <= thr[query] (or NaN) appended to the query's candidate list as (d~ bits, gallery row);
//                                      all-pairs calls walk the upper triangle and filter every off-diagonal tile in BOTH orientations --
//                                      the second one (queries = tile rows) straight from the accumulators with wave ballots: no LDS
//                                      transposition, no barrier
//                         PF_STORE     (tuning build only) d~ matrix out, for the hardware-assumption test of the error bound
//   pf_big_kernel       the filter pass for long rows (padded width >= 256): 256 x 256 tiles, four waves of 128 x 128 outputs with their
//                       accumulators in AGPRs, operands by LDS-DMA one K-chunk ahead -- see its own header further down
#include "se_common.h"
#include <type_traits>

namespace se {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float pf_f32x16 __attribute__((ext_vector_type(16)));

#ifndef SE_PF_BM
#define SE_PF_BM 128      // 256: 512-thread workgroups, one per CU (measured: 28.8 vs 26.6 ms on the D = 1000 shard -- with one workgroup per CU
#endif                    // every wave is in the same phase at the same time and nothing overlaps the load bursts)
#ifndef SE_PF_BK
#define SE_PF_BK 128
#endif
constexpr int PF_BM = SE_PF_BM, PF_BN = 128, PF_BK = SE_PF_BK;   // tile (gallery rows x queries), K-chunk (fp16 elements)
constexpr int PF_THREADS = PF_BM * 2;                       // (PF_BM / 64) x 2 waves of 64 x 64 outputs
constexpr int PF_ROWB = PF_BK * 2;                          // bytes of one operand row of a chunk (256)
constexpr int PF_PITCH = PF_ROWB + 16;                      // LDS row pitch in bytes: 68 dwords -> conflict-free ds_read_b128 over 16 rows
constexpr int PF_PPR = PF_ROWB / 16;                        // 16-byte pieces per row (16)
constexpr int PF_NLOAD_A = PF_BM * PF_PPR / PF_THREADS;     // pieces per thread: gallery operand (8)
constexpr int PF_NLOAD_B = PF_BN * PF_PPR / PF_THREADS;     // query operand (4)
// The filter epilogue's accumulator dump (36 dwords per thread) lives in the gallery operand's LDS plus this gap in front of the query
// operand: the query panel of a one-chunk job (padded width 128) stays in LDS for all tiles of the job and must not be overwritten.
constexpr int PF_DUMP_BYTES = SE_PF_BM * 2 * 36 * 4;
constexpr int PF_GAP = PF_DUMP_BYTES > PF_BM * PF_PITCH ? PF_DUMP_BYTES - PF_BM * PF_PITCH : 0;
#ifndef SE_PF_CB
#define SE_PF_CB 3        // candidates per lane, query and tile taken in straight-line code by the filter epilogue (0: the loop only)
#endif
#ifndef SE_PF_WGS
#define SE_PF_WGS (256 / SE_PF_BM)
#endif
constexpr int PF_WGS_PER_CU = SE_PF_WGS;                    // default: 8 waves per CU, <= 256 registers each

constexpr int PF_GROUPMIN = PF_EPI_GROUPMIN, PF_FILTER = PF_EPI_FILTER, PF_STORE = PF_EPI_STORE;

struct PfArgs {
    float *gm; int64_t gm_ld;          // PF_GROUPMIN
    const float *thr;                  // PF_FILTER: [queries] thresholds (distance units); NaN = nothing but NaN passes
    unsigned *rowcnt; uint2 *lists; int64_t cap;
    int spill; uint2 *spill_lists; unsigned *spill_cnt;   // PF_FILTER: sub-list slots of the query's SHARED spill region [queries][spill * cap], its fill counters [queries] (see pf_spill_append)
    int64_t sqa_stride;                // Euclidean epilogue: |a|^2 of gallery row r is sqa[r * sqa_stride]
    float *out; int64_t ldo;           // PF_STORE
    unsigned long long *prof;          // tuning build: phase cycle counters (SE_PF_PROFILE=1)
    int last_steps;                    // k = 16 MFMA steps of a tile's LAST chunk that hold any data (the rest of the padded width is zero: D = 100 -> 7 of 8)
};

// ---- conversion ---------------------------------------------------------------------------------------------------------------
// ctl words of one operand matrix (uint32; float bits are combined with atomicMax: all values are >= 0):
//   [0] max row norm of the image  [1] max residual norm  [2] number of irregular rows  [3] largest regular magnitude  [4] scale exponent e
constexpr float PF_REG_LIMIT = 1.152921504606846976e18f;     // 2^60: magnitudes from here on make a row irregular

__global__ __launch_bounds__(256) void pf_maxabs_kernel(const float *__restrict__ x, int64_t ldx, int64_t n, int d, unsigned *__restrict__ ctl)
{
    __shared__ float wmax[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
        const float *xr = x + r * ldx;
        for (int c = lane; c < d; c += 64) {
            const float a = __builtin_fabsf(xr[c]);
            m = (a < PF_REG_LIMIT && a > m) ? a : m;               // NaN / inf / huge entries do not set the scale
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
    if (lane == 0) wmax[wave] = m;
    wg_barrier();
    if (threadIdx.x == 0) {
        m = wmax[0];
        for (int i = 1; i < 4; i++) m = wmax[i] > m ? wmax[i] : m;
        atomicMax(&ctl[3], __float_as_uint(m));
    }
}

// scale exponent: largest regular magnitude m = f 2^ex (f in [0.5, 1)) -> m 2^e in [2^13, 2^14)
__device__ __forceinline__ int pf_scale_exp(float m)
{
    if (!(m > 0.f)) return 0;
    int ex;
    (void)frexpf(m, &ex);
    int e = 14 - ex;
    // |e| <= 60: the filter passes undo the scales of BOTH operand images with ldexpf(1, +-(ea + eb)), which must stay finite (a matrix whose
    // largest entry is below 2^-46 gets a coarser image -- more of it flushes to zero, its residual norms and with them eps grow: slower, not wrong);
    // the lower end cannot bind: regular magnitudes are < PF_REG_LIMIT = 2^60
    return e < -60 ? -60 : (e > 60 ? 60 : e);
}

__global__ __launch_bounds__(256) void pf_convert_kernel(const float *__restrict__ x, int64_t ldx, int64_t n, int d, int kp,
                                                         uint16_t *__restrict__ out, float *__restrict__ nrm, float *__restrict__ res,
                                                         unsigned *__restrict__ ctl)
{
    __shared__ float wm_n[4], wm_r[4];
    __shared__ unsigned wm_b[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = pf_scale_exp(__uint_as_float(ctl[3]));
    const float sc = ldexpf(1.0f, e), isc = ldexpf(1.0f, -e);
    float wmax_n = 0.f, wmax_r = 0.f;
    unsigned wbad = 0;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
        const float *xr = x + r * ldx;
        uint16_t *orow = out + r * (int64_t)kp;
        float sn = 0.f, sr = 0.f;
        bool bad = false;
        for (int c0 = lane * 4; c0 < kp; c0 += 256) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = (c0 + i < d) ? xr[c0 + i] : 0.f;
            uint16_t h[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float xs = v[i] * sc;
                _Float16 hh = (_Float16)xs;                                       // round to nearest even
                if (__builtin_fabsf(xs) < 6.103515625e-05f) hh = (_Float16)0.f;   // below fp16's normal range: flushed HERE, never a denormal input
                // image and residual in SCALED units (magnitudes up to 2^14): the squares of entries far below the matrix' largest one would
                // underflow in the operand's own units and drop out of the norms -- and eps must be a true bound
                const float hv = (float)hh;
                const float rv = xs - hv;                                         // exact (xs = v 2^e, hv its fp16 rounding or 0)
                sn = __builtin_fmaf(hv, hv, sn);
                sr = __builtin_fmaf(rv, rv, sr);
                bad = bad || !(__builtin_fabsf(v[i]) < PF_REG_LIMIT);             // NaN, inf or |v| >= 2^60
                h[i] = __builtin_bit_cast(uint16_t, hh);
            }
            *(uint2 *)(orow + c0) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
        }
        sn = wave_sum(sn);
        sr = wave_sum(sr);
        const bool any_bad = __ballot(bad) != 0ull;
        // upper bounds of the two norms: the fp32 sums above carry a relative error < (d / 64 + 8) 2^-24 < 2^-9 for d <= 2^20
        // + 2^-52 (scaled units): squares below 2^-126 -- entries below 2^-63 -- are lost to the sums: d 2^-126 <= 2^-106 in all for d <= 2^20.
        // Back in the operand's units (exact power-of-two scale, floored at the smallest normal number).
        float nn = (sqrtf(sn) * 1.002f + 2.220446e-16f) * isc + 1.1754944e-38f, rr = (sqrtf(sr) * 1.002f + 2.220446e-16f) * isc + 1.1754944e-38f;
        if (any_bad) {
            nn = rr = __builtin_nanf("");
            for (int c0 = lane * 4; c0 < kp; c0 += 256) *(uint2 *)(orow + c0) = make_uint2(0x7E007E00u, 0x7E007E00u);   // all-NaN image (fp16 quiet NaN)
            wbad += (lane == 0);
        } else {
            wmax_n = nn > wmax_n ? nn : wmax_n;
            wmax_r = rr > wmax_r ? rr : wmax_r;
        }
        if (lane == 0) { nrm[r] = nn; res[r] = rr; }
    }
    // one set of atomics per workgroup (per wave they serialised on three addresses: 0.75 ms for 50,000 short rows)
    if (lane == 0) { wm_n[wave] = wmax_n; wm_r[wave] = wmax_r; wm_b[wave] = wbad; }
    wg_barrier();
    if (threadIdx.x == 0) {
        float a = wm_n[0], b = wm_r[0];
        unsigned c = wm_b[0];
        for (int i = 1; i < 4; i++) { a = wm_n[i] > a ? wm_n[i] : a; b = wm_r[i] > b ? wm_r[i] : b; c += wm_b[i]; }
        atomicMax(&ctl[0], __float_as_uint(a));
        atomicMax(&ctl[1], __float_as_uint(b));
        if (c) atomicAdd(&ctl[2], c);
        if (blockIdx.x == 0) ctl[4] = (unsigned)e;
    }
}

// A query's candidates of (gallery range p, tile sequence j) go to sub-list p gj + j, which ONE workgroup fills through a slot counter in
// LDS.  A gallery sorted by class puts a query's neighbours into a few adjacent tiles, i.e. into two to four of the sub-lists, which
// then overflow while the others stay empty (round 4: every query of a class-sorted ILSVRC-sized shard went to the exact fallback, 127x
// the time).  So every query also has a spill region of `spill` more sub-list slots that all its sub-lists share (a separate array: the
// sub-lists keep their compact pitch -- interleaved with them the spill slots cost the filter pass 12 % at 50k x 50k): an entry
// that finds its own sub-list full takes the next place there (one global atomic on the query's spill counter -- rare on shuffled
// galleries).  pf_spill_counts_kernel afterwards turns the counters into what the refinement reads:
// min(count, cap) per real sub-list, the spill region cut into `spill` full / partial / empty virtual sub-lists (cap + 1 in the last
// one when even the spill region overflowed: the query is redone exactly).
__device__ __forceinline__ void pf_spill_append(const PfArgs &fa, int64_t query, int nsub, uint2 entry)
{
    const unsigned s = atomicAdd(&fa.spill_cnt[query], 1u);
    if ((int64_t)s < (int64_t)fa.spill * fa.cap) fa.spill_lists[query * fa.spill * fa.cap + s] = entry;
}

__global__ __launch_bounds__(256) void pf_spill_counts_kernel(unsigned *__restrict__ rowcnt, const unsigned *__restrict__ spill_cnt, int64_t queries, int nsub,
                                                              int spill, int64_t cap)
{
    // one thread per counter word (coalesced): real sub-lists are clamped in place, virtual ones derived from the query's spill count
    const int nct = nsub + spill;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= queries * nct) return;
    const int64_t q = i / nct;
    const int p = (int)(i - q * nct);
    if (p < nsub) {
        const unsigned c = rowcnt[i];
        if (c > (unsigned)cap) rowcnt[i] = (unsigned)cap;
        return;
    }
    const int64_t S = spill_cnt[q], left = S - (int64_t)(p - nsub) * cap;
    unsigned v = (unsigned)(left <= 0 ? 0 : (left < cap ? left : cap));
    if (p == nct - 1 && S > (int64_t)spill * cap) v = (unsigned)cap + 1u;
    rowcnt[i] = v;
}

int pf_spill_counts(unsigned *rowcnt, const unsigned *spill_cnt, int64_t queries, int nsub, int spill, int64_t cap, hipStream_t s)
{
    if (queries <= 0 || spill <= 0) return SE_OK;
    const int64_t words = queries * (nsub + spill);
    hipLaunchKernelGGL(pf_spill_counts_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, rowcnt, spill_cnt, queries, nsub, spill, cap);
    SE_LAUNCH_CHECK();
    return SE_OK;
}

// ---- tile loop --------------------------------------------------------------------------------------------------------------------
template <int METRIC>
__device__ __forceinline__ float pf_finish(float v, float sa, float sb)
{
    if (METRIC == SE_METRIC_COSINE) return -v;
    if (METRIC == SE_METRIC_EUCLID) return (sa + sb) - 2.0f * v;
    return v;
}

// global -> registers: chunk [k0, k0 + 128) of rows [row0, row0 + NL * 32) of an fp16 matrix with pitch `ld` elements (multiple of 128 columns,
// 16-byte aligned rows; NL pieces per thread = NL * 32 rows).  Rows beyond nrows are clamped (read twice, ignored by the epilogues): no masking anywhere in the loop.
template <int NL>
__device__ __forceinline__ void pf_load(uint4 (&v)[NL], const uint16_t *__restrict__ src, uint32_t ld, int64_t row0, int64_t nrows, int k0)
{
    constexpr int ROWS = NL * PF_THREADS / PF_PPR;
    const int tid = threadIdx.x;
    const char *base = (const char *)(src + row0 * (int64_t)ld);       // uniform
    const int rows_here = (int)((nrows - row0 < ROWS) ? (nrows - row0) : ROWS);
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int p = tid + i * PF_THREADS;
        const int r = p / PF_PPR, c = p % PF_PPR;
        const int rc = r < rows_here ? r : rows_here - 1;
        v[i] = *(const uint4 *)(base + ((uint32_t)rc * ld * 2u + (uint32_t)k0 * 2u + (uint32_t)c * 16u));
    }
}

// pieces [first, first + CNT) of the same chunk (loads dealt over the MFMA steps)
template <int NL, int CNT>
__device__ __forceinline__ void pf_load_part(uint4 (&v)[NL], int first, const uint16_t *__restrict__ src, uint32_t ld, int64_t row0, int64_t nrows, int k0)
{
    constexpr int ROWS = NL * PF_THREADS / PF_PPR;
    const int tid = threadIdx.x;
    const char *base = (const char *)(src + row0 * (int64_t)ld);
    const int rows_here = (int)((nrows - row0 < ROWS) ? (nrows - row0) : ROWS);
#pragma unroll
    for (int i = 0; i < NL; i++) {
        if (i < first || i >= first + CNT) continue;
        const int p = tid + i * PF_THREADS;
        const int r = p / PF_PPR, c = p % PF_PPR;
        const int rc = r < rows_here ? r : rows_here - 1;
        v[i] = *(const uint4 *)(base + ((uint32_t)rc * ld * 2u + (uint32_t)k0 * 2u + (uint32_t)c * 16u));
    }
}

template <int NL>
__device__ __forceinline__ void pf_stage(char *lds, const uint4 (&v)[NL])
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int p = tid + i * PF_THREADS;
        *(uint4 *)(lds + (p / PF_PPR) * PF_PITCH + (p % PF_PPR) * 16) = v[i];
    }
}

// ---- tile loop ------------------------------------------------------------------------------------------------------------------
// Work is cut into JOBS = (query tile, gallery sub-range): a workgroup keeps ONE tile of 128 queries and walks the gallery tiles
// t0 + j, t0 + j + gj, ... of one of `parts` contiguous gallery ranges.  Why query-stationary:
//   * a query's candidates of one job are appended by ONE workgroup: the slot counters live in LDS (a returning LDS atomic instead of
//     a global round trip in every tile epilogue) and the job's sub-list [query][sub-list][cap] fills front to back from one XCD, so its
//     lines are completed in that XCD's L2 instead of being written 8 bytes at a time from eight L2s;
//   * all-pairs calls need no mirrored pass (the first version walked the upper triangle and filtered every tile in both orientations:
//     two global atomic round trips per tile, 3.2 ms of a 4.4 ms call at 50k x 50k x 100 with the matrix pipe 1 % busy; the matrix
//     work saved was that 1 %).
// Which jobs run TOGETHER is what the L2 sees.  The workgroups resident on one XCD (blockIdx % 8 selects the XCD) form a gi x gj
// grid and take one SUPER-JOB = (gi consecutive query tiles) x (one gallery range) at a time: workgroup (i, j) owns query tile i and the
// gallery tiles j, j + gj, ... -- at any moment the XCD works on gi query panels and gj gallery tiles, every panel is streamed by gj
// (resp. gi) workgroups at about the same K position, and only 1 / gi (1 / gj) of those reads leave the L2.  (One workgroup per query
// tile walking a whole range alone -- the first query-stationary version -- made every workgroup stream a private query panel out of
// Infinity Cache: 31.8 ms for the D = 1000 shard against 25.7 ms of the grouped order it replaced.)
// A query's list therefore consists of parts * gj sub-lists.
struct PfJob { int sj, p, tn, t, t1; };

__device__ __forceinline__ bool pf_next_job(PfJob &jb, int step, int nsq, int nsuper, int gi, int gj, int i, int j, int tiles_m, int tiles_n, int tpp)
{
    // advance to this workgroup's next super-job in which it has a query tile AND at least one gallery tile
    for (int sj = jb.sj + step; sj < nsuper; sj += 8) {
        const int p = sj / nsq, tq = sj - p * nsq;
        const int tn = tq * gi + i;
        const int t0 = p * tpp;
        const int t1 = t0 + tpp < tiles_m ? t0 + tpp : tiles_m;
        if (tn < tiles_n && t0 + j < t1) { jb.sj = sj; jb.p = p; jb.tn = tn; jb.t = t0 + j; jb.t1 = t1; return true; }
    }
    return false;
}

template <int METRIC, int EPI>
__global__ __launch_bounds__(PF_THREADS, PF_WGS_PER_CU) void pf_tile_kernel(
    const uint16_t *__restrict__ A, uint32_t lda, const uint16_t *__restrict__ B, uint32_t ldb, const float *__restrict__ sqa,
    const float *__restrict__ sqb, int64_t NA, int64_t NB, int nchunks, int tiles_m, int tiles_n, int parts, int tpp, int gi, int gj,
    const unsigned *__restrict__ ctl_a, const unsigned *__restrict__ ctl_b, PfArgs fa)
{
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];
    char *sA = pf_smem, *sB = pf_smem + PF_BM * PF_PITCH + PF_GAP;
    // side arrays.  Per job (query columns): tCmpCol = the constant the raw accumulator is compared with, tSqCol = |q|^2, jobCnt = slot counters;
    // per tile (gallery rows): tSqRow = |g|^2, tCmpRow = its share of the Euclidean compare constant
    float *tCmpCol = (float *)(pf_smem + (PF_BM + PF_BN) * PF_PITCH + PF_GAP), *tSqCol = tCmpCol + PF_BN, *tSqRow = tSqCol + PF_BN, *tCmpRow = tSqRow + PF_BM;
    unsigned *jobCnt = (unsigned *)(tCmpRow + PF_BM);
    static_assert(PF_THREADS * 36 * 4 <= PF_BM * PF_PITCH + PF_GAP, "the epilogue's half dump must fit in front of the query operand");
    static_assert(PF_BN + PF_BM <= PF_THREADS, "side arrays are filled by one thread per entry");

    const int xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3;
    const int gi_i = slot_in_xcd / gj, gj_j = slot_in_xcd - gi_i * gj;
    const int nsq = (tiles_n + gi - 1) / gi, nsuper = nsq * parts;
    const int nsub = parts * gj;                                    // sub-lists per query
    PfJob cur = {xcd - 8, 0, 0, 0, 0};
    if (gi_i >= gi || !pf_next_job(cur, 8, nsq, nsuper, gi, gj, gi_i, gj_j, tiles_m, tiles_n, tpp)) return;
    PfJob nx = cur;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;        // (PF_BM / 64) x 2 waves, 64 x 64 outputs each
    const int col = lane & 31, hi = lane >> 5;
    const int esum = (int)ctl_a[4] + (int)ctl_b[4];
    const float unscale = ldexpf(1.0f, -esum), rescale = ldexpf(1.0f, esum);     // the images carry 2^ea, 2^eb: exact to undo

    pf_f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mi][j][r] = 0.f;

    const char *pa = sA + (wm * 64 + col) * PF_PITCH + hi * 16;     // block mi: + mi * 32 rows; k16 step s: + s * 32 bytes
    const char *pb = sB + (wn * 64 + col) * PF_PITCH + hi * 16;

    uint4 ra[PF_NLOAD_A], rb[PF_NLOAD_B];
    pf_load<PF_NLOAD_A>(ra, A, lda, (int64_t)cur.t * PF_BM, NA, 0);
    pf_load<PF_NLOAD_B>(rb, B, ldb, (int64_t)cur.tn * PF_BN, NB, 0);

    // Compare constants: a value passes when d~ <= thr (or d~ is NaN).  With a = the raw accumulator (a 2^-esum = the dot product):
    //   cosine     d~ = -a 2^-esum <= thr                     <=>  a >= -thr 2^esum                       =: tCmpCol[q]
    //   Euclidean  d~ = (sg + sq) - 2 a 2^-esum <= thr        <=>  a >= (sq - thr) 2^(esum-1) + sg 2^(esum-1)  =: tCmpCol[q] + tCmpRow[g]
    // (powers of two: exact; the one rounding of the Euclidean sum is covered by the 0.05 eps the thresholds carry above the 2 eps needed).
    // "not less than" is true for NaN accumulators (irregular rows); a NaN / invalid threshold becomes +inf: nothing but NaN passes.
#define PF_JOB_SIDE(TN_)                                                                                             \
    if (threadIdx.x < PF_BN) {                                                                                       \
        const int64_t qc_ = (int64_t)(TN_) * PF_BN + threadIdx.x;                                                    \
        const bool ok_ = qc_ < NB;                                                                                   \
        const float sq_ = METRIC == SE_METRIC_EUCLID ? sqb[ok_ ? qc_ : NB - 1] : 0.f;                                \
        if (EPI == PF_FILTER) {                                                                                      \
            const float th_ = ok_ ? fa.thr[qc_] : __builtin_nanf("");                                                \
            float c_ = METRIC == SE_METRIC_EUCLID ? (sq_ - th_) * (0.5f * rescale) : -th_ * rescale;                 \
            tCmpCol[threadIdx.x] = c_ == c_ ? c_ : __builtin_inff();                                                 \
            jobCnt[threadIdx.x] = 0;                                                                                 \
        }                                                                                                            \
        if (METRIC == SE_METRIC_EUCLID) tSqCol[threadIdx.x] = sq_;                                                   \
    }
#define PF_TILE_SIDE(T_)                                                                                             \
    if (METRIC == SE_METRIC_EUCLID && threadIdx.x >= PF_BN && threadIdx.x < PF_BN + PF_BM) {                         \
        const int64_t gr_ = (int64_t)(T_) * PF_BM + (threadIdx.x - PF_BN);                                           \
        const float sg_ = sqa[(gr_ < NA ? gr_ : NA - 1) * fa.sqa_stride];                                            \
        tSqRow[threadIdx.x - PF_BN] = sg_;                                                                           \
        tCmpRow[threadIdx.x - PF_BN] = sg_ * (0.5f * rescale);                                                       \
    }
#ifdef SE_TUNING
    uint64_t t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = fa.prof ? __builtin_amdgcn_s_memtime() : 0;
#define PF_T(i) if (fa.prof) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_acc[i] += now - t_last; t_last = now; }
#else
#define PF_T(i)
#endif
    int c = 0;
    pf_stage<PF_NLOAD_A>(sA, ra);
    pf_stage<PF_NLOAD_B>(sB, rb);
    PF_JOB_SIDE(cur.tn)
    PF_TILE_SIDE(cur.t)
    wg_barrier();
    bool more = true;
#pragma unroll 1
    while (more) {
        PF_T(0)
        // ---- request the next chunk: same tile, next tile of the job, or first tile of this workgroup's next job ----
        const bool last_chunk = (c + 1 == nchunks);
        bool have_next = true, job_ends = false;
        int nc = c + 1;
        if (last_chunk) {
            nc = 0;
            nx.t = cur.t + gj;
            if (nx.t >= cur.t1) {
                job_ends = true;
                have_next = pf_next_job(nx, 8, nsq, nsuper, gi, gj, gi_i, gj_j, tiles_m, tiles_n, tpp);
            }
        }
#ifndef SE_PF_SPREAD
#define SE_PF_SPREAD 0
#endif
        // one-chunk jobs (padded width 128, e.g. D = 100): the job's query panel is already in LDS and stays there -- every tile of the
        // job used to fetch and stage it again (a third of the operand traffic and LDS writes of the 391 tiles of a 50k x 50k job)
        const bool keep_b = nchunks == 1 && !job_ends;
        if (have_next && !SE_PF_SPREAD) {
            pf_load<PF_NLOAD_A>(ra, A, lda, (int64_t)nx.t * PF_BM, NA, nc * PF_BK);
            if (!keep_b) pf_load<PF_NLOAD_B>(rb, B, ldb, (int64_t)nx.tn * PF_BN, NB, nc * PF_BK);
        }
        PF_T(1)
        // ---- MFMA over the chunk in LDS: 8 steps of k = 16 ----
        // (round 6: the steps behind the last column with data multiply zeros -- C + 0 = C bit for bit, the accumulators are never -0 --
        // and are skipped: D = 100 padded to 128 runs seven steps, not eight)
        const int nsteps = last_chunk ? fa.last_steps : PF_BK / 16;
#pragma unroll
        for (int s = 0; s < PF_BK / 16; s++) {
            if (s >= nsteps) break;
#if SE_PF_SPREAD
            // experiment: the next chunk's loads dealt over the MFMA steps instead of one burst in front of them
            if (have_next) {
                constexpr int PA = PF_NLOAD_A / (PF_BK / 16) > 0 ? PF_NLOAD_A / (PF_BK / 16) : 1, PB = PF_NLOAD_B / (PF_BK / 16) > 0 ? PF_NLOAD_B / (PF_BK / 16) : 1;
                pf_load_part<PF_NLOAD_A, PA>(ra, s * PA, A, lda, (int64_t)nx.t * PF_BM, NA, nc * PF_BK);
                pf_load_part<PF_NLOAD_B, PB>(rb, s * PB, B, ldb, (int64_t)nx.tn * PF_BN, NB, nc * PF_BK);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            f16x8 a0 = __builtin_bit_cast(f16x8, *(const uint4 *)(pa + s * 32));
            f16x8 a1 = __builtin_bit_cast(f16x8, *(const uint4 *)(pa + 32 * PF_PITCH + s * 32));
            f16x8 b0 = __builtin_bit_cast(f16x8, *(const uint4 *)(pb + s * 32));
            f16x8 b1 = __builtin_bit_cast(f16x8, *(const uint4 *)(pb + 32 * PF_PITCH + s * 32));
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
#if SE_PF_SPREAD
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        PF_T(2)
        // the next chunk's operands are waited for HERE (value barriers: no use of a loaded register in front of the MFMA phase)
#pragma unroll
        for (int i = 0; i < PF_NLOAD_A; i++) asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));
#pragma unroll
        for (int i = 0; i < PF_NLOAD_B; i++) asm volatile("" : "+v"(rb[i].x), "+v"(rb[i].y), "+v"(rb[i].z), "+v"(rb[i].w));
        PF_T(3)
        if (last_chunk) {
            // ---- tile finished.  acc[mi][j][r]: gallery row  cur_m0 + wm*64 + mi*32 + (r&3) + 8*(r>>2) + 4*hi,
            //                                      query       cur_n0 + wn*64 + j*32 + col ----
            const int64_t cur_m0 = (int64_t)cur.t * PF_BM, cur_n0 = (int64_t)cur.tn * PF_BN;
            const int rows_here = (int)((NA - cur_m0 < PF_BM) ? (NA - cur_m0) : PF_BM);
            const int cols_here = (int)((NB - cur_n0 < PF_BN) ? (NB - cur_n0) : PF_BN);
            const bool full_rows = rows_here == PF_BM;
            const int lr0 = wm * 64 + 4 * hi;
#define PF_SA(MI_, R) (METRIC == SE_METRIC_EUCLID ? tSqRow[lr0 + (MI_) * 32 + ((R) & 3) + 8 * ((R) >> 2)] : 0.f)
#define PF_VAL(MI_, J, R) pf_finish<METRIC>(acc[MI_][J][R] * unscale, PF_SA(MI_, R), sbq)
            if (EPI == PF_STORE) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int lc = wn * 64 + j * 32 + col;
                    const float sbq = METRIC == SE_METRIC_EUCLID ? tSqCol[lc] : 0.f;
#pragma unroll
                    for (int mi = 0; mi < 2; mi++)
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int lr = lr0 + mi * 32 + (r & 3) + 8 * (r >> 2);
                            if (lr < rows_here && lc < cols_here) fa.out[(cur_m0 + lr) * fa.ldo + cur_n0 + lc] = PF_VAL(mi, j, r);
                        }
                }
            } else if (EPI == PF_GROUPMIN) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int lc = wn * 64 + j * 32 + col;
                    const bool qok = lc < cols_here;
                    const int64_t qg = cur_n0 + (qok ? lc : cols_here - 1);
                    const float sbq = METRIC == SE_METRIC_EUCLID ? tSqCol[lc] : 0.f;
#pragma unroll
                    for (int mi = 0; mi < 2; mi++) {
                        float m = __builtin_inff();
                        bool any = false;
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int lr = lr0 + mi * 32 + (r & 3) + 8 * (r >> 2);
                            const float v = PF_VAL(mi, j, r);
                            const bool ok = (full_rows || lr < rows_here) && (v == v);
                            m = (ok && v < m) ? v : m;
                            any = any || ok;
                        }
                        if (!any) m = __builtin_nanf("");     // a group of NaNs only: sorted last by the threshold kernel
                        const int64_t grp = (cur_m0 / PF_BM) * (PF_BM / 16) + (wm * 2 + mi) * 2 + hi;     // 16 gallery rows per group minimum
                        if (qok && grp < fa.gm_ld) fa.gm[qg * fa.gm_ld + grp] = m;
                    }
                }
            } else {
                // ---- PF_FILTER.  Lanes = queries (tile columns), registers = gallery rows: 2 queries x 32 values per lane, ~1 % of them
                //      candidates.  The scan costs TWO vector instructions per value and nothing else:
                //        v_cmp_nlt_f32  vcc, acc, c        (the raw accumulator against the query's constant; true for NaN)
                //        v_addc_co_u32  mask, mask, mask   (mask = 2 mask + vcc)
                //      -> a 32-bit pass mask per (lane, query), value i = mi * 16 + r at bit 31 - i.  Slots of the job's sub-list come from
                //      ONE returning LDS atomic per (lane, query).  The rare candidates are then visited bit by bit: registers cannot be
                //      indexed per lane, so the wave's accumulators are dumped into the operand LDS (free between this tile's last MFMA
                //      and the next stage; 16 ds_write_b128 per lane, lane-private rows of 64 + 4 dwords) and read back by index.
                //      (Before: a compare, an exec-mask block and a branch per VALUE, ~50 cycles each: 67 % of the kernel at D = 100.) ----
                uint32_t mask[2] = {0u, 0u};
                if (METRIC != SE_METRIC_EUCLID) {
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const float cq = tCmpCol[wn * 64 + j * 32 + col];
#pragma unroll
                        for (int mi = 0; mi < 2; mi++)
#pragma unroll
                            for (int r = 0; r < 16; r++)
                                asm volatile("v_cmp_nlt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask[j]) : "v"(acc[mi][j][r]), "v"(cq) : "vcc");
                    }
                } else {
                    // Euclidean compare constants differ per value (query part + row part: an add in front of every compare); four independent
                    // chains (2 queries x 2 row blocks) advance together, each compare into its own SGPR pair (3.68 -> 3.58 ms at 50k x 50k x 100;
                    // the cosine form above measured 0.2 ms SLOWER this way and keeps its single chain through VCC)
                    const float cq0 = tCmpCol[wn * 64 + col], cq1 = tCmpCol[wn * 64 + 32 + col];
                    uint32_t pm00 = 0u, pm01 = 0u, pm10 = 0u, pm11 = 0u;      // pm[j][mi]: 16 verdicts each
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        float c00 = cq0, c01 = cq0, c10 = cq1, c11 = cq1;
                        if (METRIC == SE_METRIC_EUCLID) {
                            const float r0 = tCmpRow[lr0 + (r & 3) + 8 * (r >> 2)], r1 = tCmpRow[lr0 + 32 + (r & 3) + 8 * (r >> 2)];
                            c00 = cq0 + r0; c01 = cq0 + r1; c10 = cq1 + r0; c11 = cq1 + r1;
                        }
                        uint64_t s0, s1, s2, s3;
                        asm volatile("v_cmp_nlt_f32_e64 %4, %8, %12\n\tv_cmp_nlt_f32_e64 %5, %9, %13\n\tv_cmp_nlt_f32_e64 %6, %10, %14\n\tv_cmp_nlt_f32_e64 %7, %11, %15\n\t"
                                     "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n\tv_addc_co_u32_e64 %1, %5, %1, %1, %5\n\t"
                                     "v_addc_co_u32_e64 %2, %6, %2, %2, %6\n\tv_addc_co_u32_e64 %3, %7, %3, %3, %7"
                                     : "+v"(pm00), "+v"(pm01), "+v"(pm10), "+v"(pm11), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3)
                                     : "v"(acc[0][0][r]), "v"(acc[1][0][r]), "v"(acc[0][1][r]), "v"(acc[1][1][r]), "v"(c00), "v"(c01), "v"(c10), "v"(c11));
                    }
                    mask[0] = (pm00 << 16) | pm01;           // value i = mi * 16 + r at bit 31 - i
                    mask[1] = (pm10 << 16) | pm11;
                }
                if (!full_rows) {       // last gallery tile: rows beyond the gallery were clamped reads
                    uint32_t vm = 0u;
#pragma unroll
                    for (int i = 0; i < 32; i++) vm |= (lr0 + (i >> 4) * 32 + (i & 3) + 8 * ((i >> 2) & 3) < rows_here) ? (0x80000000u >> i) : 0u;
                    mask[0] &= vm; mask[1] &= vm;
                }
                unsigned slotj[2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if (wn * 64 + j * 32 + col >= cols_here) mask[j] = 0u;
                    slotj[j] = 0;
                    if (mask[j]) slotj[j] = atomicAdd(&jobCnt[wn * 64 + j * 32 + col], (unsigned)__popc(mask[j]));
                }
                PF_T(4)
                wg_barrier();       // every wave is done with the operands of the last chunk: their LDS becomes the dump
                float *mine = (float *)pf_smem + threadIdx.x * 36;          // lane-private row of 32 + 4 dwords, one query's values at a time
                PF_T(5)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    uint32_t m = mask[j];
                    if (j) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the row is private: no barrier between the two halves)
#pragma unroll
                    for (int mi = 0; mi < 2; mi++)
#pragma unroll
                        for (int g = 0; g < 4; g++)
                            *(float4 *)(mine + mi * 16 + g * 4) = make_float4(acc[mi][j][4 * g], acc[mi][j][4 * g + 1], acc[mi][j][4 * g + 2], acc[mi][j][4 * g + 3]);
                    unsigned slot = slotj[j];
                    const int lc = wn * 64 + j * 32 + col;
                    uint2 *lst = fa.lists + ((cur_n0 + lc) * (int64_t)nsub + (cur.p * gj + gj_j)) * fa.cap;
                    const float sbq = METRIC == SE_METRIC_EUCLID ? tSqCol[lc] : 0.f;
                    auto emit = [&](int i, float a) {
                        const int lr = lr0 + (i >> 4) * 32 + (i & 3) + 8 * ((i >> 2) & 3);
                        const float v = pf_finish<METRIC>(a * unscale, METRIC == SE_METRIC_EUCLID ? tSqRow[lr] : 0.f, sbq);
                        if (__builtin_expect(slot < (unsigned)fa.cap, 1)) lst[slot] = make_uint2(__float_as_uint(v), (uint32_t)(cur_m0 + lr));
                        else pf_spill_append(fa, cur_n0 + lc, nsub, make_uint2(__float_as_uint(v), (uint32_t)(cur_m0 + lr)));
                        slot++;
                    };
                    // the first PF_CB candidates of every lane in straight-line code: their LDS reads are in flight together (a lane holds
                    // 0.5 candidates per query and tile on average, the busiest lane of a wave ~3: the divergent loop below paid one LDS
                    // round trip per iteration); the loop takes what is left
                    constexpr int PF_CB = SE_PF_CB;
                    int ci[PF_CB > 0 ? PF_CB : 1];
                    float ca[PF_CB > 0 ? PF_CB : 1];
#pragma unroll
                    for (int t = 0; t < PF_CB; t++) {
                        const int i = m ? __builtin_clz(m) : -1;                          // value index: mi = i >> 4, r = i & 15
                        m = m ? (m & ~(0x80000000u >> i)) : 0u;
                        ci[t] = i;
                        ca[t] = mine[i >= 0 ? i : 0];
                    }
#pragma unroll
                    for (int t = 0; t < PF_CB; t++)
                        if (ci[t] >= 0) emit(ci[t], ca[t]);
                    while (m) {
                        const int i = __builtin_clz(m);
                        m &= ~(0x80000000u >> i);
                        emit(i, mine[i]);
                    }
                }
                PF_T(6)
            }
#undef PF_VAL
#undef PF_SA
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[mi][j][r] = 0.f;
        }
        wg_barrier();   // every wave has finished reading this chunk out of LDS (and, at a tile's end, its epilogue)
        PF_T(10)
        if (last_chunk && job_ends && EPI == PF_FILTER && threadIdx.x < PF_BN) {
            // the job is complete: its slot counters -> rowcnt[query][sub-list] (counts above the capacity mark an overflow: the query is redone)
            const int64_t qc = (int64_t)cur.tn * PF_BN + threadIdx.x;
            if (qc < NB) fa.rowcnt[qc * (nsub + fa.spill) + (cur.p * gj + gj_j)] = jobCnt[threadIdx.x];
        }
        if (have_next) {
            pf_stage(sA, ra);
            if (!keep_b) pf_stage(sB, rb);
            if (last_chunk) {
                if (job_ends) { PF_JOB_SIDE(nx.tn) }
                PF_TILE_SIDE(nx.t)
                cur = nx;
            }
        }
        wg_barrier();
        c = last_chunk ? 0 : c + 1;
        more = have_next;
        PF_T(11)
    }
#ifdef SE_TUNING
    if (fa.prof && threadIdx.x == 0)
        for (int i = 0; i < 12; i++) atomicAdd(&fa.prof[i], (unsigned long long)t_acc[i]);
#endif
#undef PF_T
#undef PF_JOB_SIDE
#undef PF_TILE_SIDE
}

// ---- big-tile filter kernel (long rows: kp >= 256) -------------------------------------------------------------------------------
// The 128 x 128 kernel above needs three pipes of a CU at 100 % at the same moment to keep the matrix core busy: with 64 x 64 outputs
// per wave every v_mfma_f32_32x32x16_f16 (32 cycles) reads 1 KB of fragments out of LDS (128 B / clk = the LDS peak) and every
// 128 x 128 x 128 chunk step (1024 matrix cycles per SIMD) brings 64 KB through the L1 (64 B / clk = its fill peak) -- measured: 17 % of
// the fp16 peak on the D = 1000 shard, `prefetch-issue` (vector-memory issue stalls) the largest phase.  This kernel halves both ratios:
// 256 x 256 tiles, FOUR waves of 128 x 128 outputs (4 x 4 MFMA blocks, 256 accumulators in the AGPR half of a lone wave's 512
// registers -- one workgroup per CU).  With one wave per SIMD nothing else hides a wave's own memory phases, so the operands do not
// pass through registers at all: global_load_lds_dwordx4 writes K-chunks of 64 straight into a double-buffered LDS image (2 x 64 KB,
// rows of 128 bytes, the 16-byte pieces XOR-swizzled on the SOURCE side so that fragment reads are conflict-free without padding),
// requested one chunk ahead and in flight under the 64 MFMAs of the current chunk; ONE barrier per chunk.  Filter epilogue only (the
// sample pass is small and keeps the kernel above); job / super-job order, slot counters and list format are the same.
constexpr int PB_BM = 256, PB_BN = 256, PB_BK = 64, PB_THREADS = 256;
constexpr int PB_ROWB = PB_BK * 2;                          // bytes of one operand row of a chunk (128) = its LDS pitch
constexpr int PB_STAGE = (PB_BM + PB_BN) * PB_ROWB;         // bytes of one LDS stage (65,536)
constexpr int PB_NISSUE = PB_BM * PB_ROWB / (PB_THREADS * 16);   // load instructions per wave, chunk and operand (8; 8 rows each)
static_assert(PB_BM == PB_BN, "one loader for both operands");

typedef __attribute__((address_space(1))) const void pb_gptr_t;
typedef __attribute__((address_space(3))) void pb_lptr_t;

// byte offsets (from the tile's first row) of the 16-byte pieces this lane fetches: instruction i covers tile rows wave * 64 + i * 8 + [0, 8),
// lane l row + (l >> 3), LDS slot l & 7 <- source piece (l & 7) ^ ((row >> 1) & 7).  Rows beyond the matrix: clamped (masked by the epilogue).
__device__ __forceinline__ void pb_offsets(uint32_t (&off)[PB_NISSUE], uint32_t ld, int64_t row0, int64_t nrows, int wave, int lane)
{
    const int rows_here = (int)((nrows - row0 < PB_BM) ? (nrows - row0) : PB_BM);
#pragma unroll
    for (int i = 0; i < PB_NISSUE; i++) {
        const int r = wave * 64 + i * 8 + (lane >> 3);
        const int rc = r < rows_here ? r : rows_here - 1;
        off[i] = (uint32_t)rc * ld * 2u + (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
    }
}

// chunk [k0, k0 + 64) of one operand tile -> LDS image `dst` (wave-uniform; this wave's 64 rows)
__device__ __forceinline__ void pb_issue(const char *base, const uint32_t (&off)[PB_NISSUE], int k0, char *dst, int wave)
{
#pragma unroll
    for (int i = 0; i < PB_NISSUE; i++)
        __builtin_amdgcn_global_load_lds((pb_gptr_t *)(base + (off[i] + (uint32_t)k0 * 2u)), (pb_lptr_t *)(dst + (wave * 64 + i * 8) * PB_ROWB), 16, 0, 0);
}

template <int METRIC>
__global__ __launch_bounds__(PB_THREADS, 1) void pf_big_kernel(
    const uint16_t *__restrict__ A, uint32_t lda, const uint16_t *__restrict__ B, uint32_t ldb, const float *__restrict__ sqa,
    const float *__restrict__ sqb, int64_t NA, int64_t NB, int nchunks, int tiles_m, int tiles_n, int parts, int tpp, int gi, int gj,
    const unsigned *__restrict__ ctl_a, const unsigned *__restrict__ ctl_b, PfArgs fa)
{
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];
    // two stages of [A rows | B rows]; side arrays behind them (see pf_tile_kernel)
    float *tCmpCol = (float *)(pf_smem + 2 * PB_STAGE), *tSqCol = tCmpCol + PB_BN, *tSqRow = tSqCol + PB_BN, *tCmpRow = tSqRow + PB_BM;
    unsigned *jobCnt = (unsigned *)(tCmpRow + PB_BM);
    static_assert(PB_THREADS * 36 * 4 <= PB_STAGE, "the epilogue's half dump must fit one stage");
    static_assert(PB_BN <= PB_THREADS && PB_BM <= PB_THREADS, "side arrays are filled by one thread per entry");

    const int xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3;
    const int gi_i = slot_in_xcd / gj, gj_j = slot_in_xcd - gi_i * gj;
    const int nsq = (tiles_n + gi - 1) / gi, nsuper = nsq * parts;
    const int nsub = parts * gj;
    PfJob cur = {xcd - 8, 0, 0, 0, 0};
    if (gi_i >= gi || !pf_next_job(cur, 8, nsq, nsuper, gi, gj, gi_i, gj_j, tiles_m, tiles_n, tpp)) return;
    PfJob nx = cur;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave >> 1, wn = wave & 1;        // 2 x 2 waves, 128 x 128 outputs each
    const int col = lane & 31, hi = lane >> 5;
    const int esum = (int)ctl_a[4] + (int)ctl_b[4];
    const float unscale = ldexpf(1.0f, -esum), rescale = ldexpf(1.0f, esum);

    // fragment reads: operand row (wm | wn) * 128 + block * 32 + col (block: an immediate offset of 32 rows), k16 step s = pieces 2 s + hi,
    // stored at slot piece ^ ((row >> 1) & 7) -- the same for all four blocks (32 rows apart)
    const int swz = (col >> 1) & 7;
    const int rowa = (wm * 128 + col) * PB_ROWB, rowb = PB_BM * PB_ROWB + (wn * 128 + col) * PB_ROWB;
    int fo[PB_BK / 16];
#pragma unroll
    for (int s = 0; s < PB_BK / 16; s++) fo[s] = ((2 * s + hi) ^ swz) * 16;

    uint32_t offa[PB_NISSUE], offb[PB_NISSUE];
    pb_offsets(offa, lda, (int64_t)cur.t * PB_BM, NA, wave, lane);
    pb_offsets(offb, ldb, (int64_t)cur.tn * PB_BN, NB, wave, lane);
    const char *basea = (const char *)(A + (int64_t)cur.t * PB_BM * lda), *baseb = (const char *)(B + (int64_t)cur.tn * PB_BN * ldb);
    pb_issue(basea, offa, 0, pf_smem, wave);
    pb_issue(baseb, offb, 0, pf_smem + PB_BM * PB_ROWB, wave);

#define PB_JOB_SIDE(TN_)                                                                                             \
    {                                                                                                                \
        const int64_t qc_ = (int64_t)(TN_) * PB_BN + threadIdx.x;                                                    \
        const bool ok_ = qc_ < NB;                                                                                   \
        const float sq_ = METRIC == SE_METRIC_EUCLID ? sqb[ok_ ? qc_ : NB - 1] : 0.f;                                \
        const float th_ = ok_ ? fa.thr[qc_] : __builtin_nanf("");                                                    \
        float c_ = METRIC == SE_METRIC_EUCLID ? (sq_ - th_) * (0.5f * rescale) : -th_ * rescale;                     \
        tCmpCol[threadIdx.x] = c_ == c_ ? c_ : __builtin_inff();                                                     \
        jobCnt[threadIdx.x] = 0;                                                                                     \
        if (METRIC == SE_METRIC_EUCLID) tSqCol[threadIdx.x] = sq_;                                                   \
    }
#define PB_TILE_SIDE(T_)                                                                                             \
    if (METRIC == SE_METRIC_EUCLID) {                                                                                \
        const int64_t gr_ = (int64_t)(T_) * PB_BM + threadIdx.x;                                                     \
        const float sg_ = sqa[(gr_ < NA ? gr_ : NA - 1) * fa.sqa_stride];                                            \
        tSqRow[threadIdx.x] = sg_;                                                                                   \
        tCmpRow[threadIdx.x] = sg_ * (0.5f * rescale);                                                               \
    }
#ifdef SE_TUNING
    uint64_t t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = fa.prof ? __builtin_amdgcn_s_memtime() : 0;
#define PF_T(i) if (fa.prof) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_acc[i] += now - t_last; t_last = now; }
#else
#define PF_T(i)
#endif
    int buf = 0;
    PB_JOB_SIDE(cur.tn)
    PB_TILE_SIDE(cur.t)
    // One tile per iteration of the outer loop, its K-chunks in the inner one.  (The accumulators are cleared in front of the inner loop
    // and read behind it: written as ONE flat loop over chunks with a conditional epilogue, the compiler carried all 256 of them through the
    // loop in VGPRs, copied them into AGPRs around every MFMA and spilled 200-340 registers.)
#pragma unroll 1
    while (true) {
        pf_f32x16 acc[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mi][j][r] = 0.f;
        bool have_next = true, job_ends = false;
#pragma unroll 1
        for (int c = 0; c < nchunks; c++) {
            PF_T(0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of the chunk have landed in stage `buf`
            wg_barrier();                                       // ... everyone's have, and everyone is done reading the other stage
            PF_T(10)
            // ---- request the next chunk into the other stage: same tile, next tile of the job, or first tile of this workgroup's next job ----
            const bool last_chunk = (c + 1 == nchunks);
            int nc = c + 1;
            if (last_chunk) {
                nc = 0;
                nx.t = cur.t + gj;
                if (nx.t >= cur.t1) {
                    job_ends = true;
                    have_next = pf_next_job(nx, 8, nsq, nsuper, gi, gj, gi_i, gj_j, tiles_m, tiles_n, tpp);
                }
                if (have_next) {
                    pb_offsets(offa, lda, (int64_t)nx.t * PB_BM, NA, wave, lane);
                    basea = (const char *)(A + (int64_t)nx.t * PB_BM * lda);
                    if (job_ends) {
                        pb_offsets(offb, ldb, (int64_t)nx.tn * PB_BN, NB, wave, lane);
                        baseb = (const char *)(B + (int64_t)nx.tn * PB_BN * ldb);
                    }
                }
            }
            // (nothing follows this workgroup's last chunk: the requests below then fetch chunk 0 of the current tile again -- valid
            //  addresses, a stage nobody reads -- so that the loop body has no branch around its loads)
            char *st = pf_smem + (buf ^ 1) * PB_STAGE;
            const uint32_t kb2 = (uint32_t)nc * (PB_BK * 2u);
            PF_T(1)
            // ---- MFMA over the chunk in this stage: 4 steps of k = 16, 16 blocks each; step s + 1's fragments are requested in front of
            //      step s's MFMAs (one wave per SIMD: nobody else hides the LDS latency), and the 16 load instructions of the next chunk
            //      are dealt one per four MFMAs: issued in a burst they filled the memory pipe's queue and the wave sat in front of its
            //      MFMAs until the queue drained (28 % of the kernel) ----
            const char *sa = pf_smem + buf * PB_STAGE + rowa, *sb = pf_smem + buf * PB_STAGE + rowb;
            f16x8 af[2][4], bf[2][4];
#pragma unroll
            for (int mi = 0; mi < 4; mi++) af[0][mi] = __builtin_bit_cast(f16x8, *(const uint4 *)(sa + mi * 32 * PB_ROWB + fo[0]));
#pragma unroll
            for (int j = 0; j < 4; j++) bf[0][j] = __builtin_bit_cast(f16x8, *(const uint4 *)(sb + j * 32 * PB_ROWB + fo[0]));
#pragma unroll
            for (int s = 0; s < PB_BK / 16; s++) {
                if (s + 1 < PB_BK / 16) {
#pragma unroll
                    for (int mi = 0; mi < 4; mi++) af[(s + 1) & 1][mi] = __builtin_bit_cast(f16x8, *(const uint4 *)(sa + mi * 32 * PB_ROWB + fo[s + 1]));
#pragma unroll
                    for (int j = 0; j < 4; j++) bf[(s + 1) & 1][j] = __builtin_bit_cast(f16x8, *(const uint4 *)(sb + j * 32 * PB_ROWB + fo[s + 1]));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < 4; mi++) {
#pragma unroll
                    for (int jh = 0; jh < 2; jh++) {
                        // the next chunk's 16 load instructions: one per two MFMAs of the first two steps (issued in a burst they filled the
                        // memory pipe's queue and the wave sat in front of its MFMAs: 28 % of the kernel; spread over all four steps the
                        // last of them landed after the chunk's MFMAs were done)
                        const int li = (s * 4 + mi) * 2 + jh;
                        if (li < PB_NISSUE)
                            __builtin_amdgcn_global_load_lds((pb_gptr_t *)(basea + (offa[li] + kb2)), (pb_lptr_t *)(st + (wave * 64 + li * 8) * PB_ROWB), 16, 0, 0);
                        else if (li < 2 * PB_NISSUE)
                            __builtin_amdgcn_global_load_lds((pb_gptr_t *)(baseb + (offb[li - PB_NISSUE] + kb2)),
                                                             (pb_lptr_t *)(st + PB_BM * PB_ROWB + (wave * 64 + (li - PB_NISSUE) * 8) * PB_ROWB), 16, 0, 0);
#pragma unroll
                        for (int j = 2 * jh; j < 2 * jh + 2; j++)
                            acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s & 1][mi], bf[s & 1][j], acc[mi][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            buf ^= 1;
            PF_T(2)
        }
        {
            // ---- tile finished (its last chunk sat in stage buf ^ 1; the next tile's first chunk is on its way into stage buf).
            //      acc[mi][j][r]: gallery row  cur_m0 + wm*128 + mi*32 + (r&3) + 8*(r>>2) + 4*hi,  query  cur_n0 + wn*128 + j*32 + col.
            //      Scan as in pf_tile_kernel: two instructions per value (+ the AGPR read) -> per (lane, query) a 64-bit pass mask
            //      (two words: blocks 0-1, 2-3) ----
            const int64_t cur_m0 = (int64_t)cur.t * PB_BM, cur_n0 = (int64_t)cur.tn * PB_BN;
            const int rows_here = (int)((NA - cur_m0 < PB_BM) ? (NA - cur_m0) : PB_BM);
            const int cols_here = (int)((NB - cur_n0 < PB_BN) ? (NB - cur_n0) : PB_BN);
            const int lr0 = wm * 128 + 4 * hi;
            uint32_t mask[4][2];
            unsigned slotj[4];
            // (the matrix core's last results are not interlocked against a plain vector read of their registers)
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int lc = wn * 128 + j * 32 + col;
                const float cq = tCmpCol[lc];
                // Four independent chains (one per 32-row block) advance together: AGPR read -> compare into an SGPR pair -> shift the verdict
                // into the block's 16-bit mask.  (One wave per SIMD: a single dependent chain of read / compare / add-with-carry ran at the
                // pipeline's latency, ~24 cycles per value, 11 % of the kernel.)
                uint32_t pm0 = 0u, pm1 = 0u, pm2 = 0u, pm3 = 0u;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float c0 = cq, c1 = cq, c2 = cq, c3 = cq;
                    if (METRIC == SE_METRIC_EUCLID) {
                        const int lrr = lr0 + (r & 3) + 8 * (r >> 2);
                        c0 = cq + tCmpRow[lrr]; c1 = cq + tCmpRow[lrr + 32]; c2 = cq + tCmpRow[lrr + 64]; c3 = cq + tCmpRow[lrr + 96];
                    }
                    float t0, t1, t2, t3;
                    uint64_t s0, s1, s2, s3;
                    asm volatile("v_accvgpr_read_b32 %4, %12\n\tv_accvgpr_read_b32 %5, %13\n\tv_accvgpr_read_b32 %6, %14\n\tv_accvgpr_read_b32 %7, %15\n\t"
                                 "v_cmp_nlt_f32_e64 %8, %4, %16\n\tv_cmp_nlt_f32_e64 %9, %5, %17\n\tv_cmp_nlt_f32_e64 %10, %6, %18\n\tv_cmp_nlt_f32_e64 %11, %7, %19\n\t"
                                 "v_addc_co_u32_e64 %0, %8, %0, %0, %8\n\tv_addc_co_u32_e64 %1, %9, %1, %1, %9\n\t"
                                 "v_addc_co_u32_e64 %2, %10, %2, %2, %10\n\tv_addc_co_u32_e64 %3, %11, %3, %3, %11"
                                 : "+v"(pm0), "+v"(pm1), "+v"(pm2), "+v"(pm3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3),
                                   "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3)
                                 : "a"(acc[0][j][r]), "a"(acc[1][j][r]), "a"(acc[2][j][r]), "a"(acc[3][j][r]), "v"(c0), "v"(c1), "v"(c2), "v"(c3));
                }
                mask[j][0] = (pm0 << 16) | pm1;             // value i = (block & 1) * 16 + r of half h = block >> 1 at bit 31 - i
                mask[j][1] = (pm2 << 16) | pm3;
                if (rows_here != PB_BM) {       // last gallery tile: rows beyond the gallery were clamped reads
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        uint32_t vm = 0u;
#pragma unroll
                        for (int i = 0; i < 32; i++) vm |= (lr0 + (2 * h + (i >> 4)) * 32 + (i & 3) + 8 * ((i >> 2) & 3) < rows_here) ? (0x80000000u >> i) : 0u;
                        mask[j][h] &= vm;
                    }
                }
                if (lc >= cols_here) mask[j][0] = mask[j][1] = 0u;
                const unsigned cnt = (unsigned)(__popc(mask[j][0]) + __popc(mask[j][1]));
                slotj[j] = 0;
                if (cnt) slotj[j] = atomicAdd(&jobCnt[lc], cnt);
            }
            PF_T(4)
            wg_barrier();       // every wave is done with the last chunk's stage: it becomes the dump
            float *mine = (float *)(pf_smem + (buf ^ 1) * PB_STAGE) + threadIdx.x * 36;      // lane-private row of 32 + 4 dwords
            PF_T(5)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int lc = wn * 128 + j * 32 + col;
                uint2 *lst = fa.lists + ((cur_n0 + lc) * (int64_t)nsub + (cur.p * gj + gj_j)) * fa.cap;
                const float sbq = METRIC == SE_METRIC_EUCLID ? tSqCol[lc] : 0.f;
                unsigned slot = slotj[j];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    uint32_t m = mask[j][h];
                    if (__ballot(m != 0u) == 0ull) continue;                            // (wave-uniform) nothing passed in this half
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");               // the row is private: no barrier between rounds
#pragma unroll
                    for (int mm = 0; mm < 2; mm++)
#pragma unroll
                        for (int g = 0; g < 4; g++)
                            *(float4 *)(mine + mm * 16 + g * 4) = make_float4(acc[2 * h + mm][j][4 * g], acc[2 * h + mm][j][4 * g + 1],
                                                                              acc[2 * h + mm][j][4 * g + 2], acc[2 * h + mm][j][4 * g + 3]);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    while (m) {
                        const int i = __builtin_clz(m);                                   // value index: block 2h + (i >> 4), r = i & 15
                        m &= ~(0x80000000u >> i);
                        const int lr = lr0 + (2 * h + (i >> 4)) * 32 + (i & 3) + 8 * ((i >> 2) & 3);
                        const float a = mine[i];
                        const float v = pf_finish<METRIC>(a * unscale, METRIC == SE_METRIC_EUCLID ? tSqRow[lr] : 0.f, sbq);
                        if (__builtin_expect(slot < (unsigned)fa.cap, 1)) lst[slot] = make_uint2(__float_as_uint(v), (uint32_t)(cur_m0 + lr));
                        else pf_spill_append(fa, cur_n0 + lc, nsub, make_uint2(__float_as_uint(v), (uint32_t)(cur_m0 + lr)));
                        slot++;
                    }
                }
            }
            PF_T(6)
            wg_barrier();       // every wave is past its reads of the side arrays and its slot reservations
            if (job_ends) {
                // the job is complete: its slot counters -> rowcnt[query][sub-list] (counts above the capacity mark an overflow: the query is redone)
                const int64_t qc = (int64_t)cur.tn * PB_BN + threadIdx.x;
                if (qc < NB) fa.rowcnt[qc * (nsub + fa.spill) + (cur.p * gj + gj_j)] = jobCnt[threadIdx.x];
            }
            if (!have_next) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the last (unused) requests must have landed before this workgroup's LDS is handed on
                break;
            }
            if (job_ends) PB_JOB_SIDE(nx.tn)          // (thread t read jobCnt[t] above and resets it here: its own entry)
            PB_TILE_SIDE(nx.t)
            cur = nx;
            PF_T(11)
        }
    }
#ifdef SE_TUNING
    if (fa.prof && threadIdx.x == 0)
        for (int i = 0; i < 12; i++) atomicAdd(&fa.prof[i], (unsigned long long)t_acc[i]);
#endif
#undef PF_T
#undef PB_JOB_SIDE
#undef PB_TILE_SIDE
}

static int pf_num_cus()
{
    static const int cus = [] {
        int dev = 0, n = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        return n > 0 ? n : 256;
    }();
    return cus;
}

static int64_t pf_grid()
{
    int64_t grid = (int64_t)pf_num_cus() * PF_WGS_PER_CU;
    return grid / 8 * 8 > 0 ? grid / 8 * 8 : 8;
}

// geometry of a pass over n_a gallery rows x n_q queries: the gi x gj grid of workgroups per XCD (gi query tiles x gj interleaved
// gallery tile sequences per super-job) and the number of contiguous gallery ranges `parts` (>= ~4 super-job rounds per XCD)
// kp > 0: the FILTER pass of a problem with kp padded columns -- long rows take the 256 x 256 kernel (g.big), one workgroup per CU.
PfGeom pf_geometry(int64_t n_a, int64_t n_q, int want_parts, int kp)
{
    PfGeom g;
    // (padded width 128: the two kernels are level -- cosine 3.0 vs 3.1 ms, Euclidean 3.7 vs 3.6 ms at 50k x 50k x 100 -- and the small one stays)
    g.big = (kp >= 256 && n_a >= 4096 && n_q >= 256) ? 1 : 0;
    if (const char *e = tuning_env("SE_PF_BIG")) g.big = (kp > 0 && atoi(e) != 0) ? 1 : 0;      // -DSE_TUNING build: pins the kernel
    const int bm = g.big ? PB_BM : PF_BM, bn = g.big ? PB_BN : PF_BN;
    const int64_t tiles_m = (n_a + bm - 1) / bm, tiles_n = (n_q + bn - 1) / bn;
    const int per_xcd = (int)((g.big ? (int64_t)pf_num_cus() / 8 * 8 : pf_grid()) / 8);
    g.gi = 8;
    if (const char *e = tuning_env("SE_PF_GI")) g.gi = atoi(e) > 0 ? atoi(e) : 8;   // -DSE_TUNING build: shape of the per-XCD workgroup grid
    while (g.gi > 1 && (g.gi > tiles_n || per_xcd % g.gi)) g.gi >>= 1;          // few queries: fewer query tiles side by side, more gallery sequences
    g.gj = per_xcd / g.gi;
    if (g.gj > tiles_m) g.gj = (int)tiles_m;
    if (g.gj < 1) g.gj = 1;
    const int64_t nsq = (tiles_n + g.gi - 1) / g.gi;
    int64_t parts = want_parts > 0 ? want_parts : (4 * 8 + nsq - 1) / nsq;
    if (const char *e = tuning_env("SE_PF_PARTS")) parts = atoi(e);
    const int64_t max_parts = tiles_m / g.gj > 0 ? tiles_m / g.gj : 1;             // every range holds >= gj tiles
    if (parts > max_parts) parts = max_parts;
    if (parts * g.gj > 256) parts = 256 / g.gj > 0 ? 256 / g.gj : 1;
    if (parts < 1) parts = 1;
    g.parts = (int)parts;
    g.tpp = (int)((tiles_m + parts - 1) / parts);
    g.parts = (int)((tiles_m + g.tpp - 1) / g.tpp);
    return g;
}

template <int METRIC>
static int pf_launch_big(const uint16_t *a, int64_t lda, const uint16_t *b, int64_t ldb, const float *sqa, const float *sqb, int64_t na, int64_t nb,
                         int kp, const PfGeom &g, const unsigned *ctl_a, const unsigned *ctl_b, const PfArgs &fa, hipStream_t s)
{
    const int64_t tiles_m = (na + PB_BM - 1) / PB_BM, tiles_n = (nb + PB_BN - 1) / PB_BN;
    if (g.gi < 1 || g.gj < 1 || g.parts < 1 || (int64_t)g.tpp * g.parts < tiles_m || g.gj > g.tpp || kp % PB_BK)
        return fail(SE_ERR_INVALID, "pre-filter pass: geometry %d x %d, %d parts of %d tiles does not cover %lld gallery tiles", g.gi, g.gj, g.parts, g.tpp, (long long)tiles_m);
    if (tiles_m * tiles_n >= ((int64_t)1 << 31))
        return fail(SE_ERR_UNSUPPORTED, "se_retrieve_topk: %lld pre-filter tiles exceed the 32-bit tile counter", (long long)(tiles_m * tiles_n));
    if (lda * 2 * PB_BM >= ((int64_t)1 << 32) || ldb * 2 * PB_BN >= ((int64_t)1 << 32))
        return fail(SE_ERR_UNSUPPORTED, "se_retrieve_topk: pre-filter row pitch too large for 32-bit tile offsets");
    const size_t lds = (size_t)2 * PB_STAGE + (size_t)(2 * PB_BM + 3 * PB_BN) * sizeof(float);
    const int64_t grid = (int64_t)8 * g.gi * g.gj;
    auto kern = pf_big_kernel<METRIC>;
    SE_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PfArgs fb = fa;
    fb.prof = nullptr;
    static const bool profile = tuning_env("SE_PF_PROFILE") != nullptr;   // -DSE_TUNING build only: allocates, synchronises, prints
    if (profile) {
        SE_HIP_CHECK(hipMalloc((void **)&fb.prof, 12 * sizeof(unsigned long long)));
        SE_HIP_CHECK(hipMemsetAsync(fb.prof, 0, 12 * sizeof(unsigned long long), s));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(PB_THREADS), lds, s, a, (uint32_t)lda, b, (uint32_t)ldb, sqa, sqb, na, nb, kp / PB_BK,
                       (int)tiles_m, (int)tiles_n, g.parts, g.tpp, g.gi, g.gj, ctl_a, ctl_b, fb);
    SE_LAUNCH_CHECK();
    if (profile) {
        unsigned long long h[12];
        SE_HIP_CHECK(hipStreamSynchronize(s));
        SE_HIP_CHECK(hipMemcpy(h, fb.prof, sizeof(h), hipMemcpyDeviceToHost));
        SE_HIP_CHECK(hipFree(fb.prof));
        static const char *names[12] = {"loop-top", "request-next", "mfma", "-", "scan+reserve", "barrier+dump", "candidates", "-", "-", "-", "wait+barrier",
                                        "tile-end"};
        double tot = 0;
        for (int i = 0; i < 12; i++) tot += (double)h[i];
        fprintf(stderr, "[pf_big_kernel profile] metric=%d grid=%lld (8 x %d x %d) tiles=%lld x %lld chunks=%d parts=%d:", METRIC, (long long)grid,
                g.gi, g.gj, (long long)tiles_m, (long long)tiles_n, kp / PB_BK, g.parts);
        for (int i = 0; i < 12; i++) if (names[i][0] != '-') fprintf(stderr, " %s %.1f%%", names[i], 100.0 * (double)h[i] / tot);
        fprintf(stderr, "  (%.0f cycles per tile per workgroup)\n", tot / (double)(tiles_m * tiles_n));
    }
    return SE_OK;
}

template <int METRIC, int EPI>
static int pf_launch3(const uint16_t *a, int64_t lda, const uint16_t *b, int64_t ldb, const float *sqa, const float *sqb, int64_t na, int64_t nb,
                      int kp, const PfGeom &g, const unsigned *ctl_a, const unsigned *ctl_b, const PfArgs &fa, hipStream_t s)
{
    const int64_t tiles_m = (na + PF_BM - 1) / PF_BM, tiles_n = (nb + PF_BN - 1) / PF_BN;
    if (g.gi < 1 || g.gj < 1 || g.parts < 1 || (int64_t)g.tpp * g.parts < tiles_m || g.gj > g.tpp)
        return fail(SE_ERR_INVALID, "pre-filter pass: geometry %d x %d, %d parts of %d tiles does not cover %lld gallery tiles", g.gi, g.gj, g.parts, g.tpp, (long long)tiles_m);
    if (tiles_m * tiles_n >= ((int64_t)1 << 31))
        return fail(SE_ERR_UNSUPPORTED, "se_retrieve_topk: %lld pre-filter tiles exceed the 32-bit tile counter", (long long)(tiles_m * tiles_n));
    if (lda * 2 * PF_BM >= ((int64_t)1 << 32) || ldb * 2 * PF_BN >= ((int64_t)1 << 32))
        return fail(SE_ERR_UNSUPPORTED, "se_retrieve_topk: pre-filter row pitch too large for 32-bit tile offsets");
    const size_t lds = (size_t)(PF_BM + PF_BN) * PF_PITCH + PF_GAP + (size_t)(2 * PF_BM + 3 * PF_BN) * sizeof(float);
    const int64_t grid = (int64_t)8 * g.gi * g.gj;
    auto kern = pf_tile_kernel<METRIC, EPI>;
    SE_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PfArgs fb = fa;
    fb.prof = nullptr;
    static const bool profile = tuning_env("SE_PF_PROFILE") != nullptr;   // -DSE_TUNING build only: allocates, synchronises, prints
    if (profile) {
        SE_HIP_CHECK(hipMalloc((void **)&fb.prof, 12 * sizeof(unsigned long long)));
        SE_HIP_CHECK(hipMemsetAsync(fb.prof, 0, 12 * sizeof(unsigned long long), s));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(PF_THREADS), lds, s, a, (uint32_t)lda, b, (uint32_t)ldb, sqa, sqb, na, nb, kp / PF_BK,
                       (int)tiles_m, (int)tiles_n, g.parts, g.tpp, g.gi, g.gj, ctl_a, ctl_b, fb);
    SE_LAUNCH_CHECK();
    if (profile) {
        unsigned long long h[12];
        SE_HIP_CHECK(hipStreamSynchronize(s));
        SE_HIP_CHECK(hipMemcpy(h, fb.prof, sizeof(h), hipMemcpyDeviceToHost));
        SE_HIP_CHECK(hipFree(fb.prof));
        static const char *names[12] = {"loop-top", "prefetch-issue", "mfma", "wait-loads", "scan+reserve", "barrier+dump", "candidates", "-", "-", "-", "barrier",
                                        "stage+barrier"};
        double tot = 0;
        for (int i = 0; i < 12; i++) tot += (double)h[i];
        fprintf(stderr, "[pf_tile_kernel profile] metric=%d epi=%d grid=%lld (8 x %d x %d) tiles=%lld x %lld chunks=%d parts=%d:", METRIC, EPI, (long long)grid,
                g.gi, g.gj, (long long)tiles_m, (long long)tiles_n, kp / PF_BK, g.parts);
        for (int i = 0; i < 12; i++) if (names[i][0] != '-') fprintf(stderr, " %s %.1f%%", names[i], 100.0 * (double)h[i] / tot);
        fprintf(stderr, "  (%.0f cycles per tile per workgroup)\n", tot / (double)(tiles_m * tiles_n));
    }
    return SE_OK;
}

template <int METRIC>
static int pf_launch2(int epi, const uint16_t *a, int64_t lda, const uint16_t *b, int64_t ldb, const float *sqa, const float *sqb,
                      int64_t na, int64_t nb, int kp, const PfGeom &g, const unsigned *ca, const unsigned *cb, const PfArgs &fa, hipStream_t s)
{
    if (epi == PF_GROUPMIN) return pf_launch3<METRIC, PF_GROUPMIN>(a, lda, b, ldb, sqa, sqb, na, nb, kp, g, ca, cb, fa, s);
    if (epi == PF_FILTER) return pf_launch3<METRIC, PF_FILTER>(a, lda, b, ldb, sqa, sqb, na, nb, kp, g, ca, cb, fa, s);
#ifdef SE_TUNING
    if (epi == PF_STORE) return pf_launch3<METRIC, PF_STORE>(a, lda, b, ldb, sqa, sqb, na, nb, kp, g, ca, cb, fa, s);
#endif
    return fail(SE_ERR_UNSUPPORTED, "pre-filter pass %d", epi);
}

// ---- host interface (driver: topk.hip) -----------------------------------------------------------------------------------------
int pf_padded_dim(int64_t d) { return (int)((d + PF_BK - 1) / PF_BK * PF_BK); }

int pf_convert(const float *x, int64_t ldx, int64_t n, int64_t d, uint16_t *out, float *nrm, float *res, unsigned *ctl, hipStream_t s)
{
    const int kp = pf_padded_dim(d);
    int64_t grid = (n + 3) / 4;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(pf_maxabs_kernel, dim3((unsigned)grid), dim3(256), 0, s, x, ldx, n, (int)d, ctl);
    SE_LAUNCH_CHECK();
    hipLaunchKernelGGL(pf_convert_kernel, dim3((unsigned)grid), dim3(256), 0, s, x, ldx, n, (int)d, kp, out, nrm, res, ctl);
    SE_LAUNCH_CHECK();
    return SE_OK;
}

int pf_pass(int epi, const PfGeom *geom, int metric, const uint16_t *gallery, int64_t lda, const uint16_t *queries, int64_t ldq, const float *sqg,
            const float *sqq, int64_t n_a, int64_t n_q, int kp, const unsigned *ctl_g, const unsigned *ctl_q, const PfPassArgs &pa, hipStream_t s)
{
    PfArgs fa;
    fa.gm = pa.gm; fa.gm_ld = pa.gm_ld; fa.thr = pa.thr; fa.rowcnt = pa.rowcnt; fa.lists = pa.lists; fa.cap = pa.cap;
    fa.sqa_stride = pa.sqa_stride; fa.out = pa.out; fa.ldo = pa.ldo; fa.prof = nullptr; fa.spill = pa.spill; fa.spill_lists = pa.spill_lists; fa.spill_cnt = pa.spill_cnt;
    {
        const int in_last = pa.d_valid > 0 ? pa.d_valid - (kp / PF_BK - 1) * PF_BK : PF_BK;     // columns with data in the last chunk
        fa.last_steps = (in_last <= 0 || in_last >= PF_BK) ? PF_BK / 16 : (in_last + 15) / 16;
    }
    const PfGeom g = geom ? *geom : pf_geometry(n_a, n_q, 1, 0);
    if (g.big) {
        if (epi != PF_FILTER) return fail(SE_ERR_INVALID, "pre-filter pass: the 256 x 256 kernel only filters");
        if (metric == SE_METRIC_COSINE) return pf_launch_big<SE_METRIC_COSINE>(gallery, lda, queries, ldq, sqg, sqq, n_a, n_q, kp, g, ctl_g, ctl_q, fa, s);
        if (metric == SE_METRIC_EUCLID) return pf_launch_big<SE_METRIC_EUCLID>(gallery, lda, queries, ldq, sqg, sqq, n_a, n_q, kp, g, ctl_g, ctl_q, fa, s);
        return fail(SE_ERR_UNSUPPORTED, "pre-filter pass: metric %d", metric);
    }
    if (metric == SE_METRIC_COSINE) return pf_launch2<SE_METRIC_COSINE>(epi, gallery, lda, queries, ldq, sqg, sqq, n_a, n_q, kp, g, ctl_g, ctl_q, fa, s);
    if (metric == SE_METRIC_EUCLID) return pf_launch2<SE_METRIC_EUCLID>(epi, gallery, lda, queries, ldq, sqg, sqq, n_a, n_q, kp, g, ctl_g, ctl_q, fa, s);
    return fail(SE_ERR_UNSUPPORTED, "pre-filter pass: metric %d", metric);
}

}  // namespace se
