// se_common.h -- shared helpers for the gfx950 kernels behind include/sehip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/sehip.h"

namespace se {

// thread-local last-error text (se_last_error)
char *err_buf();
int fail(int code, const char *fmt, ...);

// Phase timing (se_phase_timing / se_phase_timing_read, include/sehip.h): when switched on, multi-kernel entry points record a HIP
// event on their stream behind each phase.  Off (the default): one relaxed atomic load per mark.
void phase_mark(const char *name, hipStream_t s);
bool phase_timing_on();
void phase_note_counters(const unsigned *dev_counters, long long rows, hipStream_t s);   // se_retrieve_topk: its 4 statistics words, copied on stream s into a library-owned pinned buffer

#define SE_HIP_CHECK(expr)                                                                     \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return se::fail(SE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                               \
    } while (0)

#define SE_LAUNCH_CHECK()                                                                      \
    do {                                                                                       \
        hipError_t e__ = hipGetLastError();                                                    \
        if (e__ != hipSuccess)                                                                 \
            return se::fail(SE_ERR_HIP, "kernel launch failed: %s (%s:%d)",                    \
                            hipGetErrorString(e__), __FILE__, __LINE__);                       \
    } while (0)

constexpr int WAVE = 64;

// Tuning / test switches (environment variables that pin a kernel variant, print phase profiles or make a kernel skip work)
// exist only in the -DSE_TUNING build (sehip/libsehip_tuning.so, loaded by tests / tools through SEHIP_LIB): the product
// library ignores them all, so no environment variable can change what an entry point computes.  Product switches that stay:
// SE_RANK_SAFE=1 (guaranteed-order ranking kernel) and SE_RANK_VERBOSE=1 (says which ranking kernel the probe selected).
#ifdef SE_TUNING
inline const char *tuning_env(const char *name) { return getenv(name); }
constexpr bool kTuning = true;
#else
inline const char *tuning_env(const char *) { return nullptr; }
constexpr bool kTuning = false;
#endif

// Work-group barrier of every kernel in this library: drain this wave's LDS queue, THEN barrier.
// __syncthreads() alone is not enough on gfx950 with this compiler: for a barrier at a loop header the waitcnt pass can leave
// the back edge without an s_waitcnt lgkmcnt(0) between the previous stage's ds_write and the s_barrier (seen in the ISA of the
// LDS bitonic sort of topk_rows_kernel).  The stores are then still queued when the wave signals the barrier; the next stage's
// reader on another SIMD normally loses that race, but not when a second kernel's waves share the SIMD's LDS queue --
// se_topk_rows returned the right k entries in the wrong order in ~1 of 5 calls with another stream busy.
__device__ __forceinline__ void wg_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

// Order-preserving float32 -> uint32 key of the canonical ranking order:
// ascending value, -0.0 == +0.0, every NaN maps to 0xFFFFFFFF (sorted last).
__device__ __forceinline__ uint32_t canon_key(float f)
{
    uint32_t u = __float_as_uint(f);
    if (f != f) return 0xFFFFFFFFu;
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even float32 -> bf16 (NaN preserved)
__device__ __forceinline__ uint16_t f32_to_bf16(float f)
{
    uint32_t u = __float_as_uint(f);
    if (f != f) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// K-block list of the canonical dot product: the FMA chain restarts per block, block sums are added in order
constexpr int SE_MAX_KB = 16;
struct KBlocks {
    int n;
    int len[SE_MAX_KB];
};
// host: validate a caller's K-block list into `kbs`; *multi = more than one block.  SE_OK or an error code (text set).
int make_kblocks(const char *who, const int32_t *kblocks, int nkb, int64_t d, KBlocks &kbs, bool &multi);

// ---- fused distance + top-k (se_retrieve_topk): the two passes that run on the tile loop of pdist_mfma.hip ----
// Both take the GALLERY as the row operand and the QUERIES as the column operand of the MFMA tiles, so that a lane owns ONE
// query per 32-column block and sees 16 gallery rows of it per tile.
constexpr int EPI_STORE = 0;      // se_pairwise_dist: the tile goes to the distance matrix
constexpr int EPI_GROUPMIN = 1;   // sample pass: min over each lane's 16 values -> gm[query, group]
constexpr int EPI_FILTER = 2;     // main pass: values <= tau[query] are appended to the query's candidate list
constexpr int FUSED_GROUP = 16;   // values per group minimum (one lane, one 32 x 32 accumulator block)
struct FusedArgs {
    float *gm;             // EPI_GROUPMIN: [queries, gm_ld] group minima; group = 8 * (tile row index) + 2 * wave row + lane half
    int64_t gm_ld;
    const float *tau;      // EPI_FILTER: [queries] thresholds
    unsigned *rowcnt;      // EPI_FILTER: [queries] candidates appended so far (may exceed cap: the row is then redone)
    uint2 *lists;          // EPI_FILTER: [queries, cap] (float bits of the distance, gallery row)
    int64_t cap;
    int64_t sqa_stride;    // Euclidean epilogue: |a|^2 of row r is sqa[r * sqa_stride] (the sample pass strides through the gallery)
};
// host: one pass over (gallery rows [n_a, d] with pitch lda) x (queries [n_q, d]); metric SE_METRIC_COSINE / SE_METRIC_EUCLID
int launch_fused_pass(int epi, const float *gallery, int64_t lda, const float *queries, int64_t ldq, const float *sqg, const float *sqq,
                      int64_t n_a, int64_t n_q, int64_t d, int metric, const KBlocks &kbs, bool multi, const FusedArgs &fa, hipStream_t s);
int64_t pdist_max_ld();

// ---- bf16 pre-filter of the fused distance + top-k (prefilter.hip; driver and exact refinement: topk.hip) ----
constexpr int PF_EPI_GROUPMIN = 1, PF_EPI_FILTER = 2, PF_EPI_STORE = 3;
struct PfPassArgs {
    float *gm; int64_t gm_ld;          // sample pass: [queries, gm_ld] group minima of d~ (d~: the half-precision distance)
    const float *thr;                  // filter pass: [queries] thresholds (NaN: only NaN values pass)
    unsigned *rowcnt; uint2 *lists; int64_t cap;     // cap: entries per (query, sub-list)
    int spill; uint2 *spill_lists; unsigned *spill_cnt;   // filter pass: sub-list slots of every query's shared spill region, [queries][spill * cap]; fill counters [queries]
    int64_t sqa_stride;
    float *out; int64_t ldo;           // PF_EPI_STORE (tuning build): d~ matrix [gallery rows, queries]
    int d_valid;                       // columns that are not zero padding (0: all kp): the 128 x 128 kernel skips the all-zero k = 16 steps of its last chunk
};
int pf_padded_dim(int64_t d);
// fp32 rows -> scaled fp16 rows [n, pf_padded_dim(d)] + per-row norm of the image / of the rounding residual (upper bounds; NaN for
// irregular rows, whose image is all NaN) + ctl[0..4] = max image norm bits, max residual norm bits, irregular rows, largest regular
// magnitude bits, scale exponent (caller zeroes the 8-word ctl block)
int pf_convert(const float *x, int64_t ldx, int64_t n, int64_t d, uint16_t *out, float *nrm, float *res, unsigned *ctl, hipStream_t s);
// ctl_g / ctl_q: the ctl blocks pf_convert filled for the two operand matrices (word 4 = the image's scale exponent)
// geometry of a pass (prefilter.hip): the workgroups resident on an XCD form a gi x gj grid and take (gi query tiles) x (one of `parts`
// contiguous gallery ranges of `tpp` tiles) at a time, workgroup (i, j) walking the range's tiles j, j + gj, ...  A query's candidates of
// (range p, sequence j) go to sub-list p * gj + j:  lists[(query * nsub + sub) * cap ...], their number to rowcnt[query * nsub + sub],
// nsub = parts * gj.  want_parts = 0: as many ranges as keep every XCD busy for >= ~4 rounds.
struct PfGeom { int gi, gj, parts, tpp, big; };      // big: the 256 x 256 filter kernel (long rows) instead of the 128 x 128 one
PfGeom pf_geometry(int64_t n_a, int64_t n_q, int want_parts, int kp);    // kp: padded columns of the FILTER pass this is for; 0: sample pass
// filter pass with a spill region: counters [queries][nsub + spill] -> per-sub-list counts as the refinement reads them (prefilter.hip)
int pf_spill_counts(unsigned *rowcnt, const unsigned *spill_cnt, int64_t queries, int nsub, int spill, int64_t cap, hipStream_t s);
int pf_pass(int epi, const PfGeom *geom, int metric, const uint16_t *gallery, int64_t lda, const uint16_t *queries, int64_t ldq, const float *sqg,
            const float *sqq, int64_t n_a, int64_t n_q, int kp, const unsigned *ctl_g, const unsigned *ctl_q, const PfPassArgs &pa, hipStream_t s);

}  // namespace se
