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
#include "se_common.h"

namespace se {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float pf_f32x16 __attribute__((ext_vector_type(16)));

constexpr int PF_BM = 128, PF_BN = 128, PF_BK = 128;       // tile, K-chunk (fp16 elements)
constexpr int PF_THREADS = 256;
constexpr int PF_ROWB = PF_BK * 2;                          // bytes of one operand row of a chunk (256)
constexpr int PF_PITCH = PF_ROWB + 16;                      // LDS row pitch in bytes: 68 dwords -> conflict-free ds_read_b128 over 16 rows
constexpr int PF_NLOAD = PF_BM * PF_ROWB / 16 / PF_THREADS; // 16-byte pieces per operand per thread (8)
constexpr int PF_PPR = PF_ROWB / 16;                        // pieces per row (16)
constexpr int PF_WGS_PER_CU = 2;
constexpr int PF_GROUP_M = 16;

constexpr int PF_GROUPMIN = PF_EPI_GROUPMIN, PF_FILTER = PF_EPI_FILTER, PF_STORE = PF_EPI_STORE;

struct PfArgs {
    float *gm; int64_t gm_ld;          // PF_GROUPMIN
    const float *thr;                  // PF_FILTER: [queries] thresholds (distance units); NaN = nothing but NaN passes
    unsigned *rowcnt; uint2 *lists; int64_t cap;
    int64_t sqa_stride;                // Euclidean epilogue: |a|^2 of gallery row r is sqa[r * sqa_stride]
    float *out; int64_t ldo;           // PF_STORE
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
    __syncthreads();
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
    return e < -120 ? -120 : (e > 120 ? 120 : e);
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
                const float hv = (float)hh * isc;                                 // the image in the operand's own units (exact: power-of-two scale)
                const float rv = v[i] - hv;
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
        float nn = sqrtf(sn) * 1.002f + 1e-37f, rr = sqrtf(sr) * 1.002f + 1e-37f;
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
    __syncthreads();
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

// ---- tile loop --------------------------------------------------------------------------------------------------------------------
template <int METRIC>
__device__ __forceinline__ float pf_finish(float v, float sa, float sb)
{
    if (METRIC == SE_METRIC_COSINE) return -v;
    if (METRIC == SE_METRIC_EUCLID) return (sa + sb) - 2.0f * v;
    return v;
}

// linear tile index -> tile origin (same walks as pdist_mfma.hip: "16 tile-rows deep" grouped order / upper triangle row-major)
template <bool SYM>
__device__ __forceinline__ void pf_tile_coords(uint32_t t, int tiles_m, int tiles_n, int &tm_out, int &tn_out)
{
    if (SYM) {
        const double T = (double)tiles_n;
        int32_t tm = (int32_t)(((2.0 * T + 1.0) - sqrt((2.0 * T + 1.0) * (2.0 * T + 1.0) - 8.0 * (double)t)) * 0.5);
        if (tm < 0) tm = 0;
        if (tm > tiles_m - 1) tm = tiles_m - 1;
        while (tm > 0 && (uint32_t)tm * (uint32_t)tiles_n - (uint32_t)tm * (uint32_t)(tm - 1) / 2u > t) tm--;
        while ((uint32_t)(tm + 1) * (uint32_t)tiles_n - (uint32_t)(tm + 1) * (uint32_t)tm / 2u <= t) tm++;
        const uint32_t off = (uint32_t)tm * (uint32_t)tiles_n - (uint32_t)tm * (uint32_t)(tm - 1) / 2u;
        tm_out = tm;
        tn_out = (int)((uint32_t)tm + (t - off));
        return;
    }
    const uint32_t per_group = (uint32_t)PF_GROUP_M * (uint32_t)tiles_n;
    const uint32_t group = t / per_group, in_g = t - group * per_group;
    const uint32_t first_m = group * PF_GROUP_M;
    const uint32_t gsz = ((uint32_t)tiles_m - first_m < (uint32_t)PF_GROUP_M) ? ((uint32_t)tiles_m - first_m) : (uint32_t)PF_GROUP_M;
    const uint32_t col_t = in_g / gsz;
    tm_out = (int)(first_m + (in_g - col_t * gsz));
    tn_out = (int)col_t;
}

// global -> registers: chunk [k0, k0 + 128) of rows [row0, row0 + 128) of an fp16 matrix with pitch `ld` elements (multiple of 128 columns,
// 16-byte aligned rows).  Rows beyond nrows are clamped (read twice, ignored by the epilogues): no masking anywhere in the loop.
__device__ __forceinline__ void pf_load(uint4 (&v)[PF_NLOAD], const uint16_t *__restrict__ src, uint32_t ld, int64_t row0, int64_t nrows, int k0)
{
    const int tid = threadIdx.x;
    const char *base = (const char *)(src + row0 * (int64_t)ld);       // uniform
    const int rows_here = (int)((nrows - row0 < PF_BM) ? (nrows - row0) : PF_BM);
#pragma unroll
    for (int i = 0; i < PF_NLOAD; i++) {
        const int p = tid + i * PF_THREADS;
        const int r = p / PF_PPR, c = p % PF_PPR;
        const int rc = r < rows_here ? r : rows_here - 1;
        v[i] = *(const uint4 *)(base + ((uint32_t)rc * ld * 2u + (uint32_t)k0 * 2u + (uint32_t)c * 16u));
    }
}

__device__ __forceinline__ void pf_stage(char *lds, const uint4 (&v)[PF_NLOAD])
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < PF_NLOAD; i++) {
        const int p = tid + i * PF_THREADS;
        *(uint4 *)(lds + (p / PF_PPR) * PF_PITCH + (p % PF_PPR) * 16) = v[i];
    }
}

template <int METRIC, bool SYM, int EPI>
__global__ __launch_bounds__(PF_THREADS, PF_WGS_PER_CU) void pf_tile_kernel(
    const uint16_t *__restrict__ A, uint32_t lda, const uint16_t *__restrict__ B, uint32_t ldb, const float *__restrict__ sqa,
    const float *__restrict__ sqb, int64_t NA, int64_t NB, int nchunks, int tiles_m, int tiles_n, int64_t ntiles, const unsigned *__restrict__ ctl_a,
    const unsigned *__restrict__ ctl_b, PfArgs fa)
{
    static_assert(EPI != PF_GROUPMIN || !SYM, "the sample pass walks the general tile order");
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];
    char *sA = pf_smem, *sB = pf_smem + PF_BM * PF_PITCH;
    // per-tile side arrays (filled together with the tile's first chunk, i.e. between the two barriers in front of its first MFMA phase,
    // and read by its epilogue): thresholds of the tile's query columns / (all-pairs) query rows, |.|^2 of its gallery rows / query columns
    float *tThrCol = (float *)(pf_smem + (PF_BM + PF_BN) * PF_PITCH), *tThrRow = tThrCol + PF_BN, *tSqRow = tThrRow + PF_BM, *tSqCol = tSqRow + PF_BM;
#define PF_SIDE(M0, N0)                                                                                              \
    if (EPI != PF_STORE || METRIC == SE_METRIC_EUCLID) {                                                             \
        const int t_ = threadIdx.x;                                                                                  \
        if (t_ < PF_BN) {                                                                                            \
            const int64_t qc_ = (N0) + t_;                                                                           \
            const bool ok_ = qc_ < NB;                                                                               \
            if (EPI == PF_FILTER) tThrCol[t_] = ok_ ? fa.thr[qc_] : -__builtin_inff();                               \
            if (METRIC == SE_METRIC_EUCLID) tSqCol[t_] = sqb[ok_ ? qc_ : NB - 1];                                    \
        } else {                                                                                                     \
            const int64_t gr_ = (M0) + (t_ - PF_BN);                                                                 \
            const bool ok_ = gr_ < NA;                                                                               \
            if (EPI == PF_FILTER && SYM) tThrRow[t_ - PF_BN] = ok_ ? fa.thr[gr_] : -__builtin_inff();                \
            if (METRIC == SE_METRIC_EUCLID) tSqRow[t_ - PF_BN] = sqa[(ok_ ? gr_ : NA - 1) * fa.sqa_stride];          \
        }                                                                                                            \
    }

    // ---- this workgroup's tile list: XCD-contiguous band, round-robin inside the XCD ----
    const int64_t b = blockIdx.x, G = gridDim.x;
    const int64_t xcd = b & 7, qq = ntiles >> 3, rr = ntiles & 7;
    const int64_t band_beg = (xcd < rr) ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq;
    const int64_t band_len = qq + (xcd < rr ? 1 : 0);
    const int64_t wg_in_xcd = b >> 3, wgs_per_xcd = (G + 7 - xcd) >> 3;
    const int64_t my_tiles = (band_len > wg_in_xcd) ? (band_len - wg_in_xcd + wgs_per_xcd - 1) / wgs_per_xcd : 0;
    if (my_tiles == 0) return;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;        // 2 x 2 waves, 64 x 64 outputs each
    const float unscale = ldexpf(1.0f, -((int)ctl_a[4] + (int)ctl_b[4]));     // the images carry 2^ea, 2^eb: exact to undo
    const int col = lane & 31, hi = lane >> 5;

    pf_f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mi][j][r] = 0.f;

    const char *pa = sA + (wm * 64 + col) * PF_PITCH + hi * 16;     // block mi: + mi * 32 rows; k16 step s: + s * 32 bytes
    const char *pb = sB + (wn * 64 + col) * PF_PITCH + hi * 16;

    uint4 ra[PF_NLOAD], rb[PF_NLOAD];
    int tm, tn;
    pf_tile_coords<SYM>((uint32_t)(band_beg + wg_in_xcd), tiles_m, tiles_n, tm, tn);
    int64_t m0 = (int64_t)tm * PF_BM, n0 = (int64_t)tn * PF_BN;
    pf_load(ra, A, lda, m0, NA, 0);
    pf_load(rb, B, ldb, n0, NB, 0);

    const int64_t total = my_tiles * nchunks;
    int c = 0;
    uint32_t tile_i = 0;
    int64_t cur_m0 = m0, cur_n0 = n0;
    pf_stage(sA, ra);
    pf_stage(sB, rb);
    PF_SIDE(m0, n0)
    __syncthreads();
#pragma unroll 1
    for (int64_t it = 0; it < total; it++) {
        // ---- request the next chunk (same tile or first chunk of the next tile) ----
        const bool last_chunk = (c + 1 == nchunks);
        const bool have_next = it + 1 < total;
        if (have_next) {
            const int nc = last_chunk ? 0 : c + 1;
            if (nc == 0) {
                pf_tile_coords<SYM>((uint32_t)(band_beg + wg_in_xcd) + (tile_i + 1u) * (uint32_t)wgs_per_xcd, tiles_m, tiles_n, tm, tn);
                m0 = (int64_t)tm * PF_BM; n0 = (int64_t)tn * PF_BN;
            }
            pf_load(ra, A, lda, m0, NA, nc * PF_BK);
            pf_load(rb, B, ldb, n0, NB, nc * PF_BK);
        }
        // ---- MFMA over the chunk in LDS: 8 steps of k = 16 ----
#pragma unroll
        for (int s = 0; s < PF_BK / 16; s++) {
            if (s == PF_BK / 32) asm volatile("" ::: "memory");       // two groups of 4 steps: all 32 operand reads hoisted at once cost 128 registers
            f16x8 a0 = __builtin_bit_cast(f16x8, *(const uint4 *)(pa + s * 32));
            f16x8 a1 = __builtin_bit_cast(f16x8, *(const uint4 *)(pa + 32 * PF_PITCH + s * 32));
            f16x8 b0 = __builtin_bit_cast(f16x8, *(const uint4 *)(pb + s * 32));
            f16x8 b1 = __builtin_bit_cast(f16x8, *(const uint4 *)(pb + 32 * PF_PITCH + s * 32));
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
        }
        // the next chunk's operands are waited for HERE (value barriers: no use of a loaded register in front of the MFMA phase)
#pragma unroll
        for (int i = 0; i < PF_NLOAD; i++) {
            asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));
            asm volatile("" : "+v"(rb[i].x), "+v"(rb[i].y), "+v"(rb[i].z), "+v"(rb[i].w));
        }

        if (last_chunk) {
            // ---- tile finished.  acc[mi][j][r]: gallery row  cur_m0 + wm*64 + mi*32 + (r&3) + 8*(r>>2) + 4*hi,
            //                                      query       cur_n0 + wn*64 + j*32 + col ----
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
                        if (qok) fa.gm[qg * fa.gm_ld + (cur_m0 / PF_BM) * 8 + (wm * 2 + mi) * 2 + hi] = m;
                    }
                }
            } else {
                // ---- PF_FILTER, orientation 1: lanes = queries (tile columns), registers = gallery rows ----
                int64_t qgj[2];
                float thrj[2], sbqj[2];
                unsigned cntj[2], slotj[2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int lc = wn * 64 + j * 32 + col;
                    const bool qok = lc < cols_here;
                    qgj[j] = cur_n0 + (qok ? lc : cols_here - 1);
                    thrj[j] = tThrCol[lc];
                    sbqj[j] = METRIC == SE_METRIC_EUCLID ? tSqCol[lc] : 0.f;
                }
#define PF_PASS1(V, J, LR) ((((V) <= thrj[J]) || ((V) != (V))) && (full_rows || (LR) < rows_here) && (wn * 64 + (J) * 32 + col < cols_here))
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float sbq = sbqj[j];
                    unsigned cnt = 0;
#pragma unroll
                    for (int mi = 0; mi < 2; mi++)
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int lr = lr0 + mi * 32 + (r & 3) + 8 * (r >> 2);
                            const float v = PF_VAL(mi, j, r);
                            cnt += PF_PASS1(v, j, lr) ? 1u : 0u;
                        }
                    cntj[j] = cnt;
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    slotj[j] = 0;
                    if (cntj[j]) slotj[j] = atomicAdd(&fa.rowcnt[qgj[j]], cntj[j]);
                }
                asm volatile("" : "+v"(slotj[0]), "+v"(slotj[1]));     // ONE wait for both reservations
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if (cntj[j]) {
                        const float sbq = sbqj[j];
                        unsigned slot = slotj[j];
                        uint2 *lst = fa.lists + qgj[j] * fa.cap;
#pragma unroll
                        for (int mi = 0; mi < 2; mi++)
#pragma unroll
                            for (int r = 0; r < 16; r++) {
                                const int lr = lr0 + mi * 32 + (r & 3) + 8 * (r >> 2);
                                const float v = PF_VAL(mi, j, r);
                                if (PF_PASS1(v, j, lr)) {
                                    if (slot < (unsigned)fa.cap) lst[slot] = make_uint2(__float_as_uint(v), (uint32_t)(cur_m0 + lr));
                                    slot++;
                                }
                            }
                    }
                }
#undef PF_PASS1
                if (SYM && cur_m0 != cur_n0) {
                    // ---- orientation 2 (all-pairs call, off-diagonal tile): queries = tile ROWS, gallery items = tile columns.  Per
                    //      register r a 64-lane ballot holds the verdicts of TWO query rows (lanes 0-31: row ..+0, lanes 32-63: row ..+4)
                    //      x 32 gallery columns.  Row R of the wave's 64 rows is owned by lane R for the slot reservation:
                    //      R = mi*32 + (r&3) + 8*(r>>2) + 4*h.  Counts -> one returning atomic per lane, ONE wait, then the stores. ----
                    const int gc0 = wn * 64 + col;                              // this lane's gallery column of block j: gc0 + 32 j
                    unsigned mycnt = 0;
                    // thresholds / norms of this lane's 2 x 16 query rows come out of the tile's side arrays one 32-row block at a time
                    // (16 + 16 live registers; a compiler fence between the blocks keeps the second block's reads behind the first's use)
                    float sbc[2];
#pragma unroll
                    for (int j = 0; j < 2; j++) sbc[j] = METRIC == SE_METRIC_EUCLID ? tSqCol[gc0 + 32 * j] : 0.f;
                    const bool c0ok = gc0 < cols_here, c1ok = gc0 + 32 < cols_here;
#define PF_ROWS2(MI_)                                                                                                        \
    float t2[16], s2[16];                                                                                                    \
    _Pragma("unroll") for (int r = 0; r < 16; r++) {                                                                         \
        t2[r] = tThrRow[lr0 + (MI_) * 32 + (r & 3) + 8 * (r >> 2)];                                                          \
        s2[r] = METRIC == SE_METRIC_EUCLID ? tSqRow[lr0 + (MI_) * 32 + (r & 3) + 8 * (r >> 2)] : 0.f;                        \
    }
#define PF_VAL2(MI_, J, R) pf_finish<METRIC>(acc[MI_][J][R] * unscale, s2[R], sbc[J])
#define PF_PASS2(V, R, OK) ((((V) <= t2[R]) || ((V) != (V))) && (OK))
#pragma unroll
                    for (int mi = 0; mi < 2; mi++) {
                        asm volatile("" ::: "memory");
                        PF_ROWS2(mi)
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const float v0 = PF_VAL2(mi, 0, r), v1 = PF_VAL2(mi, 1, r);
                            const uint64_t b0 = __ballot(PF_PASS2(v0, r, c0ok)), b1 = __ballot(PF_PASS2(v1, r, c1ok));
                            const unsigned clo = (unsigned)__popc((uint32_t)b0) + (unsigned)__popc((uint32_t)b1);
                            const unsigned chi = (unsigned)__popc((uint32_t)(b0 >> 32)) + (unsigned)__popc((uint32_t)(b1 >> 32));
                            const int R0 = mi * 32 + (r & 3) + 8 * (r >> 2);
                            mycnt = lane == R0 ? clo : mycnt;
                            mycnt = lane == R0 + 4 ? chi : mycnt;
                        }
                    }
                    // lane R reserves for query row  cur_m0 + wm*64 + R
                    const int myrow = wm * 64 + lane;
                    unsigned myslot = 0;
                    if (mycnt) myslot = atomicAdd(&fa.rowcnt[cur_m0 + myrow], mycnt);
                    asm volatile("" : "+v"(myslot));
#pragma unroll
                    for (int mi = 0; mi < 2; mi++) {
                        asm volatile("" ::: "memory");
                        PF_ROWS2(mi)
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const float v0 = PF_VAL2(mi, 0, r), v1 = PF_VAL2(mi, 1, r);
                            const bool p0 = PF_PASS2(v0, r, c0ok), p1 = PF_PASS2(v1, r, c1ok);
                            const uint64_t b0 = __ballot(p0), b1 = __ballot(p1);
                            if ((b0 | b1) == 0ull) continue;                                     // uniform: most rows of most tiles
                            const int R0 = mi * 32 + (r & 3) + 8 * (r >> 2);
                            const unsigned base_lo = (unsigned)__builtin_amdgcn_readlane((int)myslot, R0);
                            const unsigned base_hi = (unsigned)__builtin_amdgcn_readlane((int)myslot, R0 + 4);
                            // position inside this row's reservation: block 0's passing lanes (of my half) first, then block 1's
                            const uint32_t h0 = hi ? (uint32_t)(b0 >> 32) : (uint32_t)b0, h1 = hi ? (uint32_t)(b1 >> 32) : (uint32_t)b1;
                            const uint32_t below = (1u << col) - 1u;
                            const unsigned base = hi ? base_hi : base_lo;
                            const int64_t qrow = cur_m0 + wm * 64 + R0 + 4 * hi;
                            uint2 *lst = fa.lists + qrow * fa.cap;
                            if (p0) {
                                const unsigned slot = base + (unsigned)__popc(h0 & below);
                                if (slot < (unsigned)fa.cap) lst[slot] = make_uint2(__float_as_uint(v0), (uint32_t)(cur_n0 + gc0));
                            }
                            if (p1) {
                                const unsigned slot = base + (unsigned)__popc(h0) + (unsigned)__popc(h1 & below);
                                if (slot < (unsigned)fa.cap) lst[slot] = make_uint2(__float_as_uint(v1), (uint32_t)(cur_n0 + gc0 + 32));
                            }
                        }
                    }
#undef PF_ROWS2
#undef PF_VAL2
#undef PF_PASS2
                }
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
        __syncthreads();   // every wave has finished reading this chunk out of LDS
        if (have_next) {
            cur_m0 = m0; cur_n0 = n0;
            pf_stage(sA, ra);
            pf_stage(sB, rb);
            if (last_chunk) { PF_SIDE(m0, n0) }
        }
        __syncthreads();
        c = last_chunk ? 0 : c + 1;
        tile_i += last_chunk ? 1u : 0u;
    }
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

template <int METRIC, bool SYM, int EPI>
static int pf_launch3(const uint16_t *a, int64_t lda, const uint16_t *b, int64_t ldb, const float *sqa, const float *sqb, int64_t na, int64_t nb,
                      int kp, const unsigned *ctl_a, const unsigned *ctl_b, const PfArgs &fa, hipStream_t s)
{
    const int tiles_m = (int)((na + PF_BM - 1) / PF_BM), tiles_n = (int)((nb + PF_BN - 1) / PF_BN);
    const int64_t ntiles = SYM ? ((int64_t)tiles_n * (tiles_n + 1) / 2) : ((int64_t)tiles_m * tiles_n);
    if (ntiles >= ((int64_t)1 << 31) || (int64_t)PF_GROUP_M * tiles_n >= ((int64_t)1 << 31) || (SYM && tiles_n > 65535))
        return fail(SE_ERR_UNSUPPORTED, "se_retrieve_topk: %lld pre-filter tiles exceed the 32-bit tile counter", (long long)ntiles);
    if (lda * 2 * PF_BM >= ((int64_t)1 << 32) || ldb * 2 * PF_BN >= ((int64_t)1 << 32))
        return fail(SE_ERR_UNSUPPORTED, "se_retrieve_topk: pre-filter row pitch too large for 32-bit tile offsets");
    const size_t lds = (size_t)(PF_BM + PF_BN) * PF_PITCH + (size_t)(2 * PF_BM + 2 * PF_BN) * sizeof(float);
    int64_t grid = (int64_t)pf_num_cus() * PF_WGS_PER_CU;
    grid = grid / 8 * 8;
    if (grid > ntiles) grid = ntiles;
    if (grid < 1) grid = 1;
    auto kern = pf_tile_kernel<METRIC, SYM, EPI>;
    SE_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(PF_THREADS), lds, s, a, (uint32_t)lda, b, (uint32_t)ldb, sqa, sqb, na, nb, kp / PF_BK,
                       tiles_m, tiles_n, ntiles, ctl_a, ctl_b, fa);
    SE_LAUNCH_CHECK();
    return SE_OK;
}
#undef PF_SIDE

template <int METRIC>
static int pf_launch2(int epi, bool sym, const uint16_t *a, int64_t lda, const uint16_t *b, int64_t ldb, const float *sqa, const float *sqb,
                      int64_t na, int64_t nb, int kp, const unsigned *ca, const unsigned *cb, const PfArgs &fa, hipStream_t s)
{
    if (epi == PF_GROUPMIN) return pf_launch3<METRIC, false, PF_GROUPMIN>(a, lda, b, ldb, sqa, sqb, na, nb, kp, ca, cb, fa, s);
    if (epi == PF_FILTER) return sym ? pf_launch3<METRIC, true, PF_FILTER>(a, lda, b, ldb, sqa, sqb, na, nb, kp, ca, cb, fa, s)
                                      : pf_launch3<METRIC, false, PF_FILTER>(a, lda, b, ldb, sqa, sqb, na, nb, kp, ca, cb, fa, s);
#ifdef SE_TUNING
    if (epi == PF_STORE) return pf_launch3<METRIC, false, PF_STORE>(a, lda, b, ldb, sqa, sqb, na, nb, kp, ca, cb, fa, s);
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

int pf_pass(int epi, bool sym, int metric, const uint16_t *gallery, int64_t lda, const uint16_t *queries, int64_t ldq, const float *sqg,
            const float *sqq, int64_t n_a, int64_t n_q, int kp, const unsigned *ctl_g, const unsigned *ctl_q, const PfPassArgs &pa, hipStream_t s)
{
    PfArgs fa;
    fa.gm = pa.gm; fa.gm_ld = pa.gm_ld; fa.thr = pa.thr; fa.rowcnt = pa.rowcnt; fa.lists = pa.lists; fa.cap = pa.cap;
    fa.sqa_stride = pa.sqa_stride; fa.out = pa.out; fa.ldo = pa.ldo;
    if (metric == SE_METRIC_COSINE) return pf_launch2<SE_METRIC_COSINE>(epi, sym, gallery, lda, queries, ldq, sqg, sqq, n_a, n_q, kp, ctl_g, ctl_q, fa, s);
    if (metric == SE_METRIC_EUCLID) return pf_launch2<SE_METRIC_EUCLID>(epi, sym, gallery, lda, queries, ldq, sqg, sqq, n_a, n_q, kp, ctl_g, ctl_q, fa, s);
    return fail(SE_ERR_UNSUPPORTED, "pre-filter pass: metric %d", metric);
}

}  // namespace se
