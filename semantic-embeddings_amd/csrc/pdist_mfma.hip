// pdist_mfma.hip -- all-pairs query x gallery distance kernel.
//
// Replaces `pdist = -np.dot(features, features.T)` (evaluate_retrieval.py:59) and
// `pdist = A + B - 2 * C` (evaluate_retrieval.py:61-62).
//
// Arithmetic contract ("canonical arithmetic"): every dot product is ONE
// sequential fp32 FMA chain over k = 0..D-1, which is what v_mfma_f32_32x32x2_f32 computes
// (bit-for-bit an fmaf chain: lanes 0-31 supply k = 2t, lanes 32-63 supply k = 2t+1) and what the
// reference's OpenBLAS sgemm/ssyrk produces for D <= 448.  For larger D the optional K-block list
// restarts the chain per block and adds block results in order.
//
// Structure: persistent workgroups (2 per CU, 512 threads = 4 x 2 waves, 32 x 64 outputs per wave
// = 1 x 2 MFMA tiles, 32 accumulator VGPRs -> 4 waves per SIMD keep the matrix pipe fed while
// other waves stage or store) walk 128 x 128 output tiles.  Operands are staged through LDS in
// K-chunks of 64 with even and odd k de-interleaved, so each lane feeds four MFMA steps from one
// 16-byte ds_read_b128; the row pitch is padded to 68 floats (conflict-free reads).
// The (tile, chunk) sequence is software-pipelined: while the MFMAs of chunk i run out of LDS, the
// global loads of chunk i+1 -- possibly the first chunk of the NEXT tile -- are already in flight
// into registers, and the accumulator stores of a finished tile overlap the next tile's MFMAs.
// All global addressing is "uniform 64-bit tile base + 32-bit lane offset" to keep the VGPR count
// under the 128 that 4 waves per SIMD allow (a spilled prefetch register would serialise the loads).
// Tile order: block b stays on XCD b % 8; every XCD owns a contiguous band of the tile list and
// its workgroups sweep it together, so operand panels are re-read from that XCD's private L2.
//
// Symmetric mode (a == b, the reference's ssyrk case `features . features^T`): only tiles on or
// above the diagonal are computed; an off-diagonal tile is also written transposed (the transposed
// element is the same FMA chain with commuted factors, i.e. bit-identical).  The transposed write
// is 16 bytes per lane straight from the accumulator layout (4 consecutive rows per register
// group).  This halves the MFMA work, which is what bounds the kernel at D = 100
// (2*D flop per 4 output bytes: 3.2 ms of fp32 MFMA vs 1.25 ms of HBM time for 50k x 50k).
#include "se_common.h"
#include <stdlib.h>

namespace se {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef SE_PD_BK
#define SE_PD_BK 64
#endif
#ifndef SE_PD_WGS
#define SE_PD_WGS 2
#endif
constexpr int PD_BM = 128, PD_BN = 128, PD_BK = SE_PD_BK;   // K-chunk staged per barrier pair: 64 or 32
#ifndef SE_PD_THREADS
#define SE_PD_THREADS 512        // 512: 4 x 2 waves of 32 x 64 outputs; 256: 2 x 2 waves of 64 x 64 (smaller workgroups, more of them per CU)
#endif
constexpr int PD_THREADS = SE_PD_THREADS;
constexpr int PD_WAVES = PD_THREADS / 64;
constexpr int PD_WROWS = PD_BM / (PD_WAVES / 2);              // tile rows per wave (its columns: 64)
constexpr int PD_MI = PD_WROWS / 32;                          // 32-row MFMA blocks per wave along m
constexpr int PD_LD = PD_BK + 4;  // padded LDS row pitch in floats
constexpr int PD_SP = PD_BN + 4;  // row pitch of the epilogue stage
// rows of the output tile staged at a time: the whole tile when the operand buffers can hold it (BK = 64), else half
constexpr int PD_SR = (PD_BM * PD_SP <= (PD_BM + PD_BN) * PD_LD) ? PD_BM : PD_BM / 2;
static_assert(PD_SR * PD_SP <= (PD_BM + PD_BN) * PD_LD, "epilogue stage must fit in the operand LDS");
constexpr int PD_GROUP_M = 16;
constexpr int PD_MAX_KB = 16;
constexpr int PD_WGS_PER_CU = SE_PD_WGS;
constexpr int PD_F4R = PD_BK / 4;                            // 16-byte pieces per operand row of a chunk
constexpr int PD_RPP = PD_THREADS / PD_F4R;                  // operand rows covered by one piece per thread
constexpr int PD_NLOAD = PD_BM / PD_RPP;                     // pieces per operand per thread (4 at BK = 64, 2 at BK = 32)
constexpr int64_t PD_MAX_LD = (int64_t)1 << 23;              // 128 rows * ld must fit 32-bit offsets

static_assert(PD_MAX_KB == SE_MAX_KB, "KBlocks capacity");

__device__ __forceinline__ float mask_f(float x, bool keep)
{
    return __uint_as_float(__float_as_uint(x) & (keep ? 0xFFFFFFFFu : 0u));   // branch-free, NaN-safe zeroing
}

// Global -> registers: rows [row0, row0+128) x k [k0, k0+64) of `src`, zero outside [0,nrows) x [0,kend).
// Loads are UNCONDITIONAL (clamped 32-bit offsets from a uniform base, then bit-masking): a branch
// around a load makes hipcc wait for each load before issuing the next one.
template <bool VEC>
__device__ __forceinline__ void pd_load(float4 (&v)[PD_NLOAD], const float *__restrict__ src, uint32_t ld, int64_t row0,
                                        int64_t nrows, int64_t k0, int64_t kend)
{
    const int tid = threadIdx.x;
    const int r0 = tid / PD_F4R, kq = (tid % PD_F4R) * 4;
    const float *base = src + row0 * (int64_t)ld;                 // uniform
    const int rows_here = (int)((nrows - row0 < PD_BM) ? (nrows - row0) : PD_BM);   // >= 1
    const int klen = (int)(kend - k0);                            // valid k in this chunk (may exceed 64)
    if (VEC) {   // ld % 4 == 0, 16-byte aligned base, k0 % 4 == 0: a float4 at k <= ld-4 never leaves its row
        const int kmax = (int)ld - 4 - (int)k0;
        const uint32_t kc = (uint32_t)(k0 + (kq < kmax ? kq : kmax));
#pragma unroll
        for (int it = 0; it < PD_NLOAD; it++) {
            const int r = it * PD_RPP + r0;
            const int rc = r < rows_here ? r : rows_here - 1;
            v[it] = *(const float4 *)(base + ((uint32_t)rc * ld + kc));
        }
    } else {
#pragma unroll
        for (int it = 0; it < PD_NLOAD; it++) {
            const int r = it * PD_RPP + r0;
            const int rc = r < rows_here ? r : rows_here - 1;
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int kk = (kq + j < klen) ? (kq + j) : (klen - 1);
                e[j] = base[(uint32_t)rc * ld + (uint32_t)(k0 + kk)];
            }
            v[it] = make_float4(e[0], e[1], e[2], e[3]);
        }
    }
}

// Registers -> LDS, even k to [0,32), odd k to [32,64) of each row.  Elements outside the valid
// rows / valid k of the chunk (loaded from clamped addresses) are zeroed HERE, at their only use, so no
// predicate stays live across the MFMA phase.
__device__ __forceinline__ void pd_store(float *lds, const float4 (&v)[PD_NLOAD], int rows_here, int klen)
{
    const int tid = threadIdx.x;
    const int r0 = tid / PD_F4R, kq = (tid % PD_F4R) * 4;
    const int nvalid = klen - kq;
    float *o = lds + r0 * PD_LD + (kq >> 1);
#pragma unroll
    for (int it = 0; it < PD_NLOAD; it++) {
        const bool rok = (it * PD_RPP + r0) < rows_here;
        *(float2 *)(o + it * PD_RPP * PD_LD) = make_float2(mask_f(v[it].x, rok && nvalid > 0), mask_f(v[it].z, rok && nvalid > 2));
        *(float2 *)(o + it * PD_RPP * PD_LD + PD_BK / 2) = make_float2(mask_f(v[it].y, rok && nvalid > 1), mask_f(v[it].w, rok && nvalid > 3));
    }
}

// chunk c of a tile -> [k0, kend) and whether it closes a K-block
__device__ __forceinline__ void pd_chunk(int c, const KBlocks &kbs, bool multi, int64_t D, int64_t &k0, int64_t &kend,
                                         bool &closes_kb)
{
    if (!multi) {
        k0 = (int64_t)c * PD_BK;
        kend = D;
        closes_kb = (k0 + PD_BK >= D);
        return;
    }
    int64_t beg = 0;
    int cc = c;
    for (int kb = 0; kb < kbs.n; kb++) {
        const int nch = (kbs.len[kb] + PD_BK - 1) / PD_BK;
        if (cc < nch) {
            k0 = beg + (int64_t)cc * PD_BK;
            kend = beg + kbs.len[kb];
            closes_kb = (cc == nch - 1);
            return;
        }
        cc -= nch;
        beg += kbs.len[kb];
    }
    k0 = kend = D;
    closes_kb = true;
}

// linear tile index -> tile origin.  General: "16 tile-rows deep" grouped order.  Symmetric:
// row-major walk of the upper triangle (tn >= tm).  (A column-strip walk that lets an XCD re-use 32 or 64 B
// panels from its L2 cut FETCH_SIZE 12x -- 2.7 M KB -> 0.22 M KB, the panels otherwise come from Infinity
// Cache -- but measured SLOWER, 3.5 -> 3.7 ms: operand fetch is not what limits this kernel.)
template <bool SYM>
__device__ __forceinline__ void pd_tile_coords(uint32_t t, int tiles_m, int tiles_n, int64_t &m0, int64_t &n0)
{
    // 32-bit arithmetic throughout (the launcher refuses tile counts >= 2^31): a 64-bit division is ~130 instructions here
    if (SYM) {
        const double T = (double)tiles_n;
        int32_t tm = (int32_t)(((2.0 * T + 1.0) - sqrt((2.0 * T + 1.0) * (2.0 * T + 1.0) - 8.0 * (double)t)) * 0.5);
        if (tm < 0) tm = 0;
        if (tm > tiles_m - 1) tm = tiles_m - 1;
        // offset(tm) = tm*T - tm*(tm-1)/2 ; fix rounding   (tiles_n <= 65535 in symmetric mode: the products fit 32 bits)
        while (tm > 0 && (uint32_t)tm * (uint32_t)tiles_n - (uint32_t)tm * (uint32_t)(tm - 1) / 2u > t) tm--;
        while ((uint32_t)(tm + 1) * (uint32_t)tiles_n - (uint32_t)(tm + 1) * (uint32_t)tm / 2u <= t) tm++;
        const uint32_t off = (uint32_t)tm * (uint32_t)tiles_n - (uint32_t)tm * (uint32_t)(tm - 1) / 2u;
        m0 = (int64_t)tm * PD_BM;
        n0 = (int64_t)((uint32_t)tm + (t - off)) * PD_BN;
        return;
    }
    const uint32_t per_group = (uint32_t)PD_GROUP_M * (uint32_t)tiles_n;
    const uint32_t group = t / per_group, in_g = t - group * per_group;
    const uint32_t first_m = group * PD_GROUP_M;
    const uint32_t gsz = ((uint32_t)tiles_m - first_m < (uint32_t)PD_GROUP_M) ? ((uint32_t)tiles_m - first_m) : (uint32_t)PD_GROUP_M;
    const uint32_t col_t = in_g / gsz;
    m0 = (int64_t)(first_m + (in_g - col_t * gsz)) * PD_BM;
    n0 = (int64_t)col_t * PD_BN;
}


// Stage [PD_SR][PD_SP] in LDS -> global rows: thread t moves 16 bytes, 32 lanes cover one 512-byte row segment.
// EUC (the Euclidean matrix output): the stage holds the raw dot products v and the distance is finished HERE, where a thread's four
// columns are the same in every pass -- fl(fl(|row|^2 + |col|^2) - 2 v), spelled as one add and one FMA (2 v is exact, so the FMA rounds
// the same exact difference).  rown / coln: squared norms of the staged rows / columns in LDS (rown[0] = first staged row).
template <bool EUC>
__device__ __forceinline__ void pd_stream_rows(const float *stage, float *gbase, uint32_t ldo, int nrows, int ncols, bool fast, bool nt, bool dry,
                                               const float *rown = nullptr, const float *coln = nullptr)
{
    int tid = threadIdx.x;
    if (EUC) asm volatile("" : "+v"(tid));   // opaque per call: otherwise hipcc hoists the ~20 per-pass LDS / global offsets of the Euclidean form out of the tile loop and parks them in scratch
    const int r0 = tid >> 5, c4 = (tid & 31) * 4;
    constexpr int RPP = PD_THREADS / 32;                         // rows per pass of the workgroup
    char *gb = (char *)gbase;                                    // uniform base + 32-bit byte offsets
    const uint32_t ldo4 = ldo * 4u;
    float4 cn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EUC) cn = *(const float4 *)&coln[c4];
#pragma unroll
    for (int p = 0; p < PD_SR / RPP; p++) {
        const int row = p * RPP + r0;
        float4 v = *(const float4 *)&stage[row * PD_SP + c4];
        if (EUC) {
            const float rn = rown[row];
            v = make_float4(fmaf(-2.0f, v.x, rn + cn.x), fmaf(-2.0f, v.y, rn + cn.y), fmaf(-2.0f, v.z, rn + cn.z), fmaf(-2.0f, v.w, rn + cn.w));
        }
        if (dry) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); continue; }   // tuning aid: stage traffic without HBM writes
        float *dp = (float *)(gb + ((uint32_t)row * ldo4 + (uint32_t)c4 * 4u));
        if (fast) {
            if (nt) __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, (f32x4 *)dp);
            else *(float4 *)dp = v;
        } else if (row < nrows) {
            if (c4 < ncols) dp[0] = v.x;
            if (c4 + 1 < ncols) dp[1] = v.y;
            if (c4 + 2 < ncols) dp[2] = v.z;
            if (c4 + 3 < ncols) dp[3] = v.w;
        }
    }
}

template <int METRIC>
__device__ __forceinline__ float pd_finish(float v, float sa, float sb)
{
    if (METRIC == SE_METRIC_COSINE) return -v;
    if (METRIC == SE_METRIC_EUCLID) return (sa + sb) - 2.0f * v;
    return v;
}

// flags
constexpr int PDF_VEC_A = 1, PDF_VEC_B = 2, PDF_VEC_O = 4, PDF_NO_STORE = 8, PDF_NO_MFMA = 16, PDF_STAGGER = 32, PDF_PLAIN_ST = 64, PDF_NO_GSTORE = 128;

template <int METRIC, bool MULTI_KB, bool SYM, bool VEC, int EPI>
__global__ __launch_bounds__(PD_THREADS, (PD_WAVES / 4) * PD_WGS_PER_CU) void pdist_kernel(
    const float *__restrict__ A, uint32_t lda, const float *__restrict__ Bm, uint32_t ldb,
    const float *__restrict__ sqa, const float *__restrict__ sqb, int64_t Q, int64_t N, int64_t D,
    KBlocks kbs, int nchunks, float *__restrict__ out, uint32_t ldo, int tiles_m, int tiles_n, int64_t ntiles, int flags,
    unsigned long long *prof, FusedArgs fa)
{
    static_assert(EPI != EPI_GROUPMIN || !SYM, "the sample pass walks the general tile order");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sA = smem;
    float *sB = smem + PD_BM * PD_LD;
    // Euclidean matrix output: |a|^2 of the tile's 128 rows and |b|^2 of its 128 columns, behind the operands (the epilogue stage aliases
    // the operands, not this).  Fetched by LDS-DMA (global_load_lds_dword: no register holds them -- the kernel sits at the 128-VGPR
    // limit of four waves per SIMD) at the top of the tile's last chunk, i.e. under its MFMA phase.
    constexpr bool EUC_OUT = METRIC == SE_METRIC_EUCLID && EPI == EPI_STORE;
    [[maybe_unused]] float *sN = smem + (PD_BM + PD_BN) * PD_LD;      // [PD_BM + PD_BN]
    typedef __attribute__((address_space(1))) const void pd_gptr_t;
    typedef __attribute__((address_space(3))) void pd_lptr_t;

    // ---- this workgroup's tile list: XCD-contiguous band, round-robin inside the XCD ----
    const int64_t b = blockIdx.x, G = gridDim.x;
    const int64_t xcd = b & 7, qq = ntiles >> 3, rr = ntiles & 7;
    const int64_t band_beg = (xcd < rr) ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq;
    const int64_t band_len = qq + (xcd < rr ? 1 : 0);
    const int64_t wg_in_xcd = b >> 3, wgs_per_xcd = (G + 7 - xcd) >> 3;  // blocks b' = xcd (mod 8), b' < G
    const int64_t my_tiles = (band_len > wg_in_xcd) ? (band_len - wg_in_xcd + wgs_per_xcd - 1) / wgs_per_xcd : 0;
    if (my_tiles == 0) return;
    // De-phase the persistent workgroups: all tiles cost the same, so without this every workgroup on
    // the chip reaches its store burst at the same moment and the matrix pipes idle behind the HBM queue.
    if (flags & PDF_STAGGER) {
        const int slot = (int)((b >> 3) & 3);
        for (int i = 0; i < slot * 2; i++) __builtin_amdgcn_s_sleep(127);
    }

#ifdef SE_PD_PRIO
    // experiment: static priority for the second workgroup slot of every CU, so that the two co-resident workgroups cannot
    // phase-lock (both in their MFMA phase at half speed, then both staging with the matrix pipe idle)
    if (b >= (G >> 1)) __builtin_amdgcn_s_setprio(SE_PD_PRIO);
#endif
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;      // (PD_WAVES / 2) x 2 waves: PD_WROWS rows x 64 cols each
    const int col = lane & 31, hi = lane >> 5;
    const bool vec_o = flags & PDF_VEC_O;
    const bool nt = !(flags & PDF_PLAIN_ST);   // streaming (nontemporal) stores: the 10 GB result is never re-read by this kernel

    constexpr int NB = 2 * PD_MI;                 // 32 x 32 accumulator blocks per wave: [mi][j] at mi * 2 + j
    f32x16 acc[NB], tot[NB];
#pragma unroll
    for (int j = 0; j < NB; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[j][r] = 0.f; if (MULTI_KB) tot[j][r] = 0.f; }

    const float *pa = sA + (wm * PD_WROWS + col) * PD_LD + hi * (PD_BK / 2);      // block mi: + mi * 32 rows
    const float *pb0 = sB + (wn * 64 + col) * PD_LD + hi * (PD_BK / 2);
    const float *pb1 = pb0 + 32 * PD_LD;

    // ---- prologue: fetch (tile 0, chunk 0) ----
    float4 ra[PD_NLOAD], rb[PD_NLOAD];
    int64_t m0, n0, k0, kend;
    bool closes_kb;
    pd_tile_coords<SYM>((uint32_t)(band_beg + wg_in_xcd), tiles_m, tiles_n, m0, n0);
    pd_chunk(0, kbs, MULTI_KB, D, k0, kend, closes_kb);
#define PD_FETCH()                              \
    pd_load<VEC>(ra, A, lda, m0, Q, k0, kend);  \
    pd_load<VEC>(rb, Bm, ldb, n0, N, k0, kend);
    PD_FETCH()
    bool first_kb = true;

    // tuning aid (SE_PD_PROFILE=1): shader-clock cycles per phase of every workgroup's wave 0
#ifdef SE_TUNING
    uint64_t t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = prof ? __builtin_amdgcn_s_memtime() : 0;
#define PD_T(i) if (prof) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_acc[i] += now - t_last; t_last = now; }
#else
#define PD_T(i)
#endif
    // The chunk loop is ROTATED: an iteration computes on the chunk that already sits in LDS while the loads of the next chunk
    // (same tile, or first chunk of the next tile) are in flight, and stages that next chunk at its end.  Loads are thus issued and
    // consumed inside ONE iteration: with the operand registers carried across the back-edge instead (stage at the loop top), hipcc
    // re-arranged them right behind the loads -- `s_waitcnt vmcnt(4)` + two v_mov in front of every MFMA phase, i.e. every wave of
    // the workgroup sat out a global-load round trip per chunk.
    // (chunk / tile counters are carried, not derived from `it`: a 64-bit `it % nchunks`, `(it + 1) / nchunks` and the divisions of
    //  pd_tile_coords cost ~130 scalar instructions EACH on this ISA -- eight of them per 2-chunk tile, on every wave, in front of a barrier)
    const int64_t total = my_tiles * nchunks;
    int c = 0;                                     // chunk (of its tile) that sits in LDS
    uint32_t tile_i = 0;                           // index of that tile in this workgroup's list
    int64_t cur_m0 = m0, cur_n0 = n0;              // geometry of the chunk in LDS
    int kc = (int)((kend - k0 < PD_BK) ? (kend - k0) : PD_BK);
    bool cur_closes = closes_kb;
    pd_store(sA, ra, (int)((Q - cur_m0 < PD_BM) ? (Q - cur_m0) : PD_BM), kc);
    pd_store(sB, rb, (int)((N - cur_n0 < PD_BN) ? (N - cur_n0) : PD_BN), kc);
    wg_barrier();
#pragma unroll 1
    for (int64_t it = 0; it < total; it++) {
        PD_T(2)
        // ---- request the next chunk (same tile or first chunk of the next tile) ----
        // (Spreading these 8 loads over the first four MFMA groups instead -- the burst costs each wave ~3k cycles of
        // VMEM issue per chunk, SE_PD_PROFILE=1 -- was measured and is slower, 3.5 -> 3.75 ms: a load that blocks
        // inside the MFMA loop stalls the matrix pipe of its wave; so was a register double buffer of the LDS
        // operand reads, 3.5 -> 3.9 ms.)
        const bool last_chunk = (c + 1 == nchunks);
        const bool have_next = it + 1 < total;
        if (EUC_OUT && last_chunk) {
            // norms of the tile in LDS (its epilogue follows this chunk's MFMA phase; the previous tile's epilogue is behind us): waves 0-1 its
            // rows, waves 2-3 its columns, one dword per lane straight into sN (clamped: rows / columns outside the matrix are never written).
            // Issued BEFORE the operand prefetch below: loads return in order, so once those registers have landed, so have these.
            const int w_ = threadIdx.x >> 6;                       // wave-uniform
            if (w_ < (PD_BM + PD_BN) / 64) {
                const int t_ = threadIdx.x & (PD_BM - 1);
                const bool rows_ = w_ < PD_BM / 64;
                const int64_t i_ = rows_ ? (cur_m0 + t_ < Q ? cur_m0 + t_ : Q - 1) : (cur_n0 + t_ < N ? cur_n0 + t_ : N - 1);
                const float *src_ = (rows_ ? sqa : sqb) + i_;
                __builtin_amdgcn_global_load_lds((pd_gptr_t *)src_, (pd_lptr_t *)(sN + w_ * 64), 4, 0, 0);
            }
        }
        if (have_next) {
            const int nc = last_chunk ? 0 : c + 1;
            if (nc == 0) pd_tile_coords<SYM>((uint32_t)(band_beg + wg_in_xcd) + (tile_i + 1u) * (uint32_t)wgs_per_xcd, tiles_m, tiles_n, m0, n0);
            pd_chunk(nc, kbs, MULTI_KB, D, k0, kend, closes_kb);
            PD_FETCH()
        }

        PD_T(3)
        // ---- MFMA over the chunk in LDS: 2 k per step, 4 steps per 16-byte operand read ----
        const int steps = (flags & PDF_NO_MFMA) ? 0 : ((kc + 1) >> 1);
        const int full = steps & ~3;
#define PD_STEP(C)                                                                                \
    _Pragma("unroll") for (int mi = 0; mi < PD_MI; mi++) {                                        \
        acc[mi * 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[mi].C, b0.C, acc[mi * 2], 0, 0, 0);         \
        acc[mi * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[mi].C, b1.C, acc[mi * 2 + 1], 0, 0, 0); \
    }
#define PD_READ(S)                                                                                \
    float4 a4[PD_MI];                                                                             \
    _Pragma("unroll") for (int mi = 0; mi < PD_MI; mi++) a4[mi] = *(const float4 *)(pa + mi * 32 * PD_LD + (S)); \
    const float4 b0 = *(const float4 *)(pb0 + (S));                                               \
    const float4 b1 = *(const float4 *)(pb1 + (S));
        for (int s = 0; s < full; s += 4) {
            PD_READ(s)
            PD_STEP(x) PD_STEP(y) PD_STEP(z) PD_STEP(w)
        }
        if (steps & 3) {
            PD_READ(full)
            PD_STEP(x)
            if ((steps & 3) > 1) { PD_STEP(y) }
            if ((steps & 3) > 2) { PD_STEP(z) }
        }
#undef PD_READ
#undef PD_STEP

        PD_T(4)
        {
            // The next chunk's operands (requested before this MFMA phase) are waited for HERE, through value barriers the compiler
            // cannot move: (1) no use of a loaded register can be scheduled in front of the MFMA phase (hipcc otherwise re-arranges
            // two of them right behind the loads: s_waitcnt vmcnt(4) + v_mov, a global-load round trip in front of every MFMA phase
            // of every wave); (2) behind the stores of a tile epilogue, its vmcnt(N) for these loads would also wait for the stores
            // to be acknowledged.
#pragma unroll
            for (int i = 0; i < PD_NLOAD; i++) {
                asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));
                asm volatile("" : "+v"(rb[i].x), "+v"(rb[i].y), "+v"(rb[i].z), "+v"(rb[i].w));
            }
        }
        if (MULTI_KB && cur_closes) {
#pragma unroll
            for (int j = 0; j < NB; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    tot[j][r] = first_kb ? acc[j][r] : (tot[j][r] + acc[j][r]);
                    acc[j][r] = 0.f;
                }
            first_kb = false;
        }

        // ---- tile finished: accumulators (col = lane & 31, row = (r&3) + 8*(r>>2) + 4*hi) -> global ----
        if (last_chunk) {
            if (EUC_OUT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile's norms have landed in sN (hipcc does not track the LDS side of the DMA); the epilogue's first barrier publishes them
            if (flags & PDF_NO_STORE) {
#pragma unroll
                for (int j = 0; j < NB; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) asm volatile("" ::"v"(acc[j][r]));
            } else if (EPI != EPI_STORE) {
                // ---- fused top-k passes (se_retrieve_topk): rows of the tile = gallery, columns = queries; lane (col, hi) of wave
                //      (wm, wn) holds, per 32 x 32 block (mi, j), 16 gallery rows of the ONE query  cur_n0 + wn*64 + j*32 + col ----
                const int rows_here = (int)((Q - cur_m0 < PD_BM) ? (Q - cur_m0) : PD_BM);
                const int cols_here = (int)((N - cur_n0 < PD_BN) ? (N - cur_n0) : PD_BN);
                const bool full_rows = rows_here == PD_BM;
                const int lr0 = wm * PD_WROWS + 4 * hi;
#define PD_FVAL(MI_, J, R)                                                                                                   \
    pd_finish<METRIC>(MULTI_KB ? tot[(MI_) * 2 + (J)][R] : acc[(MI_) * 2 + (J)][R],                                            \
                      METRIC == SE_METRIC_EUCLID ? fsa[((R) & 3) + 4 * ((R) >> 2)] : 0.f, sbq)
                if (EPI == EPI_GROUPMIN) {
#pragma unroll
                    for (int mi = 0; mi < PD_MI; mi++) {
                        float fsa[16];
                        if (METRIC == SE_METRIC_EUCLID) {
#pragma unroll
                            for (int r = 0; r < 16; r++) {
                                const int lr = lr0 + mi * 32 + (r & 3) + 8 * (r >> 2);
                                fsa[r] = sqa[(cur_m0 + (lr < rows_here ? lr : rows_here - 1)) * fa.sqa_stride];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            const int lc = wn * 64 + j * 32 + col;
                            const bool qok = lc < cols_here;
                            const int64_t qg = cur_n0 + (qok ? lc : cols_here - 1);
                            const float sbq = METRIC == SE_METRIC_EUCLID ? sqb[qg] : 0.f;
                            float m = __builtin_inff();
                            bool any = false;
#pragma unroll
                            for (int r = 0; r < 16; r++) {
                                const int lr = lr0 + mi * 32 + (r & 3) + 8 * (r >> 2);
                                const float v = PD_FVAL(mi, j, r);
                                const bool ok = (full_rows || lr < rows_here) && (v == v);
                                m = (ok && v < m) ? v : m;
                                any = any || ok;
                            }
                            if (!any) m = __builtin_nanf("");     // a group of NaNs only: sorted last by the threshold kernel
                            if (qok) fa.gm[qg * fa.gm_ld + (cur_m0 / PD_BM) * (PD_WAVES * PD_MI) + (wm * PD_MI + mi) * 2 + hi] = m;
                        }
                    }
                } else {
                    // ---- EPI_FILTER.  Orientation 1: lanes = queries (tile columns), registers = gallery rows.  Orientation 2 (all-pairs
                    //      calls, off-diagonal tiles): lanes = queries (tile ROWS), values read back from the staged tile.  Per orientation:
                    //      counts of both 32-query blocks, then BOTH slot reservations (returning atomics) in flight together, one wait, stores. ----
                    static_assert(PD_MI == 1 || EPI != EPI_FILTER, "the filter epilogue is written for 32-row wave slabs");
                    constexpr int mi = 0;
                    const bool mirror = SYM && (cur_m0 != cur_n0);
                    {
                    float fsa[16];
                    if (METRIC == SE_METRIC_EUCLID) {
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int lr = lr0 + (r & 3) + 8 * (r >> 2);
                            fsa[r] = sqa[(cur_m0 + (lr < rows_here ? lr : rows_here - 1)) * fa.sqa_stride];
                        }
                    }
                    int64_t qgj[2];
                    float tauj[2], sbqj[2];
                    unsigned cntj[2], slotj[2];
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const int lc = wn * 64 + j * 32 + col;
                        const bool qok = lc < cols_here;
                        qgj[j] = cur_n0 + (qok ? lc : cols_here - 1);
                        tauj[j] = qok ? fa.tau[qgj[j]] : __builtin_nanf("");       // NaN: nothing passes
                        sbqj[j] = METRIC == SE_METRIC_EUCLID ? sqb[qgj[j]] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const float sbq = sbqj[j];
                        unsigned cnt = 0;
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int lr = lr0 + (r & 3) + 8 * (r >> 2);
                            cnt += ((PD_FVAL(mi, j, r) <= tauj[j]) && (full_rows || lr < rows_here)) ? 1u : 0u;
                        }
                        cntj[j] = cnt;
                    }
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        slotj[j] = 0;
                        if (cntj[j]) slotj[j] = atomicAdd(&fa.rowcnt[qgj[j]], cntj[j]);
                    }
                    // ONE wait for both reservations, in straight-line code: left to hipcc, every conditional store block below re-waits with
                    // vmcnt(0) (merged control flow) -- which also waits for the previous block's STORE to be acknowledged, i.e. serialises
                    // up to 32 store round trips per tile.  (Keeping these reservations in flight across the mirrored part below, for ONE
                    // atomic round trip per tile, was measured: the longer live ranges spill 140 B and the pass takes 5.4 instead of 3.9 ms.)
                    asm volatile("" : "+v"(slotj[0]), "+v"(slotj[1]));
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        if (cntj[j]) {
                            const float sbq = sbqj[j];
                            unsigned slot = slotj[j];
                            uint2 *lst = fa.lists + qgj[j] * fa.cap;
#pragma unroll
                            for (int r = 0; r < 16; r++) {
                                const int lr = lr0 + (r & 3) + 8 * (r >> 2);
                                const float v = PD_FVAL(mi, j, r);
                                if ((v <= tauj[j]) && (full_rows || lr < rows_here)) {
                                    if (slot < (unsigned)fa.cap) lst[slot] = make_uint2(__float_as_uint(v), (uint32_t)(cur_m0 + lr));
                                    slot++;
                                }
                            }
                        }
                    }
                    }
#define PD_MVALS(J, G, E)                                                                                                    \
    float E[4];                                                                                                              \
    {                                                                                                                        \
        const float4 raw_ = *(const float4 *)(strow[J] + 8 * (G));                                                           \
        const float rv_[4] = {raw_.x, raw_.y, raw_.z, raw_.w};                                                               \
        _Pragma("unroll") for (int cc = 0; cc < 4; cc++) {                                                                   \
            const int gc_ = gl0 + 8 * (G) + cc;                                                                              \
            const float sb_ = METRIC == SE_METRIC_EUCLID ? sqb[cur_n0 + (gc_ < cols_here ? gc_ : cols_here - 1)] : 0.f;      \
            E[cc] = pd_finish<METRIC>(rv_[cc], saq2[J], sb_);                                                                \
        }                                                                                                                    \
    }
                    if (mirror) {
                        // ---- orientation 2: stage the RAW dot products (finished after the read-back: the Euclidean epilogue then needs a
                        //      lane's 16 column norms only four at a time), count, reserve, store ----
                        int64_t qg2[2];
                        float tau2[2], saq2[2];
                        unsigned cnt2[2], slot2[2];
                        const float *strow[2];
                        const int gl0 = wm * PD_WROWS + 4 * hi;            // this lane's gallery columns of the tile: gl0 + 8 g + c
                        const bool full_cols = cols_here == PD_BN;
                        // all-pairs call (queries == gallery): only tiles on or above the diagonal are computed; an off-diagonal tile also
                        // serves the MIRRORED pairs -- queries = its rows, gallery items = its columns (the transposed element is the same
                        // FMA chain with commuted factors: bit-identical).  The tile goes through the idle operand LDS so that a lane again
                        // owns ONE query (a tile row) and reads 16 of its gallery values as 4 x 16 bytes.
                        static_assert(PD_SR == PD_BM || EPI != EPI_FILTER || !SYM, "the mirrored filter stages a whole tile");
                        wg_barrier();   // every wave has finished this chunk's MFMA reads of the operand LDS
#pragma unroll
                        for (int j = 0; j < 2; j++)
#pragma unroll
                            for (int r = 0; r < 16; r++)
                                smem[(lr0 + (r & 3) + 8 * (r >> 2)) * PD_SP + wn * 64 + j * 32 + col] = MULTI_KB ? tot[j][r] : acc[j][r];
                        wg_barrier();
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            const int ql = wn * 64 + j * 32 + col;                    // this lane's query = tile row ql
                            const bool qok = ql < rows_here;
                            qg2[j] = cur_m0 + (qok ? ql : rows_here - 1);
                            tau2[j] = qok ? fa.tau[qg2[j]] : __builtin_nanf("");
                            saq2[j] = METRIC == SE_METRIC_EUCLID ? sqa[qg2[j] * fa.sqa_stride] : 0.f;
                            strow[j] = smem + ql * PD_SP + gl0;
                        }
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            unsigned cnt = 0;
#pragma unroll
                            for (int g = 0; g < 4; g++) {
                                PD_MVALS(j, g, e)
#pragma unroll
                                for (int cc = 0; cc < 4; cc++) cnt += ((e[cc] <= tau2[j]) && (full_cols || gl0 + 8 * g + cc < cols_here)) ? 1u : 0u;
                            }
                            cnt2[j] = cnt;
                        }
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            slot2[j] = 0;
                            if (cnt2[j]) slot2[j] = atomicAdd(&fa.rowcnt[qg2[j]], cnt2[j]);
                        }
                        asm volatile("" : "+v"(slot2[0]), "+v"(slot2[1]));   // one wait for the mirrored reservations
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            if (cnt2[j]) {
                                unsigned slot = slot2[j];
                                uint2 *lst = fa.lists + qg2[j] * fa.cap;
#pragma unroll
                                for (int g = 0; g < 4; g++) {
                                    PD_MVALS(j, g, e)
#pragma unroll
                                    for (int cc = 0; cc < 4; cc++) {
                                        if ((e[cc] <= tau2[j]) && (full_cols || gl0 + 8 * g + cc < cols_here)) {
                                            if (slot < (unsigned)fa.cap) lst[slot] = make_uint2(__float_as_uint(e[cc]), (uint32_t)(cur_n0 + gl0 + 8 * g + cc));
                                            slot++;
                                        }
                                    }
                                }
                            }
                        }
                        // (the barrier behind the tile epilogue orders these LDS reads before the next chunk's operands overwrite the stage)
                    }
#undef PD_MVALS
                }
#undef PD_FVAL
            } else {
                // Staged epilogue: the finished tile goes through the (now idle) operand LDS so that HBM sees
                // whole 512-byte row segments, two rows per wave instruction, as streaming (nontemporal) 16-byte
                // stores -- 5.2 TB/s on this pattern vs 4.1 TB/s for any store straight from the accumulator
                // layout and 3.7 TB/s for the transposed tile (tools/probes/store_patterns2.hip).
                const bool mirror = SYM && (cur_m0 != cur_n0);
                const int rows_here = (int)((Q - cur_m0 < PD_BM) ? (Q - cur_m0) : PD_BM);
                const int cols_here = (int)((N - cur_n0 < PD_BN) ? (N - cur_n0) : PD_BN);
                const bool fast = (rows_here == PD_BM) && (cols_here == PD_BN) && vec_o;
                int lr0 = wm * PD_WROWS + 4 * hi;                                 // + mi * 32 + (r&3) + 8*(r>>2)
                asm volatile("" : "+v"(lr0));   // opaque per tile: nothing of the epilogue is hoisted out of the tile loop
                // (values are finished at the point of use: a [2][16] copy of the tile would cost 32 more VGPRs)
                // (Euclidean: the stage takes the raw dot products, pd_stream_rows finishes them with the norms staged in sN)
#define PD_VAL(MI_, J, R)                                                                                                    \
    (METRIC == SE_METRIC_EUCLID ? (MULTI_KB ? tot[(MI_) * 2 + (J)][R] : acc[(MI_) * 2 + (J)][R])                               \
                                : pd_finish<METRIC>(MULTI_KB ? tot[(MI_) * 2 + (J)][R] : acc[(MI_) * 2 + (J)][R], 0.f, 0.f))
#pragma unroll
                for (int h = 0; h < PD_BM / PD_SR; h++) {
                    wg_barrier();   // operands of the last chunk / the previous stage contents are no longer needed
                    PD_T(6)
                    // tile rows [h SR, (h+1) SR) -> stage[row][col]: per instruction lanes 0-31 fill 32 consecutive floats of one row
#pragma unroll
                    for (int mi = 0; mi < PD_MI; mi++)
                        if ((wm * PD_WROWS + mi * 32) / PD_SR == h) {
#pragma unroll
                            for (int j = 0; j < 2; j++)
#pragma unroll
                                for (int r = 0; r < 16; r++)
                                    smem[(lr0 + mi * 32 - h * PD_SR + (r & 3) + 8 * (r >> 2)) * PD_SP + wn * 64 + j * 32 + col] = PD_VAL(mi, j, r);
                        }
                    PD_T(7)
                    wg_barrier();
                    PD_T(8)
                    pd_stream_rows<EUC_OUT>(smem, out + ((cur_m0 + h * PD_SR) * (int64_t)ldo + cur_n0), ldo, rows_here - h * PD_SR, cols_here, fast, nt,
                                            flags & PDF_NO_GSTORE, sN + h * PD_SR, sN + PD_BM);
                    PD_T(9)
                }
                if (mirror) {
#pragma unroll
                    for (int h = 0; h < PD_BN / PD_SR; h++) {
                        // transposed tile rows (= tile columns) [h SR, (h+1) SR) -> stage[col][row]: a lane owns 4 consecutive
                        // rows of its column = 16 bytes
                        wg_barrier();
                        PD_T(6)
                        if ((wn * 64) / PD_SR == h) {
#pragma unroll
                            for (int mi = 0; mi < PD_MI; mi++)
#pragma unroll
                                for (int j = 0; j < 2; j++)
#pragma unroll
                                    for (int g = 0; g < 4; g++)
                                        *(float4 *)&smem[(wn * 64 - h * PD_SR + j * 32 + col) * PD_SP + lr0 + mi * 32 + 8 * g] =
                                            make_float4(PD_VAL(mi, j, 4 * g), PD_VAL(mi, j, 4 * g + 1), PD_VAL(mi, j, 4 * g + 2), PD_VAL(mi, j, 4 * g + 3));
                        }
                        PD_T(7)
                        wg_barrier();
                        PD_T(8)
                        pd_stream_rows<EUC_OUT>(smem, out + ((cur_n0 + h * PD_SR) * (int64_t)ldo + cur_m0), ldo, cols_here - h * PD_SR, rows_here, fast, nt,
                                                flags & PDF_NO_GSTORE, sN + PD_BM + h * PD_SR, sN);
                        PD_T(9)
                    }
                }
                // (the barrier behind the tile epilogue orders these LDS reads before the next chunk's operands overwrite the stage)
#undef PD_VAL
            }
#pragma unroll
            for (int j = 0; j < NB; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
            first_kb = true;
            PD_T(5)
        }
        wg_barrier();   // every wave has finished reading this chunk (and, EPI_STORE, the epilogue stage) out of LDS
        PD_T(0)
        if (have_next) {
            cur_m0 = m0; cur_n0 = n0;
            kc = (int)((kend - k0 < PD_BK) ? (kend - k0) : PD_BK);
            cur_closes = closes_kb;
            pd_store(sA, ra, (int)((Q - cur_m0 < PD_BM) ? (Q - cur_m0) : PD_BM), kc);
            pd_store(sB, rb, (int)((N - cur_n0 < PD_BN) ? (N - cur_n0) : PD_BN), kc);
        }
        PD_T(1)
        wg_barrier();
        c = last_chunk ? 0 : c + 1;
        tile_i += last_chunk ? 1u : 0u;
    }
#ifdef SE_TUNING
    if (prof && threadIdx.x == 0)
        for (int i = 0; i < 12; i++) atomicAdd(&prof[i], (unsigned long long)t_acc[i]);
#endif
#undef PD_T
#undef PD_FETCH
}

static int pd_num_cus()
{
    static const int cus = [] {   // (thread-safe one-time initialisation; one process drives one GPU model)
        int dev = 0, n = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        return n > 0 ? n : 256;
    }();
    return cus;
}

template <int METRIC, bool MULTI, bool SYM, bool VEC, int EPI = EPI_STORE>
static int launch_pdist3(const float *a, int64_t lda, const float *b, int64_t ldb, const float *sqa, const float *sqb,
                         int64_t q, int64_t n, int64_t d, const KBlocks &kbs, float *out, int64_t ldo, hipStream_t s,
                         const FusedArgs &fa = FusedArgs{nullptr, 0, nullptr, nullptr, nullptr, 0, 1})
{
    const int tiles_m = (int)((q + PD_BM - 1) / PD_BM), tiles_n = (int)((n + PD_BN - 1) / PD_BN);
    const int64_t ntiles = SYM ? ((int64_t)tiles_n * (tiles_n + 1) / 2) : ((int64_t)tiles_m * tiles_n);
    const size_t lds = ((size_t)(PD_BM + PD_BN) * PD_LD + (METRIC == SE_METRIC_EUCLID && EPI == EPI_STORE ? PD_BM + PD_BN : 0)) * sizeof(float);
    if (ntiles >= ((int64_t)1 << 31) || (int64_t)PD_GROUP_M * tiles_n >= ((int64_t)1 << 31) || (SYM && tiles_n > 65535))
        return fail(SE_ERR_UNSUPPORTED, "se_pairwise_dist: %lld output tiles exceed the 32-bit tile counter -- split the call", (long long)ntiles);
    int flags = 0;
    if ((lda % 4 == 0) && ((((uintptr_t)a) & 15) == 0)) flags |= PDF_VEC_A;
    if ((ldb % 4 == 0) && ((((uintptr_t)b) & 15) == 0)) flags |= PDF_VEC_B;
    if ((ldo % 4 == 0) && ((((uintptr_t)out) & 15) == 0)) flags |= PDF_VEC_O;
    flags |= PDF_STAGGER;
    if (kTuning) {   // -DSE_TUNING build only (the product library never skips work): ablations of the tuning sessions
        static const int tune = [] {
            int f = 0;
            if (const char *e = tuning_env("SE_PD_ABLATE")) { const int a = atoi(e); f |= (a & 3) * PDF_NO_STORE; if (a & 4) f |= PDF_NO_GSTORE; }   // 1 = no epilogue, 2 = no MFMA, 4 = epilogue without global stores
            if (tuning_env("SE_PD_NOSTAGGER")) f |= 1 << 30;
            if (tuning_env("SE_PD_PLAIN_ST")) f |= PDF_PLAIN_ST;
            return f;
        }();
        flags |= tune & ~(1 << 30);
        if (tune & (1 << 30)) flags &= ~PDF_STAGGER;
    }
    int nchunks = 0;
    for (int i = 0; i < kbs.n; i++) nchunks += (kbs.len[i] + PD_BK - 1) / PD_BK;
    int64_t grid = (int64_t)pd_num_cus() * PD_WGS_PER_CU;
    grid = grid / 8 * 8;
    if (grid > ntiles) grid = ntiles;
    if (grid < 1) grid = 1;
    auto kern = pdist_kernel<METRIC, MULTI, SYM, VEC, EPI>;
    SE_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    static const bool profile = tuning_env("SE_PD_PROFILE") != nullptr;   // -DSE_TUNING build only: allocates, synchronises, prints
    unsigned long long *prof = nullptr;
    if (profile) {
        SE_HIP_CHECK(hipMalloc((void **)&prof, 12 * sizeof(unsigned long long)));
        SE_HIP_CHECK(hipMemsetAsync(prof, 0, 12 * sizeof(unsigned long long), s));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(PD_THREADS), lds, s, a, (uint32_t)lda, b, (uint32_t)ldb, sqa, sqb, q, n,
                       d, kbs, nchunks, out, (uint32_t)ldo, tiles_m, tiles_n, ntiles, flags, prof, fa);
    SE_LAUNCH_CHECK();
    if (profile) {
        unsigned long long h[12];
        SE_HIP_CHECK(hipStreamSynchronize(s));
        SE_HIP_CHECK(hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost));
        SE_HIP_CHECK(hipFree(prof));
        static const char *names[12] = {"wait-barrier", "stage-operands", "barrier", "prefetch-issue", "mfma", "epilogue-rest",
                                        "epi-barrier-in", "epi-stage-write", "epi-barrier", "epi-stream-out", "-", "-"};
        double tot = 0;
        for (int i = 0; i < 12; i++) tot += (double)h[i];
        fprintf(stderr, "[se_pairwise_dist profile] sym=%d grid=%lld tiles=%lld:", (int)SYM, (long long)grid, (long long)ntiles);
        for (int i = 0; i < 10; i++) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * (double)h[i] / tot);
        fprintf(stderr, "  (%.0f cycles per tile per workgroup)\n", tot / (double)ntiles);
    }
    return SE_OK;
}

template <int METRIC, bool MULTI, bool SYM>
static int launch_pdist2(const float *a, int64_t lda, const float *b, int64_t ldb, const float *sqa, const float *sqb,
                         int64_t q, int64_t n, int64_t d, const KBlocks &kbs, float *out, int64_t ldo, hipStream_t s)
{
    // 16-byte operand loads need aligned bases, row pitches that are multiples of 4 and K-blocks that start on
    // multiples of 4 (chunk starts are then multiples of 4 as well)
    bool vec = (lda % 4 == 0) && ((((uintptr_t)a) & 15) == 0) && (ldb % 4 == 0) && ((((uintptr_t)b) & 15) == 0);
    int64_t beg = 0;
    for (int i = 0; i < kbs.n; i++) { if (beg & 3) vec = false; beg += kbs.len[i]; }
    return vec ? launch_pdist3<METRIC, MULTI, SYM, true>(a, lda, b, ldb, sqa, sqb, q, n, d, kbs, out, ldo, s)
               : launch_pdist3<METRIC, MULTI, SYM, false>(a, lda, b, ldb, sqa, sqb, q, n, d, kbs, out, ldo, s);
}

template <int METRIC>
static int launch_pdist(const float *a, int64_t lda, const float *b, int64_t ldb, const float *sqa, const float *sqb,
                        int64_t q, int64_t n, int64_t d, const KBlocks &kbs, bool multi, float *out, int64_t ldo,
                        hipStream_t s)
{
    // ssyrk case: same matrix on both sides (and, for the Euclidean epilogue, the same norms)
    const bool sym = (a == b) && (lda == ldb) && (q == n) && (METRIC != SE_METRIC_EUCLID || sqa == sqb) && n > PD_BN;
    if (sym) return multi ? launch_pdist2<METRIC, true, true>(a, lda, b, ldb, sqa, sqb, q, n, d, kbs, out, ldo, s)
                          : launch_pdist2<METRIC, false, true>(a, lda, b, ldb, sqa, sqb, q, n, d, kbs, out, ldo, s);
    return multi ? launch_pdist2<METRIC, true, false>(a, lda, b, ldb, sqa, sqb, q, n, d, kbs, out, ldo, s)
                 : launch_pdist2<METRIC, false, false>(a, lda, b, ldb, sqa, sqb, q, n, d, kbs, out, ldo, s);
}

// ---- the two tile-loop passes of the fused distance + top-k (driver: topk.hip) ----
template <int METRIC, int EPI>
static int launch_fused2(const float *g, int64_t ldg, const float *qs, int64_t ldq, const float *sqg, const float *sqq, int64_t n_a,
                         int64_t n_q, int64_t d, const KBlocks &kbs, bool multi, const FusedArgs &fa, hipStream_t s)
{
    bool vec = (ldg % 4 == 0) && ((((uintptr_t)g) & 15) == 0) && (ldq % 4 == 0) && ((((uintptr_t)qs) & 15) == 0);
    int64_t beg = 0;
    for (int i = 0; i < kbs.n; i++) { if (beg & 3) vec = false; beg += kbs.len[i]; }
    if constexpr (EPI == EPI_FILTER) {
        // all-pairs call (every item is query and gallery item, the evaluate_retrieval.py case): upper-triangle tile walk, every
        // off-diagonal tile filtered in both orientations -- half the MFMA work
        const bool sym = (g == qs) && (ldg == ldq) && (n_a == n_q) && (METRIC != SE_METRIC_EUCLID || sqg == sqq) && n_a > PD_BN &&
                         !tuning_env("SE_TOPK_NOSYM");
        if (sym) {
            if (multi) return vec ? launch_pdist3<METRIC, true, true, true, EPI>(g, ldg, qs, ldq, sqg, sqq, n_a, n_q, d, kbs, nullptr, n_q, s, fa)
                                  : launch_pdist3<METRIC, true, true, false, EPI>(g, ldg, qs, ldq, sqg, sqq, n_a, n_q, d, kbs, nullptr, n_q, s, fa);
            return vec ? launch_pdist3<METRIC, false, true, true, EPI>(g, ldg, qs, ldq, sqg, sqq, n_a, n_q, d, kbs, nullptr, n_q, s, fa)
                       : launch_pdist3<METRIC, false, true, false, EPI>(g, ldg, qs, ldq, sqg, sqq, n_a, n_q, d, kbs, nullptr, n_q, s, fa);
        }
    }
    if (multi) return vec ? launch_pdist3<METRIC, true, false, true, EPI>(g, ldg, qs, ldq, sqg, sqq, n_a, n_q, d, kbs, nullptr, n_q, s, fa)
                          : launch_pdist3<METRIC, true, false, false, EPI>(g, ldg, qs, ldq, sqg, sqq, n_a, n_q, d, kbs, nullptr, n_q, s, fa);
    return vec ? launch_pdist3<METRIC, false, false, true, EPI>(g, ldg, qs, ldq, sqg, sqq, n_a, n_q, d, kbs, nullptr, n_q, s, fa)
               : launch_pdist3<METRIC, false, false, false, EPI>(g, ldg, qs, ldq, sqg, sqq, n_a, n_q, d, kbs, nullptr, n_q, s, fa);
}

int launch_fused_pass(int epi, const float *gallery, int64_t lda, const float *queries, int64_t ldq, const float *sqg, const float *sqq,
                      int64_t n_a, int64_t n_q, int64_t d, int metric, const KBlocks &kbs, bool multi, const FusedArgs &fa, hipStream_t s)
{
    if (lda >= PD_MAX_LD || ldq >= PD_MAX_LD) return fail(SE_ERR_UNSUPPORTED, "fused top-k pass: leading dimension too large");
    if (metric == SE_METRIC_COSINE) {
        return epi == EPI_GROUPMIN ? launch_fused2<SE_METRIC_COSINE, EPI_GROUPMIN>(gallery, lda, queries, ldq, sqg, sqq, n_a, n_q, d, kbs, multi, fa, s)
                                   : launch_fused2<SE_METRIC_COSINE, EPI_FILTER>(gallery, lda, queries, ldq, sqg, sqq, n_a, n_q, d, kbs, multi, fa, s);
    }
    if (metric == SE_METRIC_EUCLID) {
        return epi == EPI_GROUPMIN ? launch_fused2<SE_METRIC_EUCLID, EPI_GROUPMIN>(gallery, lda, queries, ldq, sqg, sqq, n_a, n_q, d, kbs, multi, fa, s)
                                   : launch_fused2<SE_METRIC_EUCLID, EPI_FILTER>(gallery, lda, queries, ldq, sqg, sqq, n_a, n_q, d, kbs, multi, fa, s);
    }
    return fail(SE_ERR_UNSUPPORTED, "fused top-k pass: metric %d", metric);
}

int64_t pdist_max_ld() { return PD_MAX_LD; }

int make_kblocks(const char *who, const int32_t *kblocks, int nkb, int64_t d, KBlocks &kbs, bool &multi)
{
    kbs.n = 1;
    kbs.len[0] = (int)d;
    multi = false;
    if (kblocks && nkb > 1) {
        if (nkb > SE_MAX_KB) return fail(SE_ERR_UNSUPPORTED, "%s: at most %d K-blocks", who, SE_MAX_KB);
        int64_t sum = 0;
        for (int i = 0; i < nkb; i++) {
            if (kblocks[i] <= 0) return fail(SE_ERR_INVALID, "%s: K-block %d has length %d", who, i, kblocks[i]);
            kbs.len[i] = kblocks[i];
            sum += kblocks[i];
        }
        if (sum != d) return fail(SE_ERR_INVALID, "%s: K-blocks sum to %lld, expected %lld", who, (long long)sum, (long long)d);
        kbs.n = nkb;
        multi = true;
    }
    return SE_OK;
}

}  // namespace se

#ifdef SE_PD_WS_BUILD
namespace se {
// tools/experiments/pdist_ws.hip: wave-specialised variant, measured slower (3.8 vs 3.4 ms) and therefore NOT part of the product
// library; a tuning build can link it in with -DSE_PD_WS_BUILD.  SE_OK = done, 1 = not applicable (use the kernel in this file)
int pdist_ws_try(const float *a, int64_t lda, const float *b, int64_t ldb, const float *sqa, const float *sqb, int64_t q, int64_t n,
                 int64_t d, int metric, float *out, int64_t ldo, hipStream_t s);
}
#endif

using namespace se;

extern "C" int se_pairwise_dist(const float *a, int64_t lda, const float *b, int64_t ldb, const float *sqa,
                                const float *sqb, int64_t q, int64_t n, int64_t d, int metric,
                                const int32_t *kblocks, int nkb, float *out, int64_t ldo, se_stream_t stream)
{
    if (q < 0 || n < 0 || d <= 0) return fail(SE_ERR_INVALID, "se_pairwise_dist: bad shape q=%lld n=%lld d=%lld", (long long)q, (long long)n, (long long)d);
    if (q == 0 || n == 0) return SE_OK;
    if (!a || !b || !out) return fail(SE_ERR_INVALID, "se_pairwise_dist: null pointer");
    if (lda < d || ldb < d || ldo < n) return fail(SE_ERR_INVALID, "se_pairwise_dist: leading dimension too small");
    if (metric == SE_METRIC_EUCLID && (!sqa || !sqb)) return fail(SE_ERR_INVALID, "se_pairwise_dist: SE_METRIC_EUCLID needs sqa and sqb");
    if (lda >= PD_MAX_LD || ldb >= PD_MAX_LD || ldo >= PD_MAX_LD)
        return fail(SE_ERR_UNSUPPORTED, "se_pairwise_dist: leading dimensions must be < %lld elements", (long long)PD_MAX_LD);
    KBlocks kbs;
    bool multi = false;
    if (const int rc = make_kblocks("se_pairwise_dist", kblocks, nkb, d, kbs, multi)) return rc;
    hipStream_t s = (hipStream_t)stream;
#ifdef SE_PD_WS_BUILD
    if (!multi && (metric == SE_METRIC_COSINE || metric == SE_METRIC_EUCLID || metric == SE_METRIC_DOT)) {
        const int rc = pdist_ws_try(a, lda, b, ldb, sqa, sqb, q, n, d, metric, out, ldo, s);
        if (rc != 1) return rc;
    }
#endif
    switch (metric) {
        case SE_METRIC_COSINE: return launch_pdist<SE_METRIC_COSINE>(a, lda, b, ldb, sqa, sqb, q, n, d, kbs, multi, out, ldo, s);
        case SE_METRIC_EUCLID: return launch_pdist<SE_METRIC_EUCLID>(a, lda, b, ldb, sqa, sqb, q, n, d, kbs, multi, out, ldo, s);
        case SE_METRIC_DOT: return launch_pdist<SE_METRIC_DOT>(a, lda, b, ldb, sqa, sqb, q, n, d, kbs, multi, out, ldo, s);
        default: return fail(SE_ERR_INVALID, "se_pairwise_dist: unknown metric %d", metric);
    }
}
