// pdist_ws.hip -- wave-specialised variant of the all-pairs distance kernel.
//
// Same arithmetic contract and tile order as pdist_mfma.hip (one sequential fp32 FMA chain per output on
// v_mfma_f32_32x32x2_f32, upper-triangle tiles + mirrored write in symmetric mode), different execution structure,
// aimed at what the phase profile of that kernel shows (SE_PD_PROFILE): with two symmetric workgroups per CU each
// one spends ~60 % of a tile in phases that issue no MFMA (operand staging, load issue, barrier waits, an epilogue
// that blocks on HBM back-pressure), more than its partner's MFMA phase can cover.
//
// Here ONE 1024-thread workgroup owns a CU and its 16 waves have two roles:
//   * waves 0-7  ("matrix" waves, 4 x 2 over the 128 x 128 tile, 32 x 64 outputs each) only read operands from LDS,
//     issue MFMAs and, at tile boundaries, write the finished tile (then its transpose) into an LDS stage;
//     they never touch global memory (Euclidean: two tiny norm loads per tile), so nothing outside the matrix
//     pipe can stall them between barriers;
//   * waves 8-15 ("mover" waves) do everything else: they fetch the NEXT K-chunk global -> registers -> LDS
//     (de-interleaved even/odd k like the base kernel) into the other half of a double buffer while the matrix waves
//     work on the current one, and they stream the staged tile of the PREVIOUS tile to HBM (512-byte row segments,
//     nontemporal 16-byte stores) in the shadow of the current tile's MFMAs.  A mover wave blocked on the store
//     queue delays nobody until the next chunk barrier.
// One barrier per K-chunk.  K is cut into >= 4 nearly equal chunks of at most 32 (multiples of 4), so that the four
// hand-offs of a symmetric tile (stage tile -> stream it -> stage transpose -> stream it) each get one chunk:
//   chunk n-1 of tile T: matrix waves stage T            chunk 0 of T+1: movers stream T
//   chunk 1 of T+1:      matrix waves stage T transposed  chunk 2 of T+1: movers stream it
// LDS: 2 x (128 + 128) x 36 floats of operands (73,728 B) + 128 x 132 floats of stage (67,584 B) = 141,312 B.
// Restrictions (everything else goes to pdist_mfma.hip): single K-block, 16-byte aligned rows (lda, ldb, ldo
// multiples of 4), d >= 16.
#include "se_common.h"
#include <stdlib.h>

namespace se {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WS_BM = 128, WS_BN = 128, WS_BK = 32;
constexpr int WS_THREADS = 1024, WS_MOVERS = 512;
constexpr int WS_LD = WS_BK + 4;             // operand row pitch (floats)
constexpr int WS_SP = WS_BN + 4;             // stage row pitch (floats)
constexpr int WS_OP = WS_BM * WS_LD;         // floats per operand buffer
constexpr int WS_MAX_CHUNKS = 64;

__device__ __forceinline__ float ws_mask(float x, bool keep) { return __uint_as_float(__float_as_uint(x) & (keep ? 0xFFFFFFFFu : 0u)); }

template <int METRIC>
__device__ __forceinline__ float ws_finish(float v, float sa, float sb)
{
    if (METRIC == SE_METRIC_COSINE) return -v;
    if (METRIC == SE_METRIC_EUCLID) return (sa + sb) - 2.0f * v;
    return v;
}

// linear tile index -> tile origin (same orders as pdist_mfma.hip: row-major upper triangle / 16-row groups)
template <bool SYM>
__device__ __forceinline__ void ws_tile_coords(int64_t t, int tiles_m, int tiles_n, int64_t &m0, int64_t &n0)
{
    if (SYM) {
        const double T = (double)tiles_n;
        int64_t tm = (int64_t)(((2.0 * T + 1.0) - sqrt((2.0 * T + 1.0) * (2.0 * T + 1.0) - 8.0 * (double)t)) * 0.5);
        if (tm < 0) tm = 0;
        if (tm > tiles_m - 1) tm = tiles_m - 1;
        while (tm > 0 && tm * (int64_t)tiles_n - tm * (tm - 1) / 2 > t) tm--;
        while ((tm + 1) * (int64_t)tiles_n - (tm + 1) * tm / 2 <= t) tm++;
        const int64_t off = tm * (int64_t)tiles_n - tm * (tm - 1) / 2;
        m0 = tm * WS_BM;
        n0 = (tm + (t - off)) * WS_BN;
        return;
    }
    constexpr int GROUP_M = 16;
    const int64_t per_group = (int64_t)GROUP_M * tiles_n;
    const int64_t group = t / per_group, in_g = t % per_group;
    const int64_t first_m = group * GROUP_M;
    const int64_t gsz = (tiles_m - first_m < GROUP_M) ? (tiles_m - first_m) : GROUP_M;
    m0 = (first_m + in_g % gsz) * WS_BM;
    n0 = (in_g / gsz) * WS_BN;
}

// mover threads: global -> registers, chunk [k0, k0 + klen) of rows [row0, row0 + 128).  Unconditional loads from
// clamped addresses (a branch around a load serialises them); invalid rows / k are zeroed at the LDS store.
__device__ __forceinline__ void ws_fetch(float4 (&v)[2], const float *__restrict__ src, uint32_t ld, int64_t row0, int64_t nrows,
                                         int k0, int mt)
{
    const float *base = src + row0 * (int64_t)ld;
    const int rows_here = (int)((nrows - row0 < WS_BM) ? (nrows - row0) : WS_BM);
    const int r0 = mt >> 3, kq = (mt & 7) * 4;                 // 8 pieces of 16 bytes per operand row of a chunk
    const int kmax = (int)ld - 4 - k0;                         // a float4 at k <= ld - 4 never leaves its row
    const uint32_t kc = (uint32_t)(k0 + (kq < kmax ? kq : kmax));
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int r = it * 64 + r0;
        const int rc = r < rows_here ? r : rows_here - 1;
        v[it] = *(const float4 *)(base + ((uint32_t)rc * ld + kc));
    }
}

// mover threads: registers -> LDS operand buffer, even k to [0, 16), odd k to [16, 32) of each row
__device__ __forceinline__ void ws_put(float *lds, const float4 (&v)[2], int rows_here, int klen, int mt)
{
    const int r0 = mt >> 3, kq = (mt & 7) * 4;
    const int nvalid = klen - kq;
    float *o = lds + r0 * WS_LD + (kq >> 1);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const bool rok = (it * 64 + r0) < rows_here;
        *(float2 *)(o + it * 64 * WS_LD) = make_float2(ws_mask(v[it].x, rok && nvalid > 0), ws_mask(v[it].z, rok && nvalid > 2));
        *(float2 *)(o + it * 64 * WS_LD + WS_BK / 2) = make_float2(ws_mask(v[it].y, rok && nvalid > 1), ws_mask(v[it].w, rok && nvalid > 3));
    }
}

// mover threads: stage [128][WS_SP] -> global rows, 16 bytes per thread, 32 lanes per 512-byte row segment
__device__ __forceinline__ void ws_stream(const float *stage, float *gbase, uint32_t ldo, int nrows, int ncols, int mt)
{
    const int r0 = mt >> 5, c4 = (mt & 31) * 4;
    char *gb = (char *)gbase;
    const uint32_t ldo4 = ldo * 4u;
    const bool fast = (nrows >= WS_BM) && (ncols >= WS_BN);
#pragma unroll
    for (int p = 0; p < WS_BM / 16; p++) {
        const int row = p * 16 + r0;
        const float4 v = *(const float4 *)&stage[row * WS_SP + c4];
        float *dp = (float *)(gb + ((uint32_t)row * ldo4 + (uint32_t)c4 * 4u));
        if (fast) {
            __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, (f32x4 *)dp);
        } else if (row < nrows) {
            if (c4 < ncols) dp[0] = v.x;
            if (c4 + 1 < ncols) dp[1] = v.y;
            if (c4 + 2 < ncols) dp[2] = v.z;
            if (c4 + 3 < ncols) dp[3] = v.w;
        }
    }
}

struct WsChunks {
    int n;                       // chunks per tile (>= 4)
    int k0[WS_MAX_CHUNKS + 1];   // chunk c covers [k0[c], k0[c+1])
};

template <int METRIC, bool SYM>
__global__ __launch_bounds__(WS_THREADS, 4) void pdist_ws_kernel(const float *__restrict__ A, uint32_t lda, const float *__restrict__ Bm,
                                                                uint32_t ldb, const float *__restrict__ sqa, const float *__restrict__ sqb,
                                                                int64_t Q, int64_t N, WsChunks ch, float *__restrict__ out, uint32_t ldo,
                                                                int tiles_m, int tiles_n, int64_t ntiles, unsigned long long *prof)
{
    extern __shared__ __attribute__((aligned(16))) float ws_smem[];
    // tuning aid (SE_PD_PROFILE=1): shader-clock cycles per phase of matrix wave 0 (slots 0-3) and mover wave 8 (slots 4-7)
    uint64_t t_acc[4] = {0, 0, 0, 0}, t_last = prof ? __builtin_amdgcn_s_memtime() : 0;
#define WS_T(i) if (prof) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_acc[i] += now - t_last; t_last = now; }
    float *opA = ws_smem;                    // [2][128][36]
    float *opB = ws_smem + 2 * WS_OP;        // [2][128][36]
    float *stage = ws_smem + 4 * WS_OP;      // [128][132]

    // ---- this workgroup's tile list: XCD-contiguous band, round-robin inside the XCD ----
    const int64_t b = blockIdx.x, G = gridDim.x;
    const int64_t xcd = b & 7, qq = ntiles >> 3, rr = ntiles & 7;
    const int64_t band_beg = (xcd < rr) ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq;
    const int64_t band_len = qq + (xcd < rr ? 1 : 0);
    const int64_t wg_in_xcd = b >> 3, wgs_per_xcd = (G + 7 - xcd) >> 3;
    const int64_t my_tiles = (band_len > wg_in_xcd) ? (band_len - wg_in_xcd + wgs_per_xcd - 1) / wgs_per_xcd : 0;
    if (my_tiles == 0) return;
    const int nch = ch.n;
    const int64_t total = my_tiles * nch;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool mover = wave >= 8;

    if (mover) {
        // ================================ mover waves ================================
        const int mt = tid - WS_MOVERS;
        // Two register sets: the chunk fetched in iteration it is written to LDS in iteration it + 1 and consumed by the
        // matrix waves in iteration it + 2, so every global load has a whole chunk of MFMA time (and the store burst
        // that is queued in front of it) to land.  With a one-iteration lead the movers were load-latency-bound and the
        // matrix waves waited for them at every barrier.
        float4 raA[2], rbA[2], raB[2], rbB[2];
        int gA_ra = 0, gA_rb = 0, gA_kl = 0, gB_ra = 0, gB_rb = 0, gB_kl = 0;   // rows of A / rows of B / chunk length of each set
        int64_t fm0, fn0;               // tile of the item being fetched
        int fc = 0;                     // its chunk
        int64_t ft = 0;                 // its tile ordinal
        ws_tile_coords<SYM>(band_beg + wg_in_xcd, tiles_m, tiles_n, fm0, fn0);
#define WS_FETCH(S)                                                                                   \
    {                                                                                                 \
        ws_fetch(ra##S, A, lda, fm0, Q, ch.k0[fc], mt);                                               \
        ws_fetch(rb##S, Bm, ldb, fn0, N, ch.k0[fc], mt);                                              \
        g##S##_ra = (int)((Q - fm0 < WS_BM) ? (Q - fm0) : WS_BM);                                     \
        g##S##_rb = (int)((N - fn0 < WS_BN) ? (N - fn0) : WS_BN);                                     \
        g##S##_kl = ch.k0[fc + 1] - ch.k0[fc];                                                        \
        if (++fc == nch) {                                                                            \
            fc = 0;                                                                                   \
            ft++;                                                                                     \
            if (ft < my_tiles) ws_tile_coords<SYM>(band_beg + wg_in_xcd + ft * wgs_per_xcd, tiles_m, tiles_n, fm0, fn0); \
        }                                                                                             \
    }
#define WS_PUT(S, BUF)                                                        \
    {                                                                         \
        ws_put(opA + (BUF) * WS_OP, ra##S, g##S##_ra, g##S##_kl, mt);         \
        ws_put(opB + (BUF) * WS_OP, rb##S, g##S##_rb, g##S##_kl, mt);         \
    }
        // prologue: item 0 -> set A -> buffer 0; item 1 -> set B (stays in registers)
        WS_FETCH(A)
        WS_PUT(A, 0)
        if (total > 1) WS_FETCH(B)
        __syncthreads();
        int64_t pm0 = 0, pn0 = 0;       // previous tile (whose result sits in / goes through the stage)
        int c = 0;                      // chunk of the item the matrix waves are working on
        // one iteration: fetch item it + 2 into FETCH, stream the previous tile, write item it + 1 (set PUT) to LDS
#define WS_MOVER_ITER(PUT, FETCH)                                                                                        \
    {                                                                                                                    \
        if (it + 2 < total) WS_FETCH(FETCH)                                                                              \
        WS_T(0)                                                                                                          \
        if (it >= (int64_t)nch) {                                                                                        \
            if (c == 0) {                                                                                                \
                ws_tile_coords<SYM>(band_beg + wg_in_xcd + (it / nch - 1) * wgs_per_xcd, tiles_m, tiles_n, pm0, pn0);    \
                ws_stream(stage, out + (pm0 * (int64_t)ldo + pn0), ldo, (int)((Q - pm0 < WS_BM) ? (Q - pm0) : WS_BM),    \
                          (int)((N - pn0 < WS_BN) ? (N - pn0) : WS_BN), mt);                                             \
            } else if (SYM && c == 2 && pm0 != pn0) {                                                                    \
                ws_stream(stage, out + (pn0 * (int64_t)ldo + pm0), ldo, (int)((N - pn0 < WS_BN) ? (N - pn0) : WS_BN),    \
                          (int)((Q - pm0 < WS_BM) ? (Q - pm0) : WS_BM), mt);                                             \
            }                                                                                                            \
        }                                                                                                                \
        WS_T(1)                                                                                                          \
        if (it + 1 < total) WS_PUT(PUT, (int)((it + 1) & 1))                                                             \
        if (++c == nch) c = 0;                                                                                           \
        WS_T(2)                                                                                                          \
        __syncthreads();                                                                                                 \
        WS_T(3)                                                                                                          \
    }
#pragma unroll 1
        for (int64_t it = 0; it < total; it++) {
            WS_MOVER_ITER(B, A)
            if (++it >= total) break;
            WS_MOVER_ITER(A, B)
        }
#undef WS_MOVER_ITER
#undef WS_PUT
#undef WS_FETCH
        int64_t m0, n0;
        ws_tile_coords<SYM>(band_beg + wg_in_xcd + (my_tiles - 1) * wgs_per_xcd, tiles_m, tiles_n, m0, n0);
        // ---- drain: the last tile (m0, n0) was staged during the final iteration ----
        pm0 = m0; pn0 = n0;
        ws_stream(stage, out + (pm0 * (int64_t)ldo + pn0), ldo, (int)((Q - pm0 < WS_BM) ? (Q - pm0) : WS_BM),
                  (int)((N - pn0 < WS_BN) ? (N - pn0) : WS_BN), mt);
        if (SYM) {
            __syncthreads();   // matrix waves: stage transposed
            __syncthreads();
            if (pm0 != pn0)
                ws_stream(stage, out + (pn0 * (int64_t)ldo + pm0), ldo, (int)((N - pn0 < WS_BN) ? (N - pn0) : WS_BN),
                          (int)((Q - pm0 < WS_BM) ? (Q - pm0) : WS_BM), mt);
        }
        if (prof && tid == WS_MOVERS)
            for (int i = 0; i < 4; i++) atomicAdd(&prof[4 + i], (unsigned long long)t_acc[i]);
        return;
    }

    // ================================ matrix waves ================================
    const int wm = wave >> 1, wn = wave & 1;
    const int col = lane & 31, hi = lane >> 5;
    const int lr0 = wm * 32 + 4 * hi;                  // + (r & 3) + 8 * (r >> 2)
    f32x16 acc[2], sv[2];                              // sv: the previous tile's values, kept for its transposed staging
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[j][r] = 0.f; sv[j][r] = 0.f; }
    bool sv_mirror = false;                            // previous tile was off-diagonal (symmetric mode)
    int64_t m0, n0;
    ws_tile_coords<SYM>(band_beg + wg_in_xcd, tiles_m, tiles_n, m0, n0);
    __syncthreads();                                   // prologue barrier: item 0 is in buffer 0
#pragma unroll 1
    for (int64_t it = 0; it < total; it++) {
        const int c = (int)(it % nch);
        const int cb = (int)(it & 1);
        const float *pa = opA + cb * WS_OP + (wm * 32 + col) * WS_LD + hi * (WS_BK / 2);
        const float *pb0 = opB + cb * WS_OP + (wn * 64 + col) * WS_LD + hi * (WS_BK / 2);
        const float *pb1 = pb0 + 32 * WS_LD;
        // ---- transposed staging of the previous tile (its plain form was streamed out during chunk 0) ----
        if (SYM && c == 1 && it >= (int64_t)nch && sv_mirror) {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    *(float4 *)&stage[(wn * 64 + j * 32 + col) * WS_SP + lr0 + 8 * g] =
                        make_float4(sv[j][4 * g], sv[j][4 * g + 1], sv[j][4 * g + 2], sv[j][4 * g + 3]);
        }
        WS_T(0)
        // ---- MFMA over the chunk: 2 k per step, 4 steps per 16-byte operand read ----
        const int steps = (ch.k0[c + 1] - ch.k0[c] + 1) >> 1;
        const int full = steps & ~3;
#define WS_STEP(C)                                                                 \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.C, b0.C, acc[0], 0, 0, 0);    \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.C, b1.C, acc[1], 0, 0, 0);
        // Operand reads of group g + 1 are issued before the MFMAs of group g (two named register sets, explicit
        // fences): with only two matrix waves per SIMD the LDS round trip is otherwise exposed once per 8 MFMAs.
        // Reading one group past the end stays inside the padded LDS row (pitch 36, 16 + 16 de-interleaved floats).
#define WS_LOAD(S, G)                                    \
    a##S = *(const float4 *)(pa + 4 * (G));              \
    b0##S = *(const float4 *)(pb0 + 4 * (G));            \
    b1##S = *(const float4 *)(pb1 + 4 * (G));            \
    asm volatile("" ::: "memory");                       \
    __builtin_amdgcn_sched_barrier(0);
#define WS_RUN4(S)                                       \
    {                                                    \
        const float4 a4 = a##S, b0 = b0##S, b1 = b1##S;  \
        WS_STEP(x) WS_STEP(y) WS_STEP(z) WS_STEP(w)      \
    }                                                    \
    __builtin_amdgcn_sched_barrier(0);
#define WS_RUNP(S, CNT)                                  \
    {                                                    \
        const float4 a4 = a##S, b0 = b0##S, b1 = b1##S;  \
        WS_STEP(x)                                       \
        if ((CNT) > 1) { WS_STEP(y) }                    \
        if ((CNT) > 2) { WS_STEP(z) }                    \
    }
        {
            const int ngf = steps >> 2, rem = steps & 3;
            float4 aA, b0A, b1A, aB, b0B, b1B;
            int g = 0;
            WS_LOAD(A, 0)
            for (; g + 2 <= ngf; g += 2) {
                WS_LOAD(B, g + 1)
                WS_RUN4(A)
                WS_LOAD(A, g + 2)
                WS_RUN4(B)
            }
            if (g < ngf) {
                WS_LOAD(B, g + 1)
                WS_RUN4(A)
                if (rem) WS_RUNP(B, rem)
            } else if (rem) {
                WS_RUNP(A, rem)
            }
        }
#undef WS_LOAD
#undef WS_RUN4
#undef WS_RUNP
#undef WS_STEP
        WS_T(1)
        // ---- tile finished: finish values, keep them for the transposed staging, stage the plain tile ----
        if (c + 1 == nch) {
            const int rows_here = (int)((Q - m0 < WS_BM) ? (Q - m0) : WS_BM);
            const int cols_here = (int)((N - n0 < WS_BN) ? (N - n0) : WS_BN);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int lc = wn * 64 + j * 32 + col;
                float sb = 0.f;
                if (METRIC == SE_METRIC_EUCLID) sb = sqb[n0 + (lc < cols_here ? lc : cols_here - 1)];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int lr = lr0 + (r & 3) + 8 * (r >> 2);
                    float sa = 0.f;
                    if (METRIC == SE_METRIC_EUCLID) sa = sqa[m0 + (lr < rows_here ? lr : rows_here - 1)];
                    const float v = ws_finish<METRIC>(acc[j][r], sa, sb);
                    sv[j][r] = v;
                    stage[lr * WS_SP + lc] = v;
                    acc[j][r] = 0.f;
                }
            }
            sv_mirror = SYM && (m0 != n0);
            if (it + 1 < total) ws_tile_coords<SYM>(band_beg + wg_in_xcd + ((it + 1) / nch) * wgs_per_xcd, tiles_m, tiles_n, m0, n0);
        }
        WS_T(2)
        __syncthreads();
        WS_T(3)
    }
    // ---- drain: transposed staging of the last tile ----
    if (SYM) {
        __syncthreads();   // movers streamed the plain last tile
        if (sv_mirror) {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    *(float4 *)&stage[(wn * 64 + j * 32 + col) * WS_SP + lr0 + 8 * g] =
                        make_float4(sv[j][4 * g], sv[j][4 * g + 1], sv[j][4 * g + 2], sv[j][4 * g + 3]);
        }
        __syncthreads();
    }
    if (prof && tid == 0)
        for (int i = 0; i < 4; i++) atomicAdd(&prof[i], (unsigned long long)t_acc[i]);
#undef WS_T
}

static int ws_num_cus()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int METRIC, bool SYM>
static int launch_ws(const float *a, int64_t lda, const float *b, int64_t ldb, const float *sqa, const float *sqb, int64_t q, int64_t n,
                     const WsChunks &ch, float *out, int64_t ldo, hipStream_t s)
{
    const int tiles_m = (int)((q + WS_BM - 1) / WS_BM), tiles_n = (int)((n + WS_BN - 1) / WS_BN);
    const int64_t ntiles = SYM ? ((int64_t)tiles_n * (tiles_n + 1) / 2) : ((int64_t)tiles_m * tiles_n);
    const size_t lds = (size_t)(4 * WS_OP + WS_BM * WS_SP) * sizeof(float);
    int64_t grid = ws_num_cus() / 8 * 8;
    if (grid > ntiles) grid = ntiles;
    if (grid < 1) grid = 1;
    auto kern = pdist_ws_kernel<METRIC, SYM>;
    SE_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    static const bool profile = getenv("SE_PD_PROFILE") != nullptr;   // tuning aid only: allocates, synchronises, prints
    unsigned long long *prof = nullptr;
    if (profile) {
        SE_HIP_CHECK(hipMalloc((void **)&prof, 8 * sizeof(unsigned long long)));
        SE_HIP_CHECK(hipMemsetAsync(prof, 0, 8 * sizeof(unsigned long long), s));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WS_THREADS), lds, s, a, (uint32_t)lda, b, (uint32_t)ldb, sqa, sqb, q, n, ch, out,
                       (uint32_t)ldo, tiles_m, tiles_n, ntiles, prof);
    SE_LAUNCH_CHECK();
    if (profile) {
        unsigned long long h[8];
        SE_HIP_CHECK(hipStreamSynchronize(s));
        SE_HIP_CHECK(hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost));
        SE_HIP_CHECK(hipFree(prof));
        const double tm = (double)(h[0] + h[1] + h[2] + h[3]), tv = (double)(h[4] + h[5] + h[6] + h[7]);
        fprintf(stderr, "[pdist_ws profile] sym=%d chunks=%d tiles=%lld | matrix wave: stage-T %.1f%% mfma %.1f%% stage %.1f%% barrier %.1f%% (%.0f cyc/tile) | "
                        "mover wave: fetch %.1f%% stream %.1f%% put %.1f%% barrier %.1f%% (%.0f cyc/tile)\n", (int)SYM, ch.n, (long long)ntiles,
                100 * h[0] / tm, 100 * h[1] / tm, 100 * h[2] / tm, 100 * h[3] / tm, tm / (double)ntiles,
                100 * h[4] / tv, 100 * h[5] / tv, 100 * h[6] / tv, 100 * h[7] / tv, tv / (double)ntiles);
    }
    return SE_OK;
}

// Returns SE_OK when the wave-specialised kernel took the job, 1 when the caller should use the base kernel.
int pdist_ws_try(const float *a, int64_t lda, const float *b, int64_t ldb, const float *sqa, const float *sqb, int64_t q, int64_t n,
                 int64_t d, int metric, float *out, int64_t ldo, hipStream_t s)
{
    // OPT-IN (SE_PD_WS=1; the tests run it that way in a subprocess): measured 3.8 ms vs 3.5 ms for the base kernel on
    // 50k x 50k x 100.  Its phase profile (SE_PD_PROFILE=1) shows why specialisation does not pay here: the matrix waves
    // do saturate the pipe inside a chunk, but the mover waves spend 19k of 28k cycles per tile blocked behind their own
    // nontemporal stores -- the tile stream itself runs at ~3.5 TB/s next to the operand traffic, not at the 5.2 TB/s of
    // the isolated store probe -- and the matrix waves wait for them at every chunk barrier.
    static const bool on = getenv("SE_PD_WS") != nullptr;
    if (!on) return 1;
    if (d < 16 || d > 32 * WS_MAX_CHUNKS) return 1;
    if ((lda % 4) || (ldb % 4) || (ldo % 4) || (((uintptr_t)a) & 15) || (((uintptr_t)b) & 15) || (((uintptr_t)out) & 15)) return 1;
    if (q < 4 * WS_BM || n < 4 * WS_BN) return 1;            // small problems: the base kernel's 2 x 256 workgroups fill the chip better
    WsChunks ch;
    ch.n = (int)((d + WS_BK - 1) / WS_BK);
    if (ch.n < 4) ch.n = 4;
    int len = (int)((d + ch.n - 1) / ch.n);
    len = (len + 3) / 4 * 4;                                  // chunk starts stay multiples of 4 (16-byte loads)
    ch.n = (int)((d + len - 1) / len);
    if (ch.n < 4) return 1;
    for (int c = 0; c <= ch.n; c++) ch.k0[c] = (c * len < d) ? c * len : (int)d;
    const bool sym = (a == b) && (lda == ldb) && (q == n) && (metric != SE_METRIC_EUCLID || sqa == sqb);
    switch (metric) {
        case SE_METRIC_COSINE:
            return sym ? launch_ws<SE_METRIC_COSINE, true>(a, lda, b, ldb, sqa, sqb, q, n, ch, out, ldo, s)
                       : launch_ws<SE_METRIC_COSINE, false>(a, lda, b, ldb, sqa, sqb, q, n, ch, out, ldo, s);
        case SE_METRIC_EUCLID:
            return sym ? launch_ws<SE_METRIC_EUCLID, true>(a, lda, b, ldb, sqa, sqb, q, n, ch, out, ldo, s)
                       : launch_ws<SE_METRIC_EUCLID, false>(a, lda, b, ldb, sqa, sqb, q, n, ch, out, ldo, s);
        case SE_METRIC_DOT:
            return sym ? launch_ws<SE_METRIC_DOT, true>(a, lda, b, ldb, sqa, sqb, q, n, ch, out, ldo, s)
                       : launch_ws<SE_METRIC_DOT, false>(a, lda, b, ldb, sqa, sqb, q, n, ch, out, ldo, s);
        default: return 1;
    }
}

}  // namespace se
