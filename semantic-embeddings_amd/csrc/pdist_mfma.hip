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
// Structure: persistent workgroups (2 per CU, 256 threads = 2 x 2 waves, 64 x 64 outputs per wave
// = 2 x 2 MFMA tiles, 64 accumulator VGPRs) walk 128 x 128 output tiles.  Operands are staged
// through LDS in K-chunks of 64 with even and odd k de-interleaved, so each lane feeds four MFMA
// steps from one 16-byte ds_read_b128; the row pitch is padded to 68 floats (conflict-free reads).
// The (tile, chunk) sequence is software-pipelined: while the MFMAs of chunk i run out of LDS, the
// global loads of chunk i+1 -- possibly the first chunk of the NEXT tile -- are already in flight
// into registers, and the accumulator stores of a finished tile overlap the next tile's MFMAs.
// Tile order: block b stays on XCD b % 8; every XCD owns a contiguous band of the tile space in
// "16 tile-rows deep" grouped order and its workgroups sweep it together, so a gallery panel is
// re-read from that XCD's private L2 instead of HBM.
//
// Roofline: 2*D flop per output element against 4 output bytes: at D = 100 fp32 (157 TFLOP/s peak)
// the kernel is MFMA-bound (3.2 ms for 50k x 50k) while the HBM bound is 1.25 ms.
#include "se_common.h"

namespace se {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PD_BM = 128, PD_BN = 128, PD_BK = 64;
constexpr int PD_LD = PD_BK + 4;  // padded LDS row pitch in floats
constexpr int PD_GROUP_M = 16;
constexpr int PD_MAX_KB = 16;
constexpr int PD_WGS_PER_CU = 2;

struct KBlocks {
    int n;
    int len[PD_MAX_KB];
};

struct Staged {            // one K-chunk of a tile in registers: 8 + 8 float4 per thread
    float4 a[8], b[8];
};

// Global -> registers: rows [row0, row0+128) x k [k0, k0+64), zero filled outside [0,nrows) x [0,kend).
// Every load is UNCONDITIONAL (clamped address + select): a branch around a load makes hipcc wait for
// each load before issuing the next one, which serialises 16 L2 round trips per chunk.
__device__ __forceinline__ void pd_load(float4 (&v)[8], const float *__restrict__ src, int64_t ld, int64_t row0,
                                        int64_t nrows, int64_t k0, int64_t kend, bool vec_ok)
{
    const int tid = threadIdx.x;
    if (vec_ok) {  // ld % 4 == 0, 16-byte aligned base, k0 % 4 == 0: a float4 at k < ld never leaves its row
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int idx = it * 256 + tid;
            const int r = idx >> 4, kq = (idx & 15) * 4;
            const int64_t gr = row0 + r, gk = k0 + kq;
            const int64_t grc = gr < nrows ? gr : nrows - 1;
            const int64_t gkc = gk < ld - 4 ? gk : ld - 4;
            const float4 x = *(const float4 *)(src + grc * ld + gkc);
            const bool rok = gr < nrows;
            v[it].x = (rok && gk + 0 < kend) ? x.x : 0.f;
            v[it].y = (rok && gk + 1 < kend) ? x.y : 0.f;
            v[it].z = (rok && gk + 2 < kend) ? x.z : 0.f;
            v[it].w = (rok && gk + 3 < kend) ? x.w : 0.f;
        }
    } else {
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int idx = it * 256 + tid;
            const int r = idx >> 4, kq = (idx & 15) * 4;
            const int64_t gr = row0 + r, gk = k0 + kq;
            const int64_t grc = gr < nrows ? gr : nrows - 1;
            const float *p = src + grc * ld;
            const bool rok = gr < nrows;
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int64_t kk = gk + j < kend ? gk + j : kend - 1;
                const float x = p[kk];
                e[j] = (rok && gk + j < kend) ? x : 0.f;
            }
            v[it] = make_float4(e[0], e[1], e[2], e[3]);
        }
    }
}

// Registers -> LDS, even k to [0,32), odd k to [32,64) of each row.
__device__ __forceinline__ void pd_store(float *lds, const float4 (&v)[8])
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int idx = it * 256 + tid;
        const int r = idx >> 4, kq = (idx & 15) * 4;
        float *o = lds + r * PD_LD + (kq >> 1);
        *(float2 *)o = make_float2(v[it].x, v[it].z);
        *(float2 *)(o + 32) = make_float2(v[it].y, v[it].w);
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

__device__ __forceinline__ void pd_tile_coords(int64_t t, int tiles_m, int tiles_n, int64_t &m0, int64_t &n0)
{
    const int64_t per_group = (int64_t)PD_GROUP_M * tiles_n;
    const int64_t group = t / per_group, in_g = t % per_group;
    const int64_t first_m = group * PD_GROUP_M;
    const int64_t gsz = (tiles_m - first_m < PD_GROUP_M) ? (tiles_m - first_m) : PD_GROUP_M;
    m0 = (first_m + in_g % gsz) * PD_BM;
    n0 = (in_g / gsz) * PD_BN;
}

template <int METRIC, bool MULTI_KB>
__global__ __launch_bounds__(256, 2) void pdist_kernel(
    const float *__restrict__ A, int64_t lda, const float *__restrict__ Bm, int64_t ldb,
    const float *__restrict__ sqa, const float *__restrict__ sqb, int64_t Q, int64_t N, int64_t D,
    KBlocks kbs, int nchunks, float *__restrict__ out, int64_t ldo, int tiles_m, int tiles_n, int vec_a, int vec_b)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sA = smem;
    float *sB = smem + PD_BM * PD_LD;

    // ---- this workgroup's tile list: XCD-contiguous band, round-robin inside the XCD ----
    const int64_t nblk = (int64_t)tiles_m * tiles_n;
    const int64_t b = blockIdx.x, G = gridDim.x;
    const int64_t xcd = b & 7, qq = nblk >> 3, rr = nblk & 7;
    const int64_t band_beg = (xcd < rr) ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq;
    const int64_t band_len = qq + (xcd < rr ? 1 : 0);
    const int64_t wg_in_xcd = b >> 3, wgs_per_xcd = (G + 7 - xcd) >> 3;  // blocks b' = xcd (mod 8), b' < G
    const int64_t my_tiles = (band_len > wg_in_xcd) ? (band_len - wg_in_xcd + wgs_per_xcd - 1) / wgs_per_xcd : 0;
    if (my_tiles == 0) return;

    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int col = lane & 31, hi = lane >> 5;

    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) { acc[i][j][r] = 0.f; if (MULTI_KB) tot[i][j][r] = 0.f; }

    const float *pa0 = sA + (wm * 64 + col) * PD_LD + hi * 32;
    const float *pa1 = pa0 + 32 * PD_LD;
    const float *pb0 = sB + (wn * 64 + col) * PD_LD + hi * 32;
    const float *pb1 = pb0 + 32 * PD_LD;

    // ---- prologue: fetch (tile 0, chunk 0) ----
    Staged st;
    int64_t m0, n0, k0, kend;
    bool closes_kb;
    pd_tile_coords(band_beg + wg_in_xcd, tiles_m, tiles_n, m0, n0);
    pd_chunk(0, kbs, MULTI_KB, D, k0, kend, closes_kb);
    pd_load(st.a, A, lda, m0, Q, k0, kend, vec_a && ((k0 & 3) == 0));
    pd_load(st.b, Bm, ldb, n0, N, k0, kend, vec_b && ((k0 & 3) == 0));
    bool first_kb = true;

    const int64_t total = my_tiles * nchunks;
#pragma unroll 1
    for (int64_t it = 0; it < total; it++) {
        const int c = (int)(it % nchunks);
        // current chunk geometry (what `st` holds)
        const int64_t cur_m0 = m0, cur_n0 = n0;
        const int kc = (int)((kend - k0 < PD_BK) ? (kend - k0) : PD_BK);
        const bool cur_closes = closes_kb;

        __syncthreads();  // LDS free: everyone finished the previous chunk's MFMAs
        pd_store(sA, st.a);
        pd_store(sB, st.b);
        __syncthreads();

        // ---- prefetch the next chunk (same tile or first chunk of the next tile) ----
        if (it + 1 < total) {
            const int nc = (c + 1 == nchunks) ? 0 : c + 1;
            if (nc == 0) pd_tile_coords(band_beg + wg_in_xcd + ((it + 1) / nchunks) * wgs_per_xcd, tiles_m, tiles_n, m0, n0);
            pd_chunk(nc, kbs, MULTI_KB, D, k0, kend, closes_kb);
            pd_load(st.a, A, lda, m0, Q, k0, kend, vec_a && ((k0 & 3) == 0));
            pd_load(st.b, Bm, ldb, n0, N, k0, kend, vec_b && ((k0 & 3) == 0));
        }

        // ---- MFMA over the chunk in LDS ----
        const int steps = (kc + 1) >> 1;
        for (int s = 0; s < steps; s += 4) {
            const float4 a0 = *(const float4 *)(pa0 + s);
            const float4 a1 = *(const float4 *)(pa1 + s);
            const float4 b0 = *(const float4 *)(pb0 + s);
            const float4 b1 = *(const float4 *)(pb1 + s);
#define PD_STEP(C)                                                                        \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.C, b0.C, acc[0][0], 0, 0, 0);     \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.C, b1.C, acc[0][1], 0, 0, 0);     \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.C, b0.C, acc[1][0], 0, 0, 0);     \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.C, b1.C, acc[1][1], 0, 0, 0);
            PD_STEP(x)
            if (s + 1 < steps) { PD_STEP(y) }
            if (s + 2 < steps) { PD_STEP(z) }
            if (s + 3 < steps) { PD_STEP(w) }
#undef PD_STEP
        }

        if (MULTI_KB && cur_closes) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        tot[i][j][r] = first_kb ? acc[i][j][r] : (tot[i][j][r] + acc[i][j][r]);
                        acc[i][j][r] = 0.f;
                    }
            first_kb = false;
        }

        // ---- tile finished: accumulators (col = lane & 31, row = (r&3) + 8*(r>>2) + 4*hi) -> global ----
        if (c + 1 == nchunks) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int64_t gc = cur_n0 + wn * 64 + j * 32 + col;
                float sb = 0.f;
                if (METRIC == SE_METRIC_EUCLID && gc < N) sb = sqb[gc];
#pragma unroll
                for (int i = 0; i < 2; i++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int64_t gr = cur_m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (gr < Q && gc < N) {
                            float v = MULTI_KB ? tot[i][j][r] : acc[i][j][r];
                            if (METRIC == SE_METRIC_COSINE) v = -v;
                            else if (METRIC == SE_METRIC_EUCLID) v = (sqa[gr] + sb) - 2.0f * v;
                            out[gr * ldo + gc] = v;
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
            first_kb = true;
        }
    }
}

static int pd_num_cus()
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

template <int METRIC>
static int launch_pdist(const float *a, int64_t lda, const float *b, int64_t ldb, const float *sqa,
                        const float *sqb, int64_t q, int64_t n, int64_t d, const KBlocks &kbs, bool multi,
                        float *out, int64_t ldo, hipStream_t s)
{
    const int tiles_m = (int)((q + PD_BM - 1) / PD_BM), tiles_n = (int)((n + PD_BN - 1) / PD_BN);
    const int64_t nblk = (int64_t)tiles_m * tiles_n;
    const size_t lds = (size_t)(PD_BM + PD_BN) * PD_LD * sizeof(float);
    const int vec_a = (lda % 4 == 0) && ((((uintptr_t)a) & 15) == 0);
    const int vec_b = (ldb % 4 == 0) && ((((uintptr_t)b) & 15) == 0);
    int nchunks = 0;
    for (int i = 0; i < kbs.n; i++) nchunks += (kbs.len[i] + PD_BK - 1) / PD_BK;
    int64_t grid = (int64_t)pd_num_cus() * PD_WGS_PER_CU;
    grid = grid / 8 * 8;
    if (grid > nblk) grid = nblk;
    if (grid < 1) grid = 1;
    if (multi) {
        auto kern = pdist_kernel<METRIC, true>;
        SE_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, a, lda, b, ldb, sqa, sqb, q, n, d, kbs, nchunks, out, ldo, tiles_m, tiles_n, vec_a, vec_b);
    } else {
        auto kern = pdist_kernel<METRIC, false>;
        SE_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, a, lda, b, ldb, sqa, sqb, q, n, d, kbs, nchunks, out, ldo, tiles_m, tiles_n, vec_a, vec_b);
    }
    SE_LAUNCH_CHECK();
    return SE_OK;
}

}  // namespace se

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
    if (d > 0x7FFFFFFFll) return fail(SE_ERR_UNSUPPORTED, "se_pairwise_dist: d too large");
    KBlocks kbs;
    kbs.n = 1;
    kbs.len[0] = (int)d;
    bool multi = false;
    if (kblocks && nkb > 1) {
        if (nkb > PD_MAX_KB) return fail(SE_ERR_UNSUPPORTED, "se_pairwise_dist: at most %d K-blocks", PD_MAX_KB);
        int64_t sum = 0;
        for (int i = 0; i < nkb; i++) {
            if (kblocks[i] <= 0) return fail(SE_ERR_INVALID, "se_pairwise_dist: K-block %d has length %d", i, kblocks[i]);
            kbs.len[i] = kblocks[i];
            sum += kblocks[i];
        }
        if (sum != d) return fail(SE_ERR_INVALID, "se_pairwise_dist: K-blocks sum to %lld, expected %lld", (long long)sum, (long long)d);
        kbs.n = nkb;
        multi = true;
    }
    hipStream_t s = (hipStream_t)stream;
    switch (metric) {
        case SE_METRIC_COSINE: return launch_pdist<SE_METRIC_COSINE>(a, lda, b, ldb, sqa, sqb, q, n, d, kbs, multi, out, ldo, s);
        case SE_METRIC_EUCLID: return launch_pdist<SE_METRIC_EUCLID>(a, lda, b, ldb, sqa, sqb, q, n, d, kbs, multi, out, ldo, s);
        case SE_METRIC_DOT: return launch_pdist<SE_METRIC_DOT>(a, lda, b, ldb, sqa, sqb, q, n, d, kbs, multi, out, ldo, s);
        default: return fail(SE_ERR_INVALID, "se_pairwise_dist: unknown metric %d", metric);
    }
}
