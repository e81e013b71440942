// glds_gemm.hip -- probe: the fp32-MFMA tile loop of pdist_mfma.hip with its operand staging replaced by LDS-DMA
// (global_load_lds_dwordx4 straight into a double-buffered, XOR-swizzled LDS image; K-chunks of 32; ONE barrier per chunk; no
// staging registers, no ds_write pass).  No epilogue: every tile's accumulators are folded into one checksum word per lane, so the
// number is the ceiling of the main loop.  Also checks a small problem against the sequential fmaf chain (bit-exact).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/glds_gemm.hip -o /tmp/glds_gemm && /tmp/glds_gemm
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int BM = 128, BN = 128, BK = 32, THREADS = 512;
constexpr int BUF_FLOATS = (BM + BN) * BK;          // one buffer: A image [128][32] then B image [128][32], 32 KB

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// one K-chunk of one tile: rows [m0, m0+128) of A and [n0, n0+128) of B, k in [k0, k0+32) -> buffer `buf`
__device__ __forceinline__ void issue_chunk(const float *A, int lda, const float *B, int ldb, int Q, int N, int D, int m0, int n0, int k0,
                                            float *buf, const float *zero_page, int wave, int lane)
{
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int row = 16 * wave + 8 * q + (lane >> 3);            // tile row written by this lane
        const int piece = (lane & 7) ^ ((row >> 1) & 7);            // which 16-byte piece of the row lands in this lane's slot (swizzle on the SOURCE)
        const int k = k0 + 4 * piece;
        int ga = m0 + row; ga = ga < Q ? ga : Q - 1;
        int gb = n0 + row; gb = gb < N ? gb : N - 1;
        const float *sa = (k < D) ? A + (int64_t)ga * lda + k : zero_page;
        const float *sb = (k < D) ? B + (int64_t)gb * ldb + k : zero_page;
        __builtin_amdgcn_global_load_lds((gptr_t *)sa, (lptr_t *)(buf + (16 * wave + 8 * q) * BK), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t *)sb, (lptr_t *)(buf + BM * BK + (16 * wave + 8 * q) * BK), 16, 0, 0);
    }
}

template <bool STORE>
__global__ __launch_bounds__(THREADS, 4) void glds_gemm_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb, int Q, int N,
                                                               int D, const float *__restrict__ zero_page, float *__restrict__ out, int ldo,
                                                               int tiles_m, int tiles_n, float *__restrict__ sink)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, col = lane & 31, hi = lane >> 5;
    const int nchunks = (D + BK - 1) / BK;
    const int ntiles = tiles_m * tiles_n;
    // the product kernel's tile order: block b stays on XCD b % 8, every XCD owns a contiguous band of the tile list (16 tile-rows deep
    // groups, column-major inside a group) and its workgroups sweep it together
    const int b = blockIdx.x, G = gridDim.x, xcd = b & 7, qq = ntiles >> 3, rr = ntiles & 7;
    const int band_beg = (xcd < rr) ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq;
    const int band_len = qq + (xcd < rr ? 1 : 0);
    const int wg_in_xcd = b >> 3, wgs_per_xcd = (G + 7 - xcd) >> 3;
    const int my_tiles = (band_len > wg_in_xcd) ? (band_len - wg_in_xcd + wgs_per_xcd - 1) / wgs_per_xcd : 0;
    if (my_tiles == 0) return;
    const int total = my_tiles * nchunks;

    f32x16 acc0, acc1;
    float fold = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }

    // swizzled read slots of this lane's three operand rows
    const int rowA = wm * 32 + col, rowB0 = wn * 64 + col, rowB1 = rowB0 + 32;
    const int swA = (rowA >> 1) & 7, swB0 = (rowB0 >> 1) & 7, swB1 = (rowB1 >> 1) & 7;

    int t = band_beg + wg_in_xcd, c = 0;               // (tile, chunk) of the NEXT chunk to request
    auto coords = [&](int tt, int &m0, int &n0) {
        const unsigned per_group = 16u * (unsigned)tiles_n, group = (unsigned)tt / per_group, in_g = (unsigned)tt - group * per_group;
        const unsigned first_m = group * 16u, gsz = ((unsigned)tiles_m - first_m < 16u) ? ((unsigned)tiles_m - first_m) : 16u;
        const unsigned col_t = in_g / gsz;
        m0 = (int)(first_m + (in_g - col_t * gsz)) * BM;
        n0 = (int)col_t * BN;
    };
    int m0, n0;
    coords(t, m0, n0);
    issue_chunk(A, lda, B, ldb, Q, N, D, m0, n0, 0, smem, zero_page, wave, lane);
    int cur_m0 = m0, cur_n0 = n0, cur_c = 0;
    c = 1;
    if (c == nchunks) { c = 0; t += wgs_per_xcd; }
#pragma unroll 1
    for (int it = 0; it < total; it++) {
        float *buf = smem + (it & 1) * BUF_FLOATS;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of chunk `it` have landed
        __builtin_amdgcn_s_barrier();                           // ... everyone's have, and everyone is done reading the other buffer
        int nm0 = 0, nn0 = 0, nc = c;
        if (it + 1 < total) {
            coords(t, nm0, nn0);
            issue_chunk(A, lda, B, ldb, Q, N, D, nm0, nn0, c * BK, smem + ((it + 1) & 1) * BUF_FLOATS, zero_page, wave, lane);
            c++;
            if (c == nchunks) { c = 0; t += wgs_per_xcd; }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int klen = (D - cur_c * BK < BK) ? (D - cur_c * BK) : BK;
        const int groups = (klen + 3) >> 2;
        const float *pa = buf + rowA * BK, *pb0 = buf + BM * BK + rowB0 * BK, *pb1 = buf + BM * BK + rowB1 * BK;
#pragma unroll 2
        for (int s = 0; s < groups; s++) {
            const float4 a4 = *(const float4 *)(pa + ((s ^ swA) << 2));
            const float4 b04 = *(const float4 *)(pb0 + ((s ^ swB0) << 2));
            const float4 b14 = *(const float4 *)(pb1 + ((s ^ swB1) << 2));
            const float ae = hi ? a4.y : a4.x, ao = hi ? a4.w : a4.z;
            const float b0e = hi ? b04.y : b04.x, b0o = hi ? b04.w : b04.z;
            const float b1e = hi ? b14.y : b14.x, b1o = hi ? b14.w : b14.z;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, b0e, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, b1e, acc1, 0, 0, 0);
            if (s * 4 + 2 < klen) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, b0o, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, b1o, acc1, 0, 0, 0);
            }
        }
        if (cur_c + 1 == nchunks) {                            // tile finished
            if (STORE) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int lr = wm * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                    const int gm = cur_m0 + lr, gn0 = cur_n0 + wn * 64 + col, gn1 = gn0 + 32;
                    if (gm < Q && gn0 < N) out[(int64_t)gm * ldo + gn0] = acc0[r];
                    if (gm < Q && gn1 < N) out[(int64_t)gm * ldo + gn1] = acc1[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) fold += acc0[r] + acc1[r];
            }
#pragma unroll
            for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
        }
        cur_c++;
        if (cur_c == nchunks) cur_c = 0;
        if (it + 1 < total && nc == 0) { cur_m0 = nm0; cur_n0 = nn0; }
    }
    if (!STORE) sink[blockIdx.x * THREADS + threadIdx.x] = fold;
}

int main()
{
    float *zero;
    CHECK(hipMalloc(&zero, 256));
    CHECK(hipMemset(zero, 0, 256));
    CHECK(hipFuncSetAttribute((const void *)glds_gemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_FLOATS * 4));
    CHECK(hipFuncSetAttribute((const void *)glds_gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_FLOATS * 4));
    // ---- correctness: 300 x 260 x D in {100, 64, 36, 1000} against the sequential fmaf chain ----
    for (int D : {100, 64, 36, 1000}) {
        const int Q = 300, N = 260, ld = D;
        std::vector<float> a((size_t)Q * ld), b((size_t)N * ld), want((size_t)Q * N), got((size_t)Q * N);
        srand(D);
        for (auto &v : a) v = (float)rand() / RAND_MAX - 0.5f;
        for (auto &v : b) v = (float)rand() / RAND_MAX - 0.5f;
        for (int i = 0; i < Q; i++)
            for (int j = 0; j < N; j++) {
                float s = 0.f;
                for (int k = 0; k < D; k++) s = fmaf(a[(size_t)i * ld + k], b[(size_t)j * ld + k], s);
                want[(size_t)i * N + j] = s;
            }
        float *da, *db, *dout, *sink;
        CHECK(hipMalloc(&da, a.size() * 4)); CHECK(hipMalloc(&db, b.size() * 4)); CHECK(hipMalloc(&dout, want.size() * 4)); CHECK(hipMalloc(&sink, 4));
        CHECK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        const int tm = (Q + BM - 1) / BM, tn = (N + BN - 1) / BN;
        hipLaunchKernelGGL(glds_gemm_kernel<true>, dim3(8), dim3(THREADS), 2 * BUF_FLOATS * 4, 0, da, ld, db, ld, Q, N, D, zero, dout, N, tm, tn, sink);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < got.size(); i++) bad += (memcmp(&got[i], &want[i], 4) != 0);
        printf("check D=%4d: %zu of %zu outputs differ from the fmaf chain\n", D, bad, got.size());
        hipFree(da); hipFree(db); hipFree(dout); hipFree(sink);
    }
    // ---- timing: main loop only ----
    struct Shape { int Q, N, D; } shapes[] = {{50000, 50000, 100}, {16384, 160146, 1000}};
    for (auto sh : shapes) {
        const int ld = sh.D;
        float *da, *db, *sink;
        CHECK(hipMalloc(&da, (size_t)sh.Q * ld * 4)); CHECK(hipMalloc(&db, (size_t)sh.N * ld * 4)); CHECK(hipMalloc(&sink, 512 * THREADS * 4));
        std::vector<float> h((size_t)1 << 20);
        for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
        for (size_t off = 0; off < (size_t)sh.Q * ld; off += h.size()) CHECK(hipMemcpy(da + off, h.data(), std::min(h.size(), (size_t)sh.Q * ld - off) * 4, hipMemcpyHostToDevice));
        for (size_t off = 0; off < (size_t)sh.N * ld; off += h.size()) CHECK(hipMemcpy(db + off, h.data(), std::min(h.size(), (size_t)sh.N * ld - off) * 4, hipMemcpyHostToDevice));
        const int tm = (sh.Q + BM - 1) / BM, tn = (sh.N + BN - 1) / BN;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(glds_gemm_kernel<false>, dim3(512), dim3(THREADS), 2 * BUF_FLOATS * 4, 0, da, ld, db, ld, sh.Q, sh.N, sh.D, zero, nullptr, 0, tm, tn, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("glds main loop %d x %d x %d: %.3f ms  %.1f TFLOP/s\n", sh.Q, sh.N, sh.D, ms, 2.0 * sh.Q * sh.N * sh.D / ms / 1e9);
        }
        hipFree(da); hipFree(db); hipFree(sink);
    }
    return 0;
}
