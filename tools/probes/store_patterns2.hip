// Probe 2: what bounds tile-shaped HBM writes into a [N, N] float matrix (N = 50000)?
//   hipcc --offload-arch=gfx950 -O3 store_patterns2.hip -o store_patterns2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE> __device__ inline void st(float *p, f4 v)
{
    if (MODE == 0) *(f4 *)p = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, (f4 *)p);
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// generic tile writer: TM x TN tiles, rows written as contiguous float4 runs; ORDER 0 = XCD bands (each XCD a
// contiguous band of the tile list, its 64 WGs on adjacent tiles), 1 = chip-wide row-major (WG b takes tile it*G+b)
template <int TM, int TN, int MODE, int ORDER>
__global__ __launch_bounds__(512) void tile_rows(float *out, int64_t N)
{
    const int tiles_n = (int)(N / TN), tiles_m = (int)(N / TM);
    const int64_t ntiles = (int64_t)tiles_n * tiles_m;
    const int64_t b = blockIdx.x, G = gridDim.x;
    const int t = threadIdx.x;
    constexpr int LPR = TN / 4;              // lanes per row
    constexpr int RPP = 512 / LPR;           // rows per pass
    for (int64_t it = 0;; it++) {
        int64_t g;
        if (ORDER == 0) {
            const int64_t xcd = b & 7, qq = ntiles >> 3;
            const int64_t tt = (b >> 3) + it * (G >> 3);
            if (tt >= qq) break;
            g = xcd * qq + tt;
        } else {
            g = it * G + b;
            if (g >= ntiles) break;
        }
        const int64_t m0 = (g / tiles_n) * TM, n0 = (g % tiles_n) * TN;
        float *base = out + m0 * N + n0;
#pragma unroll
        for (int p = 0; p < TM / RPP; p++) {
            const int row = p * RPP + t / LPR, c = (t % LPR) * 4;
            st<MODE>(base + (int64_t)row * N + c, (f4){1.f, 2.f, 3.f, (float)p});
        }
    }
}


// accumulator-layout patterns of the MFMA kernel (wave (wm, wn) owns 32 x 64 of a 128 x 128 tile), XCD bands
template <int MODE> __device__ inline void st1(float *p, float v)
{
    if (MODE == 0) *p = v; else __builtin_nontemporal_store(v, p);
}
template <int PAT, int MODE>
__global__ __launch_bounds__(512) void tile_acc(float *out, int64_t N)
{
    const int tiles_n = (int)(N / 128), tiles_m = (int)(N / 128);
    const int64_t ntiles = (int64_t)tiles_n * tiles_m;
    const int64_t b = blockIdx.x, G = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1, col = lane & 31, hi = lane >> 5;
    for (int64_t it = 0;; it++) {
        const int64_t xcd = b & 7, qq = ntiles >> 3;
        const int64_t tt = (b >> 3) + it * (G >> 3);
        if (tt >= qq) break;
        const int64_t g = xcd * qq + tt;
        const int64_t m0 = (g / tiles_n) * 128, n0 = (g % tiles_n) * 128;
        float *base = out + m0 * N + n0;
        for (int j = 0; j < 2; j++) {
            if (PAT == 0) {          // dword, per instr 2 rows x 128 B
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int lr = wm * 32 + 4 * hi + (r & 3) + 8 * (r >> 2), lc = wn * 64 + j * 32 + col;
                    st1<MODE>(base + (int64_t)lr * N + lc, (float)r);
                }
            } else if (PAT == 1) {   // quad-transposed dwordx4, per instr 8 rows x 128 B
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int lr = wm * 32 + 4 * hi + 8 * gq + (lane & 3), lc = wn * 64 + j * 32 + (col & ~3);
                    st<MODE>(base + (int64_t)lr * N + lc, (f4){1.f, 2.f, 3.f, (float)gq});
                }
            } else if (PAT == 2) {   // mirror-style dwordx4, per instr 32 rows x 32 B
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int row = wn * 64 + j * 32 + col, c = wm * 32 + 4 * hi + 8 * gq;
                    st<MODE>(base + (int64_t)row * N + c, (f4){1.f, 2.f, 3.f, (float)gq});
                }
            } else {                 // mirror after a quad transpose: per instr 8 rows x 128 B
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int row = wn * 64 + j * 32 + (col & ~3) + gq, c = wm * 32 + 4 * hi + 8 * (col & 3);
                    st<MODE>(base + (int64_t)row * N + c, (f4){1.f, 2.f, 3.f, (float)gq});
                }
            }
        }
    }
}

// linear fill in 64 KB chunks per WG, chunk order scrambled (is per-WG contiguity enough?)
__global__ __launch_bounds__(512) void chunks_scrambled(float *out, int64_t nchunks)
{
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t cc = (c * 7919) % nchunks;     // 7919 prime, nchunks not a multiple
        f4 *p = (f4 *)(out + cc * 16384);
#pragma unroll
        for (int i = 0; i < 8; i++) p[i * 512 + threadIdx.x] = (f4){1.f, 2.f, 3.f, 4.f};
    }
}

__global__ void fill_linear(f4 *p, int64_t n4)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) p[i] = (f4){1.f, 2.f, 3.f, 4.f};
}

int main()
{
    const int64_t N = 50000;
    float *out;
    CK(hipMalloc(&out, N * N * 4 + (1 << 20)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char *name, auto launch, double bytes) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("%-52s %8.3f ms  %8.1f GB/s\n", name, best, bytes / best / 1e6);
    };
    const double full = (double)N * N * 4;
#define TB(TM, TN) ((double)(N / TM) * (N / TN) * TM * TN * 4)
    run("linear float4 fill, grid 2048x256", [&] { hipLaunchKernelGGL(fill_linear, dim3(2048), dim3(256), 0, 0, (f4 *)out, N * N / 4); }, full);
    run("linear float4 fill, grid 512x512", [&] { hipLaunchKernelGGL(fill_linear, dim3(512), dim3(512), 0, 0, (f4 *)out, N * N / 4); }, full);
    run("64 KB chunks, scrambled order", [&] { hipLaunchKernelGGL(chunks_scrambled, dim3(512), dim3(512), 0, 0, out, N * N / 16384); }, (double)(N * N / 16384) * 65536);
    run("128x128 rows f4, XCD bands, plain", [&] { hipLaunchKernelGGL((tile_rows<128, 128, 0, 0>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("128x128 rows f4, XCD bands, nt", [&] { hipLaunchKernelGGL((tile_rows<128, 128, 1, 0>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("128x128 rows f4, XCD bands, sc1", [&] { hipLaunchKernelGGL((tile_rows<128, 128, 2, 0>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("128x128 rows f4, XCD bands, sc0 sc1", [&] { hipLaunchKernelGGL((tile_rows<128, 128, 3, 0>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("128x128 rows f4, chip-wide row-major, plain", [&] { hipLaunchKernelGGL((tile_rows<128, 128, 0, 1>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("128x128 rows f4, chip-wide row-major, nt", [&] { hipLaunchKernelGGL((tile_rows<128, 128, 1, 1>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("64x256 rows f4, XCD bands, plain", [&] { hipLaunchKernelGGL((tile_rows<64, 256, 0, 0>), dim3(512), dim3(512), 0, 0, out, N); }, TB(64, 256));
    run("64x256 rows f4, chip-wide row-major, plain", [&] { hipLaunchKernelGGL((tile_rows<64, 256, 0, 1>), dim3(512), dim3(512), 0, 0, out, N); }, TB(64, 256));
    run("32x512 rows f4, chip-wide row-major, plain", [&] { hipLaunchKernelGGL((tile_rows<32, 512, 0, 1>), dim3(512), dim3(512), 0, 0, out, N); }, TB(32, 512));
    run("32x512 rows f4, chip-wide row-major, nt", [&] { hipLaunchKernelGGL((tile_rows<32, 512, 1, 1>), dim3(512), dim3(512), 0, 0, out, N); }, TB(32, 512));
    run("16x1024 rows f4, chip-wide row-major, plain", [&] { hipLaunchKernelGGL((tile_rows<16, 1024, 0, 1>), dim3(512), dim3(512), 0, 0, out, N); }, TB(16, 1024));
    run("8x2048 rows f4, chip-wide row-major, plain", [&] { hipLaunchKernelGGL((tile_rows<8, 2048, 0, 1>), dim3(512), dim3(512), 0, 0, out, N); }, TB(8, 2048));
    run("256x64 rows f4, XCD bands, plain", [&] { hipLaunchKernelGGL((tile_rows<256, 64, 0, 0>), dim3(512), dim3(512), 0, 0, out, N); }, TB(256, 64));
    run("128x128 rows f4, XCD bands, plain, grid 256", [&] { hipLaunchKernelGGL((tile_rows<128, 128, 0, 0>), dim3(256), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("128x128 rows f4, XCD bands, plain, grid 1024", [&] { hipLaunchKernelGGL((tile_rows<128, 128, 0, 0>), dim3(1024), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("acc dword 2x128B, plain", [&] { hipLaunchKernelGGL((tile_acc<0, 0>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("acc dword 2x128B, nt", [&] { hipLaunchKernelGGL((tile_acc<0, 1>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("acc quad f4 8x128B, plain", [&] { hipLaunchKernelGGL((tile_acc<1, 0>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("acc quad f4 8x128B, nt", [&] { hipLaunchKernelGGL((tile_acc<1, 1>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("mirror f4 32x32B, plain", [&] { hipLaunchKernelGGL((tile_acc<2, 0>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("mirror f4 32x32B, nt", [&] { hipLaunchKernelGGL((tile_acc<2, 1>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("mirror quad f4 8x128B, plain", [&] { hipLaunchKernelGGL((tile_acc<3, 0>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    run("mirror quad f4 8x128B, nt", [&] { hipLaunchKernelGGL((tile_acc<3, 1>), dim3(512), dim3(512), 0, 0, out, N); }, TB(128, 128));
    (void)hipFree(out);
    return 0;
}
