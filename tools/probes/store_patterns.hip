// Probe: achievable HBM write bandwidth for different store patterns into a [N, N] float matrix
// (N = 50000, row pitch 200000 B).  hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o store_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// (a) linear fill, float4 per lane, grid-stride
__global__ void fill_linear(float4 *p, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// tile walkers: persistent WGs (grid = 512), 512 threads, 128x128 tiles, tile list split in 8 XCD bands
__device__ inline bool next_tile(int64_t it, int64_t ntiles, int tiles_n, int64_t &m0, int64_t &n0) {
    const int64_t b = blockIdx.x, G = gridDim.x, xcd = b & 7, qq = ntiles >> 3, rr = ntiles & 7;
    const int64_t beg = (xcd < rr) ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq, len = qq + (xcd < rr ? 1 : 0);
    const int64_t t = (b >> 3) + it * (G >> 3);
    if (t >= len) return false;
    const int64_t g = beg + t;
    m0 = (g / tiles_n) * 128; n0 = (g % tiles_n) * 128;
    return true;
}

// (b) accumulator-layout dword stores: wave (wm, wn) 32 x 64, per instr 2 rows x 128 B
__global__ __launch_bounds__(512) void tile_dword(float *out, int64_t N, int tiles_n, int64_t ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1, col = lane & 31, hi = lane >> 5;
    int64_t m0, n0;
    for (int64_t it = 0; next_tile(it, ntiles, tiles_n, m0, n0); it++) {
        if (m0 + 128 > N || n0 + 128 > N) continue;
        char *base = (char *)(out + m0 * N + n0);
        const uint32_t ld4 = (uint32_t)N * 4u;
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t lr = wm * 32 + 4 * hi + (r & 3) + 8 * (r >> 2), lc = wn * 64 + j * 32 + col;
                *(float *)(base + lr * ld4 + lc * 4) = (float)r;
            }
    }
}

// (c) row-major float4: thread t writes 16 B, 32 lanes cover one 512 B row segment (LDS-transposed epilogue equivalent)
__global__ __launch_bounds__(512) void tile_rows_f4(float *out, int64_t N, int tiles_n, int64_t ntiles) {
    const int t = threadIdx.x;
    int64_t m0, n0;
    for (int64_t it = 0; next_tile(it, ntiles, tiles_n, m0, n0); it++) {
        if (m0 + 128 > N || n0 + 128 > N) continue;
        char *base = (char *)(out + m0 * N + n0);
        const uint32_t ld4 = (uint32_t)N * 4u;
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const uint32_t row = p * 16 + (t >> 5), c4 = (t & 31) * 16;
            *(float4 *)(base + row * ld4 + c4) = make_float4(1.f, 2.f, 3.f, (float)p);
        }
    }
}

// (d) quad-transposed float4: per instr 8 rows x 128 B
__global__ __launch_bounds__(512) void tile_quad_f4(float *out, int64_t N, int tiles_n, int64_t ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1, col = lane & 31, hi = lane >> 5;
    int64_t m0, n0;
    for (int64_t it = 0; next_tile(it, ntiles, tiles_n, m0, n0); it++) {
        if (m0 + 128 > N || n0 + 128 > N) continue;
        char *base = (char *)(out + m0 * N + n0);
        const uint32_t ld4 = (uint32_t)N * 4u;
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t lr = wm * 32 + 4 * hi + 8 * g + (lane & 3), lc = wn * 64 + j * 32 + (col & ~3);
                *(float4 *)(base + lr * ld4 + lc * 4) = make_float4(1.f, 2.f, 3.f, (float)g);
            }
    }
}

// (e) mirror-style float4: per instr 32 rows x 32 B
__global__ __launch_bounds__(512) void tile_mirror_f4(float *out, int64_t N, int tiles_n, int64_t ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1, col = lane & 31, hi = lane >> 5;
    int64_t m0, n0;
    for (int64_t it = 0; next_tile(it, ntiles, tiles_n, m0, n0); it++) {
        if (m0 + 128 > N || n0 + 128 > N) continue;
        char *base = (char *)(out + m0 * N + n0);
        const uint32_t ld4 = (uint32_t)N * 4u;
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t row = wn * 64 + j * 32 + col, c = wm * 32 + 4 * hi + 8 * g;
                *(float4 *)(base + row * ld4 + c * 4) = make_float4(1.f, 2.f, 3.f, (float)g);
            }
    }
}

// (f) row-major float4 over 32 x 512 tiles (2 KB contiguous per row)
__global__ __launch_bounds__(512) void tile_wide_rows(float *out, int64_t N) {
    const int t = threadIdx.x;
    const int tiles_n = (int)(N / 512), tiles_m = (int)(N / 32);
    const int64_t ntiles = (int64_t)tiles_n * tiles_m;
    for (int64_t g = blockIdx.x; g < ntiles; g += gridDim.x) {
        const int64_t m0 = (g / tiles_n) * 32, n0 = (g % tiles_n) * 512;
        char *base = (char *)(out + m0 * N + n0);
        const uint32_t ld4 = (uint32_t)N * 4u;
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const uint32_t row = p * 4 + (t >> 7), c4 = (t & 127) * 16;
            *(float4 *)(base + row * ld4 + c4) = make_float4(1.f, 2.f, 3.f, (float)p);
        }
    }
}

int main() {
    const int64_t N = 50000;
    float *out;
    CK(hipMalloc(&out, N * N * 4));
    const int tiles_n = (int)((N + 127) / 128);
    const int64_t ntiles = (int64_t)tiles_n * tiles_n;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char *name, auto launch, double bytes) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("%-28s %8.3f ms  %8.1f GB/s\n", name, best, bytes / best / 1e6);
    };
    const double full = (double)N * N * 4, tiled = (double)(N / 128) * (N / 128) * 128 * 128 * 4;
    run("linear float4 fill", [&] { hipLaunchKernelGGL(fill_linear, dim3(2048), dim3(256), 0, 0, (float4 *)out, N * N / 4); }, full);
    run("tile dword (acc layout)", [&] { hipLaunchKernelGGL(tile_dword, dim3(512), dim3(512), 0, 0, out, N, tiles_n, ntiles); }, tiled);
    run("tile rows float4 (512B/row)", [&] { hipLaunchKernelGGL(tile_rows_f4, dim3(512), dim3(512), 0, 0, out, N, tiles_n, ntiles); }, tiled);
    run("tile quad float4 (8x128B)", [&] { hipLaunchKernelGGL(tile_quad_f4, dim3(512), dim3(512), 0, 0, out, N, tiles_n, ntiles); }, tiled);
    run("tile mirror float4 (32x32B)", [&] { hipLaunchKernelGGL(tile_mirror_f4, dim3(512), dim3(512), 0, 0, out, N, tiles_n, ntiles); }, tiled);
    run("32x512 tiles rows float4", [&] { hipLaunchKernelGGL(tile_wide_rows, dim3(2048), dim3(512), 0, 0, out, N); }, (double)(N / 32) * (N / 512) * 32 * 512 * 4);
    hipFree(out);
    return 0;
}
