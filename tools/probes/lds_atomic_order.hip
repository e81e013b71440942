// Probe: in which order does one wave-wide ds_add_rtn_u32 serve lanes that hit the SAME LDS address?
// (A stable counting rank could use one returning atomic per key instead of 8 ballots if the order
// were always lane-ascending.  It is not architecturally promised, so this only gathers evidence.)
//   hipcc --offload-arch=gfx950 -O3 lds_atomic_order.hip -o lds_atomic_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ __launch_bounds__(512) void probe(const uint32_t *digits, int steps, uint32_t *ret)
{
    __shared__ uint32_t cnt[8][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 256; i += 64) cnt[wave][i] = 0;
    __syncthreads();
    for (int s = 0; s < steps; s++) {
        const size_t at = ((size_t)blockIdx.x * steps + s) * 512 + threadIdx.x;
        const uint32_t d = digits[at];
        ret[at] = atomicAdd(&cnt[wave][d], 1u);
    }
}

int main()
{
    const int blocks = 512, steps = 64;
    const size_t n = (size_t)blocks * steps * 512;
    std::vector<uint32_t> h(n), r(n);
    uint32_t seed = 12345;
    for (size_t i = 0; i < n; i++) {
        seed = seed * 1664525u + 1013904223u;
        const int mode = (int)((i / 512 / steps) % 4);     // per block: 0 all-same, 1 four values, 2 sixteen, 3 uniform 256
        const uint32_t rnd = seed >> 24;
        h[i] = mode == 0 ? 7u : mode == 1 ? (rnd & 3u) : mode == 2 ? (rnd & 15u) * 16u : rnd;
    }
    uint32_t *dd, *dr;
    hipMalloc(&dd, n * 4); hipMalloc(&dr, n * 4);
    hipMemcpy(dd, h.data(), n * 4, hipMemcpyHostToDevice);
    long bad_total = 0;
    for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, dd, steps, dr);
        hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int b = 0; b < blocks; b++)
            for (int w = 0; w < 8; w++) {
                uint32_t expect[256] = {0};
                for (int s = 0; s < steps; s++)
                    for (int l = 0; l < 64; l++) {
                        const size_t at = ((size_t)b * steps + s) * 512 + w * 64 + l;
                        if (r[at] != expect[h[at]]) bad++;
                        expect[h[at]]++;
                    }
            }
        printf("rep %d: %ld of %zu returns differ from the (step, lane)-ascending order\n", rep, bad, n);
        bad_total += bad;
    }
    printf("%s\n", bad_total == 0 ? "LANE-ASCENDING in every case observed" : "NOT lane-ascending");
    return 0;
}
