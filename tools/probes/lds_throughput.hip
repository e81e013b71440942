// LDS throughput probe for the ranking kernel's three LDS primitives on gfx950 (one workgroup per CU, like the kernel):
//   A  ds_add_rtn_u32 on wave-private packed counters (random word / conflict-free / one address)
//   W  ds_write_b16 scatter into a 100 KB exchange buffer (random halfword / linear)
//   R  ds_read_u16 (random / linear)
// Reports LDS-pipe cycles per wave instruction per CU = elapsed shader cycles / (instructions issued per CU), for
// 8, 12 and 16 waves per workgroup and 4 / 8 / 16 instructions in flight per wave.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/lds_throughput.hip -o /tmp/lds_throughput
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int XBUF_HALFWORDS = 50176;

template <int OP, int DEPTH>
__global__ void probe(uint32_t *out, unsigned long long *cyc, int iters, int pattern)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(raw) + wave * 1024;            // 4 KB of counters per wave
    uint16_t *xbuf = reinterpret_cast<uint16_t *>(reinterpret_cast<uint32_t *>(raw) + nw * 1024);
    for (int i = tid; i < nw * 1024; i += blockDim.x) reinterpret_cast<uint32_t *>(raw)[i] = 0;
    for (int i = tid; i < XBUF_HALFWORDS; i += blockDim.x) xbuf[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t cb = (uint32_t)(uintptr_t)cnt, xb = (uint32_t)(uintptr_t)xbuf;
    uint32_t rng = (uint32_t)(blockIdx.x * 9781 + tid * 6271 + 12345) | 1u;
    uint32_t acc = 0;
    // addresses are fixed per (lane, slot) so that the timed loop holds no VALU work: the conflict pattern of a wave
    // instruction is what matters, not fresh addresses every iteration
    uint32_t addr[DEPTH], val[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        rng = rng * 1664525u + 1013904223u;
        const uint32_t h = rng >> 8;
        if (OP == 0) {
            const uint32_t word = pattern == 0 ? (h & 1023u) : pattern == 1 ? (((uint32_t)lane + 64u * (h & 15u)) & 1023u) : 5u;
            addr[d] = cb + 4u * word;
            val[d] = (h & 0x2000u) ? 0x10000u : 1u;
        } else {
            const uint32_t hw = pattern == 0 ? (h % XBUF_HALFWORDS) : (uint32_t)((tid + blockDim.x * d) % XBUF_HALFWORDS);
            addr[d] = xb + 2u * hw;
            val[d] = h;
        }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        uint32_t r[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (OP == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(r[d]) : "v"(addr[d]), "v"(val[d]) : "memory");
            else if (OP == 1) { asm volatile("ds_write_b16 %0, %1" : : "v"(addr[d]), "v"(val[d]) : "memory"); r[d] = 0; }
            else asm volatile("ds_read_u16 %0, %1" : "=v"(r[d]) : "v"(addr[d]) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < DEPTH; d++) acc += r[d];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + tid] = acc + xbuf[tid];
}

template <int OP, int DEPTH>
static int run(const char *name, int waves, int pattern, const char *pname, uint32_t *out, unsigned long long *cyc)
{
    const int iters = 2048 / DEPTH, grid = 256, threads = waves * 64;
    const size_t lds = (size_t)waves * 4096 + XBUF_HALFWORDS * 2;
    CK(hipFuncSetAttribute((const void *)probe<OP, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<OP, DEPTH><<<grid, threads, lds>>>(out, cyc, iters, pattern);
    CK(hipEventRecord(e0));
    probe<OP, DEPTH><<<grid, threads, lds>>>(out, cyc, iters, pattern);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= grid;
    const double instr = (double)iters * DEPTH * waves;       // wave instructions per CU
    // s_memtime ticks at 100 MHz on gfx9; convert with the measured kernel time instead: cycles at the shader clock are
    // unknown, so report ns per wave instruction per CU and the equivalent cycles at 2.1 GHz
    const double ns = (double)ms * 1e6 / instr;
    printf("%-4s %-14s waves %2d depth %2d : %6.2f ns / wave-instr / CU  (= %5.1f cycles at 2.1 GHz)   [memtime %.0f ticks]\n", name, pname, waves, DEPTH, ns,
           ns * 2.1, mean);
    return 0;
}

int main()
{
    uint32_t *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    for (int waves : {8, 12, 16}) {
        if (run<0, 4>("A", waves, 0, "random word", out, cyc)) return 1;
        if (run<0, 8>("A", waves, 0, "random word", out, cyc)) return 1;
        if (run<0, 16>("A", waves, 0, "random word", out, cyc)) return 1;
        if (run<0, 8>("A", waves, 1, "conflict-free", out, cyc)) return 1;
        if (run<0, 8>("A", waves, 2, "one address", out, cyc)) return 1;
        if (run<1, 8>("W", waves, 0, "random b16", out, cyc)) return 1;
        if (run<1, 8>("W", waves, 1, "linear b16", out, cyc)) return 1;
        if (run<2, 8>("R", waves, 0, "random u16", out, cyc)) return 1;
        if (run<2, 8>("R", waves, 1, "linear u16", out, cyc)) return 1;
    }
    return 0;
}
