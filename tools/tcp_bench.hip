// tcp_bench.hip — what does a 16-byte-per-lane vector load cost the CU's L1 (TCP) by access pattern?
// The traversal microbenchmark (tools/trace_bench.hip) runs at the same speed whatever its VALU instruction count, and its time follows
// the number of vector loads: the per-CU load path is the limiter. This tool prices the patterns a BVH walk can choose between, from a
// table that stays in the CU's 32 KiB L1 (16 KiB, 128 lines): per wave-instruction, at 16 waves per CU,
//   0  coalesced      lane l reads chunk l % 8 of line (l / 8 + k): 8 whole lines per instruction (1 KiB contiguous)
//   1  one-per-lane   every lane reads one chunk of its own (random) line: what a node step's loads are today
//   2  pairs          lanes 2j, 2j+1 read adjacent chunks of one (random) line: 32 lines per instruction
//   3  quads          four lanes per line (64 contiguous bytes)
//   4  octets         eight lanes per line (a whole random line)
//   5  same           all lanes read the same 16 bytes
//   6  one-per-lane, 27 random lanes active (the traced kernels' lane utilisation)
// Prints shader cycles per wave-instruction per CU (s_memtime around the loop, summed over the CU's waves / instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static constexpr int ITER = 2048, UNROLL = 8;
typedef float v4f __attribute__((ext_vector_type(4)));

template <int P>
__global__ void __launch_bounds__(256) k(const v4f* __restrict__ table, const uint32_t* __restrict__ rnd, float* out, unsigned long long* ticks) {
    extern __shared__ float pad[];      // dynamic LDS: 40 KiB -> 4 workgroups (16 waves) per CU
    if (rnd[0] == 0xdeadbeefu) pad[threadIdx.x] = 1.0f;
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    v4f acc = {0, 0, 0, 0};
    bool active = true;
    if (P == 6) active = (rnd[(wave * 64u + lane) & 4095u] % 64u) < 27u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t s = rnd[(wave * 64u + lane) & 4095u];
    if (active)
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            s = s * 747796405u + 2891336453u;          // per-lane pseudo-random line (the same sequence for the lanes that must share one)
            const uint32_t r = s >> 20;
            uint32_t idx;      // in 16-byte chunks; table = 1024 chunks = 128 lines
            if (P == 0) idx = ((uint32_t)(it * UNROLL + u) * 64u + lane) & 1023u;
            else if (P == 1 || P == 6) idx = (r & 127u) * 8u + (lane & 7u);
            else if (P == 2) idx = (__shfl(r, lane & ~1u, 64) & 127u) * 8u + (lane & 1u) + (u & 3) * 2u;
            else if (P == 3) idx = (__shfl(r, lane & ~3u, 64) & 127u) * 8u + (lane & 3u) + (u & 1) * 4u;
            else if (P == 4) idx = (__shfl(r, lane & ~7u, 64) & 127u) * 8u + (lane & 7u);
            else idx = (uint32_t)(it + u) & 1023u;
            const v4f v = table[idx];
            acc += v;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (lane == 0) ticks[wave] = t1 - t0;
}

template <int P>
static void run(const char* name, int cus, const v4f* d_t, const uint32_t* d_r, float* d_o, unsigned long long* d_ticks) {
    const int grid = cus * 4;
    const size_t lds = 40 * 1024;
    CHECK(hipFuncSetAttribute((const void*)k<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<P>), dim3(grid), dim3(256), lds, 0, d_t, d_r, d_o, d_ticks);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<P>), dim3(grid), dim3(256), lds, 0, d_t, d_r, d_o, d_ticks);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> t((size_t)grid * 4);
    CHECK(hipMemcpy(t.data(), d_ticks, t.size() * 8, hipMemcpyDeviceToHost));
    std::sort(t.begin(), t.end());
    const double wave_cycles = (double)t[t.size() / 2];
    const double n = (double)ITER * UNROLL;
    // 16 waves per CU run concurrently: the CU retires 16 * n wave-instructions in `wave_cycles`
    printf("%-28s %8.1f us   a wave: %7.1f cycles per load   the CU: %6.1f cycles per wave-instruction\n", name, ms * 1e3, wave_cycles / n, wave_cycles / n / 16.0);
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    std::vector<float> tab(1024 * 4, 1.0f);
    std::vector<uint32_t> rnd(4096);
    uint32_t s = 1234567u;
    for (auto& r : rnd) { s = s * 1664525u + 1013904223u; r = s | 1u; }
    v4f* d_t; uint32_t* d_r; float* d_o; unsigned long long* d_ticks;
    CHECK(hipMalloc((void**)&d_t, tab.size() * 4)); CHECK(hipMemcpy(d_t, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc((void**)&d_r, rnd.size() * 4)); CHECK(hipMemcpy(d_r, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc((void**)&d_o, (size_t)cus * 4 * 256 * 4)); CHECK(hipMalloc((void**)&d_ticks, (size_t)cus * 16 * 8));
    run<0>("0 coalesced (8 lines)", cus, d_t, d_r, d_o, d_ticks);
    run<1>("1 one line per lane", cus, d_t, d_r, d_o, d_ticks);
    run<2>("2 pairs (32 lines)", cus, d_t, d_r, d_o, d_ticks);
    run<3>("3 quads (16 lines)", cus, d_t, d_r, d_o, d_ticks);
    run<4>("4 octets (8 random lines)", cus, d_t, d_r, d_o, d_ticks);
    run<5>("5 same 16 bytes", cus, d_t, d_r, d_o, d_ticks);
    run<6>("6 one per lane, 27 lanes", cus, d_t, d_r, d_o, d_ticks);
    return 0;
}
