// valu_calib.hip — what does one VALU wave-instruction cost on gfx950, and what do the SQ counters report for it?
//
// Calibration for DESIGN.md §6 (the "which bound" question): every wave runs ITER x 16 independent v_fma_f32 (16 accumulators,
// no memory traffic) at 1, 2, 4 or 8 waves per SIMD (grid = 256 CUs x K workgroups of 256 threads = K waves on each SIMD), plus
// a DEPENDENT chain variant (one accumulator) that shows the issue->issue latency of one wave's own stream.
// Reports cycles per wave-instruction per SIMD from s_memtime (shader clock) and from the event time at the reported clock.
// Under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU` the per-kernel
// rows give the counter value per instruction (tools/valu_calib_summary.py).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_calib.hip -o gpurun_out/valu_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

static constexpr int ITER = 8192;

template <int K, bool DEP, int LANES = 64>
__global__ void __launch_bounds__(256) fma_kernel(float* out, unsigned long long* ticks, float b, float c) {
    extern __shared__ float s_pad[];      // dynamic LDS = 160 KiB / K: at most K workgroups (K waves per SIMD) fit on a CU
    if (b == 12345.0f) s_pad[threadIdx.x] = c;
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i);
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // LANES < 64: only the first LANES lanes of every wave run the loop (EXEC = the low lanes): does a wave64 instruction whose upper
    // 32 lanes are all off still occupy the SIMD-32 for two passes?
    if ((int)(threadIdx.x & 63) < LANES)
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = wall_clock64();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { ticks[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; ticks[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = w1 - w0; }
}

template <int K, bool DEP, int LANES = 64>
static void run(int cus, float* d_out, unsigned long long* d_ticks, double) {
    // ROUNDS x (cus x K) workgroups: the chip stays full (K resident workgroups per CU, LDS-limited) for ROUNDS generations, so
    // the event time / ROUNDS is the steady-state time of K waves per SIMD and uneven first placement washes out.
    const int ROUNDS = 6, grid = cus * K * ROUNDS;
    const size_t lds = (size_t)(160 * 1024) / K - 512;
    hipFuncSetAttribute((const void*)fma_kernel<K, DEP, LANES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((fma_kernel<K, DEP, LANES>), dim3(grid), dim3(256), lds, 0, d_out, d_ticks, 1.0000001f, 1e-9f);   // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((fma_kernel<K, DEP, LANES>), dim3(grid), dim3(256), lds, 0, d_out, d_ticks, 1.0000001f, 1e-9f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> t((size_t)grid * 8);
    hipMemcpy(t.data(), d_ticks, t.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> own, mhz;
    for (size_t w = 0; w < (size_t)grid * 4; ++w) { own.push_back((double)t[2 * w]); mhz.push_back((double)t[2 * w] / (double)t[2 * w + 1] * 100.0); }
    std::sort(own.begin(), own.end()); std::sort(mhz.begin(), mhz.end());
    const double n_inst = (double)ITER * 16.0;           // wave-instructions per wave
    const double f_ghz = mhz[mhz.size() / 2] * 1e-3;     // shader clock while the kernel ran: s_memtime ticks per 100 MHz wall_clock64 tick
    const double simd_cycles = ms * 1e-3 * f_ghz * 1e9 / (n_inst * K * ROUNDS);
    if (LANES < 64) printf("[first %d lanes only] ", LANES);
    printf("%s K=%d waves/SIMD (LDS-limited, %d rounds): shader clock %.3f GHz (s_memtime vs wall_clock64); a wave needs %.2f cycles per OWN instruction (median); "
           "event %.1f us -> %.2f SIMD cycles per wave-instruction\n",
           DEP ? "dependent  " : "independent", K, ROUNDS, f_ghz, own[own.size() / 2] / n_inst, ms * 1e3, simd_cycles);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("%s: %d CUs, clockRate attribute %.2f GHz\n", p.gcnArchName, cus, ghz);
    float* d_out; unsigned long long* d_ticks;
    hipMalloc(&d_out, (size_t)cus * 8 * 6 * 256 * 4);
    hipMalloc(&d_ticks, (size_t)cus * 8 * 6 * 4 * 2 * 8);
    run<1, false>(cus, d_out, d_ticks, ghz);
    run<2, false>(cus, d_out, d_ticks, ghz);
    run<4, false>(cus, d_out, d_ticks, ghz);
    run<8, false>(cus, d_out, d_ticks, ghz);
    run<1, true>(cus, d_out, d_ticks, ghz);
    run<2, true>(cus, d_out, d_ticks, ghz);
    run<4, true>(cus, d_out, d_ticks, ghz);
    run<8, false, 32>(cus, d_out, d_ticks, ghz);
    run<4, false, 32>(cus, d_out, d_ticks, ghz);
    run<8, false, 16>(cus, d_out, d_ticks, ghz);
    hipFree(d_out); hipFree(d_ticks);
    return 0;
}
