// Developer microbenchmark: issue rate of v_mad_u64_u32 on gfx950 as a function of the number of independent accumulator
// chains per wave (CH) and of resident waves per SIMD.  Prints wave-instructions per SIMD-cycle equivalents.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench/mad_chain.hip -o gpurun_out/mad_chain
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int CH>
__global__ __launch_bounds__(64) void k_chain(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
    uint64_t acc[CH];
    uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
    for (int c = 0; c < CH; c++) acc[c] = c + threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int c = 0; c < CH; c++)
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) s += acc[c];
    out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

// the same with 32-bit add chains, for the baseline VALU cadence
template <int CH>
__global__ __launch_bounds__(64) void k_addchain(uint32_t* out, uint32_t a0, int iters) {
    uint32_t acc[CH];
    uint32_t a = a0 + threadIdx.x;
#pragma unroll
    for (int c = 0; c < CH; c++) acc[c] = c + threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int c = 0; c < CH; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(a));
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) s += acc[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <class K>
static void run(const char* name, K kern, int ch, int waves_per_simd, uint32_t* d_out, bool mad) {
    const int iters = 4096;
    const int blocks = 256 * 4 * waves_per_simd;   // one-wave blocks: waves_per_simd per SIMD if spread evenly
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0, 0);
        if (mad) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d_out, 12345u, 999u, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * iters * 16 * ch;   // wave-instructions
    printf("%s chains=%d waves/SIMD=%d  %.3f ms  %.1f G wave-inst/s  = %.3f per SIMD per ns\n", name, ch, waves_per_simd,
           ms, insts / ms / 1e6, insts / 1024 / (ms * 1e6));
}

int main() {
    uint32_t* d_out;
    hipMalloc(&d_out, 256 * 4 * 8 * 64 * 4);
    for (int w : {1, 2, 3, 4, 8}) {
        run("mad64", k_chain<1>, 1, w, d_out, true);
        run("mad64", k_chain<2>, 2, w, d_out, true);
        run("mad64", k_chain<4>, 4, w, d_out, true);
    }
    hipFree(d_out);
    return 0;
}
