// Micro-benchmark: 29-bit-limb Montgomery product (fp29.hpp f29_mul, the bucket accumulation's multiplier) throughput vs
// waves per SIMD and independent product chains per lane: is a single dependent chain per wave at the accumulation's
// occupancy (3 waves per SIMD) limited by dependent-issue latency?
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../jellyfish_amd/csrc mul29_occ.hip -o mul29_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "fp29.hpp"
using namespace jf;

template <class C, int CH>
__global__ __launch_bounds__(256) void k_mul(uint32_t iters, uint32_t seed, uint32_t* out) {
    extern __shared__ uint32_t dyn[];
    F29<C> x[CH], y[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        x[c] = f29_from_fp<C>(Fp<C>::one());
        y[c] = f29_from_fp<C>(Fp<C>::r2());
        x[c].v[0] ^= (threadIdx.x + c) & 1;
        y[c].v[1] ^= seed & 1;
    }
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) x[c] = f29_mul<C>(x[c], y[c]);
#pragma unroll
        for (int c = 0; c < CH; c++) y[c] = f29_mul<C>(y[c], x[c]);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) acc ^= x[c].v[0] ^ y[c].v[1];
    if (acc == 0x12345u) out[0] = acc + dyn[0];
}

template <int CH>
void run(int waves_per_simd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    uint32_t* d;
    hipMalloc(&d, 4096);
    // one 256-thread workgroup = 1 wave per SIMD; LDS sized so that exactly `waves_per_simd` workgroups fit a CU
    size_t lds = (160 * 1024) / waves_per_simd - 512;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k_mul<Bn254Fq, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const uint32_t iters = 512 / CH, blocks = 256 * 8;
    hipLaunchKernelGGL((k_mul<Bn254Fq, CH>), dim3(blocks), dim3(256), lds, 0, 4, 1u, d);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_mul<Bn254Fq, CH>), dim3(blocks), dim3(256), lds, 0, iters, 1u, d);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double muls = (double)blocks * 256.0 * iters * 2.0 * CH;
    printf("f29_mul chains/lane %d  waves/SIMD %d : %.3e mul/s  (%s)\n", CH, waves_per_simd, muls / (ms * 1e-3), hipGetErrorString(hipGetLastError()));
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 3, 4, 8}) run<1>(w);
    for (int w : {1, 2, 3, 4}) run<2>(w);
    return 0;
}
