// Developer microbenchmark: issue cost (cycles per wave-instruction per SIMD, at 8 waves / SIMD and 4 independent chains
// per wave) of the integer VALU instructions the field kernels are made of, on gfx950.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench/valu_rate.hip -o scripts/ubench/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define KERNEL(NAME, DECL, ASM)                                                                           \
    __global__ __launch_bounds__(64) void NAME(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {      \
        DECL acc[4];                                                                                      \
        uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;                                              \
        for (int c = 0; c < 4; c++) acc[c] = c + threadIdx.x;                                             \
        for (int i = 0; i < iters; i++) {                                                                 \
            _Pragma("unroll") for (int u = 0; u < 16; u++) {                                              \
                _Pragma("unroll") for (int c = 0; c < 4; c++) asm volatile(ASM : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc"); \
            }                                                                                             \
        }                                                                                                 \
        uint64_t s = 0;                                                                                   \
        for (int c = 0; c < 4; c++) s += (uint64_t)acc[c];                                                \
        out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);                           \
    }

KERNEL(k_mad64, uint64_t, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL(k_lshr64, uint64_t, "v_lshrrev_b64 %0, 1, %0")
KERNEL(k_lshladd64, uint64_t, "v_lshl_add_u64 %0, %0, 0, %0")
KERNEL(k_mullo, uint32_t, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mulhi, uint32_t, "v_mul_hi_u32 %0, %0, %1")
KERNEL(k_add, uint32_t, "v_add_u32 %0, %0, %1")
KERNEL(k_addco, uint32_t, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL(k_addc, uint32_t, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL(k_and, uint32_t, "v_and_b32 %0, %0, %1")
KERNEL(k_alignbit, uint32_t, "v_alignbit_b32 %0, %0, %1, 29")
KERNEL(k_add3, uint32_t, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_cndmask, uint32_t, "v_cndmask_b32 %0, %0, %1, vcc")
// select forms: v_cndmask with the mask in an SGPR pair set once before the loop (what compiled code does), and the
// arithmetic select v_bfi_b32 with a 0 / ~0 lane mask in a VGPR
#define KERNEL_SEL(NAME, ASM)                                                                             \
    __global__ __launch_bounds__(64) void NAME(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {      \
        uint32_t acc[4];                                                                                  \
        uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;                                              \
        const uint64_t m = __ballot((threadIdx.x * 2654435761u) >> 31);                                   \
        const uint32_t vm = ((threadIdx.x * 2654435761u) >> 31) ? ~0u : 0u;                               \
        for (int c = 0; c < 4; c++) acc[c] = c + threadIdx.x;                                             \
        for (int i = 0; i < iters; i++) {                                                                 \
            _Pragma("unroll") for (int u = 0; u < 16; u++) {                                              \
                _Pragma("unroll") for (int c = 0; c < 4; c++) asm volatile(ASM : "+v"(acc[c]) : "v"(a), "v"(b), "s"(m), "v"(vm)); \
            }                                                                                             \
        }                                                                                                 \
        uint64_t s = 0;                                                                                   \
        for (int c = 0; c < 4; c++) s += (uint64_t)acc[c];                                                \
        out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);                           \
    }
KERNEL_SEL(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, %3")
KERNEL_SEL(k_bfi, "v_bfi_b32 %0, %4, %0, %1")
KERNEL_SEL(k_and_or, "v_and_or_b32 %0, %0, %4, %1")
KERNEL(k_mad24, uint32_t, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL(k_mov, uint32_t, "v_mov_b32 %0, %1")
KERNEL(k_lshr32, uint32_t, "v_lshrrev_b32 %0, 1, %0")
KERNEL(k_bfe, uint32_t, "v_bfe_u32 %0, %0, 3, 29")
KERNEL(k_dot, uint32_t, "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL(k_madi16, uint32_t, "v_mad_u32_u16 %0, %1, %2, %0")

template <class K>
static void run(const char* name, K kern, uint32_t* d_out) {
    const int iters = 2048, waves = 8, blocks = 256 * 4 * waves;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d_out, 12345u, 999u, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double insts = (double)blocks * iters * 16 * 4;
    const double per_simd_ns = ms * 1e6 / (insts / 1024);
    printf("%-12s %.3f ms  %.3f ns per wave-inst per SIMD  (= %.2f cycles at 2.4 GHz, %.2f relative to v_add_u32)\n", name, ms,
           per_simd_ns, per_simd_ns * 2.4, 0.0);
}

int main() {
    uint32_t* d_out;
    (void)hipMalloc(&d_out, 256 * 4 * 8 * 64 * 4);
#define RUN(K) run(#K, K, d_out)
    RUN(k_add); RUN(k_add); RUN(k_mad64); RUN(k_lshr64); RUN(k_lshladd64); RUN(k_mullo); RUN(k_mulhi); RUN(k_addco); RUN(k_addc);
    RUN(k_and); RUN(k_alignbit); RUN(k_add3); RUN(k_cndmask); RUN(k_cndmask_sgpr); RUN(k_bfi); RUN(k_and_or); RUN(k_mad24); RUN(k_mov); RUN(k_lshr32); RUN(k_bfe);
    RUN(k_dot); RUN(k_madi16);
    (void)hipFree(d_out);
    return 0;
}
