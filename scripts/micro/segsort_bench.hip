// Developer micro-benchmark: isolates what makes the segment counting sort slow on big segments.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template <int MODE>
__global__ void k(const uint2* __restrict__ in, const uint32_t* __restrict__ segbase, int shift, int bits,
                  uint2* __restrict__ out) {
    __shared__ uint32_t cnt[512];
    __shared__ uint32_t part[1024];
    const uint32_t nk = 1u << bits, kmask = nk - 1u, nt = blockDim.x;
    const uint32_t lo = segbase[blockIdx.x], hi = segbase[blockIdx.x + 1];
    const uint32_t len = hi - lo;
    for (uint32_t k = threadIdx.x; k < nk; k += nt) cnt[k] = 0;
    __syncthreads();
    if (MODE != 2) {
        for (uint32_t i = threadIdx.x; i < len; i += nt) atomicAdd(&cnt[(in[lo + i].y >> shift) & kmask], 1u);
    }
    __syncthreads();
    if (MODE == 1) { if (threadIdx.x < nk) out[lo + threadIdx.x].x = cnt[threadIdx.x]; return; }
    uint32_t v0 = 0, v1 = 0;
    if (2 * threadIdx.x < nk) v0 = cnt[2 * threadIdx.x];
    if (2 * threadIdx.x + 1 < nk) v1 = cnt[2 * threadIdx.x + 1];
    part[threadIdx.x] = v0 + v1;
    __syncthreads();
    for (uint32_t d = 1; d < nt; d <<= 1) {
        uint32_t t = (threadIdx.x >= d) ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t excl = part[threadIdx.x] - (v0 + v1);
    if (2 * threadIdx.x < nk) cnt[2 * threadIdx.x] = excl;
    if (2 * threadIdx.x + 1 < nk) cnt[2 * threadIdx.x + 1] = excl + v0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < len; i += nt) {
        uint2 en = in[lo + i];
        uint32_t pos;
        if (MODE == 2) pos = i;
        else {
            pos = atomicAdd(&cnt[(en.y >> shift) & kmask], 1u);
            if (MODE == 3) pos = i;
        }
        out[lo + pos] = en;
    }
}

int main(int argc, char** argv) {
    size_t nseg = argc > 1 ? atol(argv[1]) : 1024;
    size_t len = argc > 2 ? atol(argv[2]) : 196608;
    int bits = argc > 3 ? atoi(argv[3]) : 6;
    int nt = argc > 4 ? atoi(argv[4]) : 1024;
    size_t total = nseg * len;
    std::vector<uint2> h(total);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < total; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = make_uint2((uint32_t)i, (uint32_t)(s >> 20)); }
    std::vector<uint32_t> sb(nseg + 1);
    for (size_t i = 0; i <= nseg; i++) sb[i] = (uint32_t)(i * len);
    uint2 *din, *dout; uint32_t* dsb;
    CK(hipMalloc(&din, total * 8)); CK(hipMalloc(&dout, total * 8)); CK(hipMalloc(&dsb, (nseg + 1) * 4));
    CK(hipMemcpy(din, h.data(), total * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), (nseg + 1) * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 4; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nseg), dim3(nt), 0, 0, din, dsb, 5, bits, dout);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nseg), dim3(nt), 0, 0, din, dsb, 5, bits, dout);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nseg), dim3(nt), 0, 0, din, dsb, 5, bits, dout);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(nseg), dim3(nt), 0, 0, din, dsb, 5, bits, dout);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const char* names[] = {"full", "count-only", "copy-no-atomics", "atomics+identity-write"};
        printf("nseg=%zu len=%zu bits=%d nt=%d mode=%s: %.3f ms\n", nseg, len, bits, nt, names[mode], best);
    }
    return 0;
}
