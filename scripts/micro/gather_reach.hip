// Micro-benchmark: what limits random 64-byte gathers over a multi-GiB table on MI355X (the second wall of
// k_msm_accumulate29)?  Three variants over tables of 1 .. 12 GiB:
//   malloc  : hipMalloc'ed table, every lane gathers uniformly over the whole table (what the MSM does)
//   xcd     : same table, but workgroup g only touches slab g % 8 of the table (workgroups go round-robin to the 8 XCDs,
//             so each XCD's L2 / translation caches see 1/8 of the table)
//   vmm     : table mapped through hipMemCreate / hipMemMap with the recommended granularity (one physical allocation)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/gather_reach scripts/micro/gather_reach.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

template <int XCD>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_gather(const uint4* __restrict__ table,
                                                                                           uint64_t npts, uint32_t L,
                                                                                           uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint32_t h = t * 2654435761u + 12345u;
    const uint64_t span = XCD ? npts / 8 : npts;
    const uint64_t base = XCD ? (uint64_t)(blockIdx.x & 7u) * span : 0;
    for (uint32_t k = 0; k < L; k++) {
        h = h * 1664525u + 1013904223u;
        const uint64_t idx = base + (((uint64_t)(h ^ (h >> 15)) * span) >> 32);   // < base + span <= npts
        const uint4* p = table + idx * 4;
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w;
        acc.y += a.y + b.x;
        acc.z ^= c.x + d.y;
        acc.w += a.w ^ d.x;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) out[0] = acc.x;
}

static double run(const void* tab, size_t bytes, int xcd, uint32_t* out) {
    const uint32_t lanes = 256 * 4 * 4 * 64, L = 256;   // 4 waves per SIMD resident, one round
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&] {
        if (xcd) hipLaunchKernelGGL((k_gather<1>), dim3(lanes / 128), dim3(128), 0, 0, (const uint4*)tab, (uint64_t)(bytes / 64), L, out);
        else hipLaunchKernelGGL((k_gather<0>), dim3(lanes / 128), dim3(128), 0, 0, (const uint4*)tab, (uint64_t)(bytes / 64), L, out);
    };
    launch();
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return (double)lanes * L / (ms * 1e-3);
}

int main(int argc, char** argv) {
    uint32_t* out;
    CK(hipMalloc(&out, 256));
    const int sizes[] = {1, 2, 4, 8, 12};
    for (int s : sizes) {
        const size_t bytes = (size_t)s << 30;
        void* tab = nullptr;
        CK(hipMalloc(&tab, bytes));
        CK(hipMemset(tab, 1, bytes));
        CK(hipDeviceSynchronize());
        const double a = run(tab, bytes, 0, out), b = run(tab, bytes, 1, out);
        printf("{\"GiB\": %d, \"alloc\": \"hipMalloc\", \"uniform_gathers_per_s\": %.4g, \"xcd_slab_gathers_per_s\": %.4g}\n", s, a, b);
        fflush(stdout);
        CK(hipFree(tab));
    }
    // virtual-memory API: one physical allocation mapped with the recommended granularity
    int dev = 0;
    CK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gmin = 0, grec = 0;
    if (hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum) != hipSuccess ||
        hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) {
        printf("{\"vmm\": \"granularity query failed\"}\n");
        return 0;
    }
    printf("{\"vmm_granularity_min\": %zu, \"vmm_granularity_recommended\": %zu}\n", gmin, grec);
    for (int s : {4, 12}) {
        const size_t bytes = (size_t)s << 30;
        hipMemGenericAllocationHandle_t h;
        void* va = nullptr;
        if (hipMemCreate(&h, bytes, &prop, 0) != hipSuccess) {
            printf("{\"vmm\": \"hipMemCreate failed\", \"GiB\": %d}\n", s);
            continue;
        }
        if (hipMemAddressReserve(&va, bytes, (size_t)1 << 30, nullptr, 0) != hipSuccess) {
            printf("{\"vmm\": \"reserve failed\"}\n");
            (void)hipMemRelease(h);
            continue;
        }
        CK(hipMemMap(va, bytes, 0, h, 0));
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, bytes, &acc, 1));
        CK(hipMemset(va, 1, bytes));
        CK(hipDeviceSynchronize());
        const double a = run(va, bytes, 0, out), b = run(va, bytes, 1, out);
        printf("{\"GiB\": %d, \"alloc\": \"hipMemCreate+hipMemMap (va aligned 1 GiB)\", \"uniform_gathers_per_s\": %.4g, \"xcd_slab_gathers_per_s\": %.4g}\n", s, a, b);
        fflush(stdout);
        CK(hipMemUnmap(va, bytes));
        CK(hipMemAddressFree(va, bytes));
        CK(hipMemRelease(h));
    }
    CK(hipFree(out));
    return 0;
}
