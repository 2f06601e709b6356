// Exploratory micro-benchmark (jfhip_probe kinds 6 and 7; no product path uses it): can the matrix cores multiply a batch of
// 256-bit field elements by a CONSTANT faster than the VALU does?  The NTT's twiddles are constants shared by every
// polynomial of a batch and every column of a tile, so x -> x t mod p is, per twiddle, a fixed linear map on the bytes of
// x: z = T x with T the 64 x 32 Toeplitz matrix of the bytes of t, an int8 GEMM against a batch of x (v_mfma_i32_32x32x32_i8,
// 32 elements per instruction), and the Montgomery step is two more such maps (by -p^-1 mod 2^256 and by p).
//
// kind 6: the VALU baseline -- fp_mul_const29 (fp29.hpp), the product the NTT passes use today, in a dependent chain.
// kind 7: the matrix-core pipeline for the same map, as a LOWER BOUND of a correct implementation: per batch of 32 elements
//   * the 10 MFMA instructions (T and its one-byte shift against the magnitude and sign-bit operands: the instruction is
//     signed x signed and the data bytes are unsigned; 4 for z = T x, 2 for m = z_low p', 4 for z += m p),
//   * the three carry normalisations of the int32 accumulator columns back to bytes (the accumulators hold 64 column sums
//     of up to 2^21 per product; the next GEMM takes bytes),
//   * the re-packing of those bytes into the next GEMM's operand (4 bytes per word, and from the accumulator's row-block
//     layout to the operand's K-major layout: one LDS round trip),
//   * the final conditional subtraction.
//   The arithmetic it performs is NOT a correct modular product (constants are arbitrary, cross-lane carry hand-offs are
//   single shuffles): it has the instruction mix and the dependences of one, nothing more is claimed.  DESIGN.md 3.3 has
//   the numbers.
#include "common.hpp"
#include "fp29.hpp"

namespace jf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <class C>
__global__ __launch_bounds__(256) void k_probe_const29(uint32_t iters, uint32_t seed, uint32_t* __restrict__ out) {
    Fp<C> x = Fp<C>::one();
    x.v[0] ^= (threadIdx.x & 1);
    x = fp_reduce_once<C>(x);
    F29<C> t = f29_const_from_fp<C>(Fp<C>::r2());
    t.v[1] ^= seed & 1;
    for (uint32_t i = 0; i < iters; i++) {
        x = fp_mul_const29<C>(x, t);
        x = fp_mul_const29<C>(x, t);
    }
    if (x.v[0] == 0x12345u && x.v[1] == 0x777u) out[0] = x.v[2];
}

// carry-normalise 16 column sums to bytes; returns the carry out (handed to the next row block)
__device__ __forceinline__ int normalise16(v16i& z, int carry) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int t = z[k] + carry;
        carry = t >> 8;
        z[k] = t & 0xff;
    }
    return carry;
}
// 16 bytes (one per register) -> 4 packed words
__device__ __forceinline__ v4i pack16(const v16i& z) {
    v4i w;
#pragma unroll
    for (int k = 0; k < 4; k++)
        w[k] = z[4 * k] | (z[4 * k + 1] << 8) | (z[4 * k + 2] << 16) | (z[4 * k + 3] << 24);
    return w;
}

__global__ __launch_bounds__(256) void k_probe_mfma_const(uint32_t iters, uint32_t seed, uint32_t* __restrict__ out) {
    __shared__ uint4 xch[256];
    const uint32_t tid = threadIdx.x;
    v4i A[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        const uint32_t h = (tid * 2654435761u) ^ (seed + 77u * k);
        A[k] = v4i{(int)h, (int)(h * 3u), (int)(h * 5u), (int)(h * 7u)};
    }
    v4i xs = {(int)(tid * 11u + seed), (int)(tid * 13u), (int)(tid * 17u), (int)(tid * 19u)};
    const v16i zero = {};
    for (uint32_t it = 0; it < iters; it++) {
        // sign bits of the unsigned bytes, as a second operand against the tiles shifted by one byte
        v4i sg;
#pragma unroll
        for (int k = 0; k < 4; k++) sg[k] = ((uint32_t)xs[k] >> 7) & 0x01010101u;
        v16i z0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0], xs, zero, 0, 0, 0);
        z0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1], sg, z0, 0, 0, 0);
        v16i z1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[2], xs, zero, 0, 0, 0);
        z1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[3], sg, z1, 0, 0, 0);
        // columns -> bytes; the carry between the row blocks of a column crosses lanes
        int c = normalise16(z0, 0);
        c = __shfl_xor(c, 32, 64);
        c = normalise16(z1, c);
        // low half -> operand of the next GEMM: pack, exchange through LDS (accumulator layout -> K-major operand layout)
        v4i lowp = pack16(z0);
        xch[tid] = make_uint4(lowp[0], lowp[1], lowp[2], lowp[3]);
        __builtin_amdgcn_wave_barrier();
        uint4 q = xch[tid ^ 32];
        v4i lo = {(int)q.x, (int)q.y, (int)q.z, (int)q.w};
#pragma unroll
        for (int k = 0; k < 4; k++) sg[k] = ((uint32_t)lo[k] >> 7) & 0x01010101u;
        // m = z_low * (-p^-1) mod 2^256
        v16i m = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[4], lo, zero, 0, 0, 0);
        m = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[5], sg, m, 0, 0, 0);
        (void)normalise16(m, 0);
        v4i mp = pack16(m);
        xch[tid] = make_uint4(mp[0], mp[1], mp[2], mp[3]);
        __builtin_amdgcn_wave_barrier();
        q = xch[tid ^ 32];
        v4i mo = {(int)q.x, (int)q.y, (int)q.z, (int)q.w};
#pragma unroll
        for (int k = 0; k < 4; k++) sg[k] = ((uint32_t)mo[k] >> 7) & 0x01010101u;
        // z += m p: the low half cancels (its carry feeds the high half), the high half is the product
        z0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[6], mo, z0, 0, 0, 0);
        z0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[7], sg, z0, 0, 0, 0);
        z1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[8], mo, z1, 0, 0, 0);
        z1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[9], sg, z1, 0, 0, 0);
        c = normalise16(z0, 0);
        c = __shfl_xor(c, 32, 64);
        c = normalise16(z1, c);
        // conditional subtraction of p on the bytes of the high half (borrow chain + select)
        int borrow = 0;
        v16i d;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int t = z1[k] - (int)((A[9][k & 3] >> (8 * (k & 3))) & 0xff) - borrow;
            borrow = (t >> 8) & 1;
            d[k] = t & 0xff;
        }
        borrow = __shfl_xor(borrow, 32, 64) | (c & 1);
#pragma unroll
        for (int k = 0; k < 16; k++) z1[k] = borrow ? z1[k] : d[k];
        v4i r = pack16(z1);
        xch[tid] = make_uint4(r[0], r[1], r[2], r[3]);
        __builtin_amdgcn_wave_barrier();
        q = xch[tid ^ 32];
        xs = v4i{(int)q.x, (int)q.y, (int)q.z, (int)q.w};
    }
    if (xs[0] == 0x12345 && xs[1] == 0x777) out[0] = (uint32_t)xs[2];
}

// kind 6: arg = field id (0 BN254 Fr, 2 BLS12-381 Fr); kind 7: the matrix-core skeleton.  *out = products per second
int32_t probe_const_mul(int32_t kind, int32_t arg, double* out) {
    hipEvent_t e0, e1;
    JF_HIP(hipEventCreate(&e0));
    JF_HIP(hipEventCreate(&e1));
    void* scratch = nullptr;
    JF_HIP(hipMalloc(&scratch, 4096));
    const uint32_t blocks = 256 * 8;
    double work = 0;
    if (kind == 6) {
        const uint32_t iters = 512;
        auto launch = [&](uint32_t it) {
            if (arg == 2) hipLaunchKernelGGL((k_probe_const29<Bls381Fr>), dim3(blocks), dim3(256), 0, 0, it, 1u, (uint32_t*)scratch);
            else hipLaunchKernelGGL((k_probe_const29<Bn254Fr>), dim3(blocks), dim3(256), 0, 0, it, 1u, (uint32_t*)scratch);
        };
        launch(8);
        JF_HIP(hipEventRecord(e0, 0));
        launch(iters);
        JF_HIP(hipEventRecord(e1, 0));
        work = (double)blocks * 256.0 * iters * 2.0;
    } else {
        const uint32_t iters = 2048;
        hipLaunchKernelGGL(k_probe_mfma_const, dim3(blocks), dim3(256), 0, 0, 8u, 1u, (uint32_t*)scratch);
        JF_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_probe_mfma_const, dim3(blocks), dim3(256), 0, 0, iters, 1u, (uint32_t*)scratch);
        JF_HIP(hipEventRecord(e1, 0));
        work = (double)blocks * 4.0 * iters * 32.0;   // 4 waves per workgroup, 32 products per wave and iteration
    }
    JF_HIP(hipGetLastError());
    JF_HIP(hipEventSynchronize(e1));
    float ms = 0;
    JF_HIP(hipEventElapsedTime(&ms, e0, e1));
    JF_HIP(hipFree(scratch));
    JF_HIP(hipEventDestroy(e0));
    JF_HIP(hipEventDestroy(e1));
    *out = work / ((double)ms * 1e-3);
    return JFHIP_OK;
}

}  // namespace jf
