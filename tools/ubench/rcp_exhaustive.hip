// Exhaustive check on the device: is rcp + one FMA Newton step the correctly rounded 1/d (== IEEE 1.0f / d)
// for every fp32 d with 2^-100 <= |d| <= 2^100?   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float fast_rcp(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float fast_rcp2(float d) {          // two steps
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

__global__ void check(uint32_t exp_lo, uint32_t exp_hi, unsigned long long *bad, unsigned long long *bad2, unsigned long long *bad0, uint32_t *examples) {
    const uint64_t per = 1ull << 23;
    const uint64_t total = (uint64_t)(exp_hi - exp_lo + 1) * per;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = ((exp_lo + (uint32_t)(i >> 23)) << 23) | (uint32_t)(i & (per - 1));
        for (uint32_t sgn = 0; sgn < 2; ++sgn) {
            const float d = __uint_as_float(bits | (sgn << 31));
            const float want = 1.0f / d;
            if (__float_as_uint(fast_rcp(d)) != __float_as_uint(want)) {
                const unsigned long long k = atomicAdd(bad, 1ull);
                if (k < 64) examples[k] = bits | (sgn << 31);
            }
            if (__float_as_uint(__builtin_amdgcn_rcpf(d)) != __float_as_uint(want)) atomicAdd(bad0, 1ull);   // sanity: the raw instruction is NOT exact
            if (__float_as_uint(fast_rcp2(d)) != __float_as_uint(want)) atomicAdd(bad2, 1ull);
        }
    }
}

int main() {
    unsigned long long *bad, *bad2, *bad0; uint32_t *ex;
    hipMalloc(&bad, 8); hipMalloc(&bad2, 8); hipMalloc(&bad0, 8); hipMemset(bad0, 0, 8); hipMalloc(&ex, 256);
    hipMemset(bad, 0, 8); hipMemset(bad2, 0, 8); hipMemset(ex, 0, 256);
    check<<<4096, 256>>>(127 - 100, 127 + 100, bad, bad2, bad0, ex);
    unsigned long long hb = 0, hb2 = 0, hb0 = 0; hipMemcpy(&hb0, bad0, 8, hipMemcpyDeviceToHost); uint32_t hex[64];
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hb2, bad2, 8, hipMemcpyDeviceToHost);
    hipMemcpy(hex, ex, 256, hipMemcpyDeviceToHost);
    hipMemcpy(&hb0, bad0, 8, hipMemcpyDeviceToHost);
    printf("raw v_rcp_f32: %llu mismatches\n", hb0);
    printf("one step: %llu mismatches, two steps: %llu mismatches (of %llu inputs)\n", hb, hb2, 2ull * 201 * (1ull << 23));
    for (int k = 0; k < 64 && k < (int)hb; ++k) { float f; __builtin_memcpy(&f, &hex[k], 4); printf("  0x%08x %g\n", hex[k], f); }
    return 0;
}
