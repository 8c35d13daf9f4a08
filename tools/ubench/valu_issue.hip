// Micro-benchmark (analysis only): VALU issue rate per SIMD on gfx950 for the instruction mix of the
// grid traversal (fp32 add/mul, int add, cndmask, cmp, pk_f32, mul_lo_u32, mul_u32_u24, LDS read).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    float a = seed + threadIdx.x, b = a * 0.5f, c = b + 1.f, d = c * 0.25f;
    int ia = threadIdx.x, ib = ia * 3 + 1;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p = {a, b}, q = {c, d};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) { a = a + b; c = c * d; b = b + 1.0f; d = d * 0.999f; }                 // 4 plain fp32
            if (MODE == 1) { ia = ia + ib; ib = ib ^ ia; ia = ia & 0xffff; ib = ib + 7; }            // 4 int
            if (MODE == 2) { a = a < b ? c : a; b = b < c ? d : b; c = c < d ? a : c; d = d < a ? b : d; } // cmp+cndmask x4 = 8
            if (MODE == 3) { p = p + q; q = q * p; p = p + q; q = q * p; }                           // 4 pk
            if (MODE == 4) { ia = ia * ib; ib = ib * 3 + ia; ia = ia * ib; ib = ib * 5 + ia; }       // mul_lo x4 (+adds)
            if (MODE == 5) { ia = __mul24(ia, ib) & 0xffff; ib = __mul24(ib, 3) & 0xfff; ia = __mul24(ia, ib) & 0xffff; ib = __mul24(ib, 5) & 0xfff; }
            if (MODE == 6) { ia = lds[ia & 4095] + ia; ib = lds[ib & 4095] ^ ib; }                   // 2 dependent LDS reads + 4 int
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + ia + ib + p.x + p.y + q.x + q.y;
}

int main() {
    float *out; CHECK(hipMalloc(&out, 512 * 1024 * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 512;
    const char *names[] = {"fp32 add/mul (4/unit)", "int add/xor/and (4)", "cmp+cndmask (8)", "pk_f32 (4)", "mul_lo_u32 (4 mul + 2 add)", "mul24+and (8)", "lds dependent (2 lds + 4 int)"};
    const double instr[] = {4, 4, 8, 4, 6, 8, 6};
#define RUN(M) { k<M><<<blocks, 1024>>>(out, 10, 1.f); hipEventRecord(e0); k<M><<<blocks, 1024>>>(out, iters, 1.f); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
    double waves_per_simd = blocks * 16.0 / 1024.0; double n = instr[M] * 16.0 * iters * waves_per_simd; \
    printf("%-34s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD @2.4GHz (8 waves/SIMD resident)\n", names[M], ms, ms * 1e-3 * 2.4e9 / n); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    return 0;
}
